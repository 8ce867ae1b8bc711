// Micro-benchmark: does reading the solver's per-voxel record as 4-B-per-lane SoA streams or as
// 16-B-per-lane packed streams change the achievable HBM rate?  (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR>
__global__ void __launch_bounds__(256) k_soa(const float *const *in, float *o0, float *o1, float *o2, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NR; a++) s += in[a][i];
    o0[i] = s; o1[i] = s * 0.5f; o2[i] = s * 0.25f;
}
template <int NR4>
__global__ void __launch_bounds__(256) k_aos(const float4 *const *in, float *o0, float *o1, float *o2, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NR4; a++) { float4 v = in[a][i]; s += v.x + v.y + v.z + v.w; }
    o0[i] = s; o1[i] = s * 0.5f; o2[i] = s * 0.25f;
}
int main()
{
    const long long n = 1LL << 27;  // 134M voxels
    const int NR = 12;
    float *buf[NR], *o[3];
    for (int a = 0; a < NR; a++) { CK(hipMalloc(&buf[a], n * 4)); CK(hipMemset(buf[a], 0, n * 4)); }
    for (int a = 0; a < 3; a++) CK(hipMalloc(&o[a], n * 4));
    const float **dptr; const float4 **dptr4;
    CK(hipMalloc(&dptr, NR * sizeof(void *))); CK(hipMalloc(&dptr4, 3 * sizeof(void *)));
    CK(hipMemcpy(dptr, buf, NR * sizeof(void *), hipMemcpyHostToDevice));
    // packed view: 3 float4 arrays over the same memory (buf[0..3] contiguous? no -> allocate)
    float4 *p4[3];
    for (int a = 0; a < 3; a++) { CK(hipMalloc(&p4[a], n * 16)); CK(hipMemset(p4[a], 0, n * 16)); }
    CK(hipMemcpy(dptr4, p4, 3 * sizeof(void *), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)n * 4 * (NR + 3);
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_soa<NR>, dim3((n + 255) / 256), dim3(256), 0, 0, dptr, o[0], o[1], o[2], n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("SoA 12x4B reads + 3 writes : %.2f TB/s\n", bytes * 5 / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_aos<3>, dim3((n + 255) / 256), dim3(256), 0, 0, dptr4, o[0], o[1], o[2], n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("AoS 3x16B reads + 3 writes : %.2f TB/s\n", bytes * 5 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
