// k_misc.hip -- K9: pointwise helpers (accumulate, fill, channels-last <-> planar).
#include "fr3d_internal.h"

namespace fr3d {

__global__ void __launch_bounds__(256) k_axpy(float *__restrict__ y, const float *__restrict__ x, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = y[t] + x[t];
}

__global__ void __launch_bounds__(256) k_fill(float *__restrict__ y, float v, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = v;
}

__global__ void __launch_bounds__(256)
k_pack(const float *__restrict__ planar, int C, long long n, float *__restrict__ inter)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    long long v = t / C;
    int c = (int)(t - v * C);
    inter[t] = planar[(size_t)c * n + v];
}

__global__ void __launch_bounds__(256)
k_unpack(const float *__restrict__ inter, int C, long long n, float *__restrict__ planar)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int c = (int)(t / n);
    long long v = t - (long long)c * n;
    planar[t] = inter[(size_t)v * C + c];
}

void launch_axpy(hipStream_t st, float *y, const float *x, long long n)
{
    if (n > 0) hipLaunchKernelGGL(k_axpy, dim3(cdiv(n, 256)), dim3(256), 0, st, y, x, n);
}
void launch_fill(hipStream_t st, float *y, float v, long long n)
{
    if (n > 0) hipLaunchKernelGGL(k_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, y, v, n);
}
void launch_pack(hipStream_t st, const float *planar, int C, long long n, float *inter)
{
    if (n > 0) hipLaunchKernelGGL(k_pack, dim3(cdiv(n * C, 256)), dim3(256), 0, st, planar, C, n, inter);
}
void launch_unpack(hipStream_t st, const float *inter, int C, long long n, float *planar)
{
    if (n > 0) hipLaunchKernelGGL(k_unpack, dim3(cdiv(n * C, 256)), dim3(256), 0, st, inter, C, n, planar);
}

}  // namespace fr3d
