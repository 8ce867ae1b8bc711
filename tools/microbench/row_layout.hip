// Micro-benchmark: does the HOLE in every row of the skewed layout cost bandwidth?
// The SOR operands live in rows of pitch Yp (a multiple of 64 elements) of which only the first `len`
// elements are valid (len is between 1 and min(X,Y); the mean is about a third of Yp at 256^3).  A workgroup
// streams 12 arrays (9 frozen-system entries + 3 increments) of a tile of 4 rows x 64 lanes and writes 3.
// Variants, same number of waves / loads / useful bytes:
//   holes   : row pitch P = 256 elements, valid prefix L          (what the engine does)
//   dense   : row pitch = ceil(L/64)*64                            (rows packed, 256-B aligned starts)
//   dense9  : dense, and the 9 system entries interleaved per voxel (one 36-B record), increments as 12-B records
// usage: ./row_layout   (prints useful TB/s for several L)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256)
k_soa(const float *__restrict__ base, long long arr_stride, float *__restrict__ out, long long out_stride, int L,
      int pitch, int tiles_per_row, long long rows)
{
    const long long tile = blockIdx.x;
    const long long row = (tile / tiles_per_row) * 4 + threadIdx.y;
    const int jj = (int)(tile % tiles_per_row) * 64 + threadIdx.x;
    if (row >= rows || jj >= L) return;
    const long long o = row * pitch + jj;
    float v[12];
#pragma unroll
    for (int a = 0; a < 12; a++) v[a] = base[a * arr_stride + o];
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 12; a++) s += v[a];
    out[o] = s; out[out_stride + o] = s * 0.5f; out[2 * out_stride + o] = s * 0.25f;
}

struct __attribute__((packed)) Rec9 { float m[9]; };
struct __attribute__((packed)) Rec3 { float d[3]; };
__global__ void __launch_bounds__(256)
k_aos(const Rec9 *__restrict__ M, const Rec3 *__restrict__ D, Rec3 *__restrict__ out, int L, int pitch,
      int tiles_per_row, long long rows)
{
    const long long tile = blockIdx.x;
    const long long row = (tile / tiles_per_row) * 4 + threadIdx.y;
    const int jj = (int)(tile % tiles_per_row) * 64 + threadIdx.x;
    if (row >= rows || jj >= L) return;
    const long long o = row * pitch + jj;
    const Rec9 m = M[o];
    const Rec3 d = D[o];
    float s = d.d[0] + d.d[1] + d.d[2];
#pragma unroll
    for (int a = 0; a < 9; a++) s += m.m[a];
    Rec3 r;
    r.d[0] = s; r.d[1] = s * 0.5f; r.d[2] = s * 0.25f;
    out[o] = r;
}

int main()
{
    const int Ls[] = {40, 64, 100, 128, 170, 200, 256};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long cap = 1LL << 27;  // elements per array (512 MiB)
    float *in, *out;
    CK(hipMalloc(&in, cap * 4 * 12)); CK(hipMemset(in, 0, cap * 4 * 12));
    CK(hipMalloc(&out, cap * 4 * 3));
    for (int L : Ls) {
        for (int variant = 0; variant < 3; variant++) {
            const int pitch = variant == 0 ? 256 : ((L + 63) / 64) * 64;
            const long long rows = (cap / 256);  // same number of rows (= useful work) in every variant
            const int tpr = (L + 63) / 64;
            const long long tiles = (rows / 4) * tpr;
            const double useful = (double)rows * L * 4.0 * 15.0;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < 4; r++) {
                    if (variant < 2) hipLaunchKernelGGL(k_soa, dim3((unsigned)tiles), dim3(64, 4), 0, 0, in, cap, out, cap, L, pitch, tpr, rows);
                    else hipLaunchKernelGGL(k_aos, dim3((unsigned)tiles), dim3(64, 4), 0, 0, (const Rec9 *)in, (const Rec3 *)(in + cap * 9), (Rec3 *)out, L, pitch, tpr, rows);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("L %3d  %-7s pitch %3d : %.2f TB/s useful (%.1f us per launch)\n", L, variant == 0 ? "holes" : variant == 1 ? "dense" : "dense9",
                   pitch, useful * 4 / (ms * 1e-3) / 1e12, ms * 1e3 / 4);
        }
    }
    return 0;
}
