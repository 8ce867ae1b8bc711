// k_median.hip -- K8: exact 5x5x5 median with mirror boundary
// (scipy.ndimage.median_filter(size=(5,5,5), mode="mirror"), core/optical_flow_3d.py:517-526).
//
// Rank 62 of 125 by "forgetful selection": keep a register-resident working set that starts with
// 64 samples; its minimum and maximum can never be the median, so both are dropped and one new
// sample is taken in -- 61 rounds later three samples are left and their median is the answer.
// All loops are fully unrolled so the working set stays in VGPRs (no scratch); compute-bound
// (~4.1k compare-exchanges per voxel), exact for any input (no histogram / approximation).
#include "fr3d_internal.h"

namespace fr3d {

__device__ __forceinline__ void cex(float &a, float &b)
{
    float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

__device__ __forceinline__ int mirror(int i, int n)
{
    // whole-sample symmetric: -1 -> 1, n -> n-2 (window radius 2, n may be as small as 1)
    if (n == 1) return 0;
    int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

template <int N>
struct Round {
    // working set a[0..N-1] -> drop min and max, take sample number (125 - (N - 3)) ... see caller
    template <typename F>
    static __device__ __forceinline__ void run(float (&a)[64], F &&next)
    {
#pragma unroll
        for (int q = 1; q < N; q++) cex(a[0], a[q]);
#pragma unroll
        for (int q = 1; q < N - 1; q++) cex(a[q], a[N - 1]);
        a[0] = next(64 + (64 - N));
        Round<N - 1>::run(a, next);
    }
};
template <>
struct Round<3> {
    template <typename F>
    static __device__ __forceinline__ void run(float (&)[64], F &&) {}
};

__global__ void __launch_bounds__(256)
k_median5(const float *__restrict__ in, int Z, int Y, int X, float *__restrict__ out)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    long long zo[5], yo[5];
    int xo[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        zo[q] = (long long)mirror(z + q - 2, Z) * Y * X;
        yo[q] = (long long)mirror(y + q - 2, Y) * X;
        xo[q] = mirror(x + q - 2, X);
    }
    auto sample = [&](int n) -> float {  // n in 0..124, window enumerated z-major
        return in[zo[n / 25] + yo[(n / 5) % 5] + xo[n % 5]];
    };
    float a[64];
#pragma unroll
    for (int q = 0; q < 64; q++) a[q] = sample(q);
    Round<64>::run(a, sample);
    // median of the last three
    float lo = fminf(a[0], a[1]), hi = fmaxf(a[0], a[1]);
    out[t] = fmaxf(lo, fminf(hi, a[2]));
}

void launch_median5(hipStream_t st, const float *in, int Z, int Y, int X, float *out)
{
    long long total = (long long)Z * Y * X;
    hipLaunchKernelGGL(k_median5, dim3(cdiv(total, 256)), dim3(256), 0, st, in, Z, Y, X, out);
}

}  // namespace fr3d
