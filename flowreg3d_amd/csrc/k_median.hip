// k_median.hip -- K8: exact 5x5x5 median with mirror boundary
// (scipy.ndimage.median_filter(size=(5,5,5), mode="mirror"), core/optical_flow_3d.py:517-526).
//
// Rank 62 of 125 by a selection network: the window is held in 128 VGPRs (3 slots padded with
// +inf) and pushed through Batcher's odd-even merge sort with every index static, of which the
// compiler keeps only the min/max operations that can reach output 62.  Compute-bound, exact for
// any input (no histogram / approximation), no scratch.
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

// Batcher's odd-even merge sort on a register array, fully unrolled (all indices static).  Only
// a[62] is read afterwards, so the compiler removes every min/max that cannot reach it: what is
// left is a rank-62 selection network (~1.2k min/max pairs instead of the 4.1k compare-exchanges
// of round-by-round forgetful selection).
template <int LO, int N, int R>
struct OEMerge {
    static __device__ __forceinline__ void run(float (&a)[128])
    {
        constexpr int M = R * 2;
        if constexpr (M < N) {
            OEMerge<LO, N, M>::run(a);
            OEMerge<LO + R, N, M>::run(a);
#pragma unroll
            for (int i = LO + R; i + R < LO + N; i += M) cex(a[i], a[i + R]);
        } else {
            cex(a[LO], a[LO + R]);
        }
    }
};
template <int LO, int N>
struct OESort {
    static __device__ __forceinline__ void run(float (&a)[128])
    {
        if constexpr (N > 1) {
            OESort<LO, N / 2>::run(a);
            OESort<LO + N / 2, N / 2>::run(a);
            OEMerge<LO, N, 1>::run(a);
        }
    }
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
    float a[128];
#pragma unroll
    for (int n = 0; n < 125; n++) a[n] = in[zo[n / 25] + yo[(n / 5) % 5] + xo[n % 5]];
    a[125] = a[126] = a[127] = __builtin_inff();  // padding sorts to the top: rank 62 is unchanged
    OESort<0, 128>::run(a);
    out[t] = a[62];
}

void launch_median5(hipStream_t st, const float *in, int Z, int Y, int X, float *out)
{
    long long total = (long long)Z * Y * X;
    hipLaunchKernelGGL(k_median5, dim3(cdiv(total, 256)), dim3(256), 0, st, in, Z, Y, X, out);
}

}  // namespace fr3d
