// k_median.hip -- K8: exact 5x5x5 median with mirror boundary
// (scipy.ndimage.median_filter(size=(5,5,5), mode="mirror"), core/optical_flow_3d.py:517-526).
//
// Rank 62 of 125 by selection networks built from Batcher's odd-even merge sort with every index
// static, of which the compiler keeps only the min/max operations that can reach the wanted outputs.
// k_median5_lds (default) sorts every 5x5 slab once per workgroup and builds two neighbouring outputs
// per thread from six sorted slabs (345 min/max pairs per output); k_median5_x2 is the same pairing
// without the LDS exchange (631 pairs) and k_median5 the one-output-per-thread form (1122 pairs),
// both kept for A/B runs (FR3D_MEDIAN=2 / 1).  Compute-bound, exact for any input (no
// histogram / approximation).
#include <cstdlib>

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


// ---- two outputs per thread -------------------------------------------------------------------
// The windows of (x0, x0+1) share the four x-columns x0-1..x0+2 = 100 of their 125 elements.  An
// element of that shared set S with S-rank r has window rank r..r+25, so only S[37..62] can be the
// rank-62 element of either window: the thread sorts S once (network pruned to those 26 outputs),
// then per output sorts the private 5x5 slab P (x0-2 resp. x0+3) and takes rank 25 of S[37..62] u P
// as min_i max(S[36+i], P[25-i]), i = 1..26 (the k-th smallest of two sorted lists as a min of
// maxes; P[-1] = -inf).  631 min/max pairs per output instead of 1122
// (tools/numerics/median_network.py), no padding values anywhere: comparators that would touch a
// slot >= NREAL are dropped statically (+inf pads in the top slots never move in a sorting network).
template <int NARR, int NREAL, int LO, int N, int R>
struct OEMergeB {
    static __device__ __forceinline__ void run(float (&a)[NARR])
    {
        constexpr int M = R * 2;
        if constexpr (M < N) {
            OEMergeB<NARR, NREAL, LO, N, M>::run(a);
            OEMergeB<NARR, NREAL, LO + R, N, M>::run(a);
#pragma unroll
            for (int i = LO + R; i + R < LO + N; i += M)
                if (i + R < NREAL) cex(a[i], a[i + R]);
        } else {
            if constexpr (LO + R < NREAL) cex(a[LO], a[LO + R]);
        }
    }
};
template <int NARR, int NREAL, int LO, int N>
struct OESortB {
    static __device__ __forceinline__ void run(float (&a)[NARR])
    {
        if constexpr (N > 1 && LO < NREAL) {
            OESortB<NARR, NREAL, LO, N / 2>::run(a);
            OESortB<NARR, NREAL, LO + N / 2, N / 2>::run(a);
            OEMergeB<NARR, NREAL, LO, N, 1>::run(a);
        }
    }
};

__device__ __forceinline__ float select_shared_private(const float (&s)[128], const float (&p)[32])
{
    // rank 25 of s[37..62] (26 sorted) u p[0..24] (25 sorted)
    float best = s[62];  // i = 26: max(S[62], -inf)
#pragma unroll
    for (int i = 1; i <= 25; i++) best = fminf(best, fmaxf(s[36 + i], p[25 - i]));
    return best;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
k_median5_x2(const float *__restrict__ in, int Z, int Y, int X, float *__restrict__ out)
{
    const int XP = (X + 1) >> 1;  // output pairs per row
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * XP;
    if (t >= total) return;
    const int xp = (int)(t % XP);
    long long r = t / XP;
    const int y = (int)(r % Y);
    const int z = (int)(r / Y);
    const int x0 = 2 * xp;
    long long zo[5], yo[5];
    int xo[6];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        zo[q] = (long long)mirror(z + q - 2, Z) * Y * X;
        yo[q] = (long long)mirror(y + q - 2, Y) * X;
    }
#pragma unroll
    for (int q = 0; q < 6; q++) xo[q] = mirror(x0 + q - 2, X);  // columns x0-2 .. x0+3
    float s[128];
#pragma unroll
    for (int n = 0; n < 100; n++) s[n] = in[zo[n / 20] + yo[(n / 4) % 5] + xo[1 + n % 4]];
    OESortB<128, 100, 0, 128>::run(s);
    const size_t o = ((size_t)z * Y + y) * X + x0;
    {
        float p[32];
#pragma unroll
        for (int n = 0; n < 25; n++) p[n] = in[zo[n / 5] + yo[n % 5] + xo[0]];
        OESortB<32, 25, 0, 32>::run(p);
        out[o] = select_shared_private(s, p);
    }
    if (x0 + 1 < X) {
        float p[32];
#pragma unroll
        for (int n = 0; n < 25; n++) p[n] = in[zo[n / 5] + yo[n % 5] + xo[5]];
        OESortB<32, 25, 0, 32>::run(p);
        out[o + 1] = select_shared_private(s, p);
    }
}


// ---- slabs sorted once, exchanged through LDS ---------------------------------------------------
// The 5x5 (z,y) slab at one x belongs to five windows.  A workgroup of 4 rows x 64 output pairs sorts
// every slab of its 133 columns once (each thread its own two columns; the 20 halo columns of the four
// rows in one extra pass of one wave), publishes them in LDS as order-preserving integer keys, and each thread then builds
// its pair of medians from six sorted slabs: the four shared ones are MERGED (odd-even merge levels
// only, 360 min/max pairs for the 26 candidate ranks instead of 932 for sorting them from scratch),
// the two private ones are used as they are.  345 pairs per output.  Lists of 25 sit in 32-slot
// blocks padded with INT_MAX; integer min/max against that constant folds away at compile time
// (float min/max against +inf cannot, because of NaN semantics), which is why keys are integers.
// (A fifth wave for the halo columns was tried: 320-thread workgroups fit only twice per CU next to
// 53 KB of LDS each, 5.0 ms instead of 4.2 ms per 256^3 volume.)
__device__ __forceinline__ int f2key(float f)
{
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);  // signed-integer order == float order; its own inverse
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
__device__ __forceinline__ void cexi(int &a, int &b)
{
    const int lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}
template <int NARR, int LO, int N, int R>
struct OEMergeI {
    static __device__ __forceinline__ void run(int (&a)[NARR])
    {
        constexpr int M = R * 2;
        if constexpr (M < N) {
            OEMergeI<NARR, LO, N, M>::run(a);
            OEMergeI<NARR, LO + R, N, M>::run(a);
#pragma unroll
            for (int i = LO + R; i + R < LO + N; i += M) cexi(a[i], a[i + R]);
        } else {
            cexi(a[LO], a[LO + R]);
        }
    }
};
template <int NARR, int LO, int N>
struct OESortI {
    static __device__ __forceinline__ void run(int (&a)[NARR])
    {
        if constexpr (N > 1) {
            OESortI<NARR, LO, N / 2>::run(a);
            OESortI<NARR, LO + N / 2, N / 2>::run(a);
            OEMergeI<NARR, LO, N, 1>::run(a);
        }
    }
};

#define MB_Y 4
#define MB_XP 64                    // output pairs per block row -> 128 outputs, 133 slab columns
#define MB_NE (MB_XP + 3)           // even columns c = 0,2,..,132
#define MB_NO (MB_XP + 2)           // odd columns  c = 1,3,..,131
#define KEY_PAD 0x7fffffff

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
k_median5_lds(const float *__restrict__ in, int Z, int Y, int X, float *__restrict__ out)
{
    // column c of the block is x = X0 - 2 + c; even and odd columns are stored apart so that the
    // lanes of a wave (consecutive pairs) read consecutive words
    __shared__ int sE[25][MB_Y][MB_NE];
    __shared__ int sO[25][MB_Y][MB_NO];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int z = blockIdx.z;
    const int y = blockIdx.y * MB_Y + ty;
    const int X0 = blockIdx.x * (2 * MB_XP);
    const bool row_ok = y < Y;
    long long zo[5], yo[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        zo[q] = (long long)mirror(z + q - 2, Z) * Y * X;
        yo[q] = (long long)mirror(row_ok ? y + q - 2 : 0, Y) * X;
    }
    auto sort_column = [&](int c, int row, const long long (&yoff)[5]) {
        const int xm = mirror(X0 - 2 + c, X);
        int k[32];
#pragma unroll
        for (int n = 0; n < 25; n++) k[n] = f2key(in[zo[n / 5] + yoff[n % 5] + xm]);
#pragma unroll
        for (int n = 25; n < 32; n++) k[n] = KEY_PAD;
        OESortI<32, 0, 32>::run(k);
        if (c & 1) {
#pragma unroll
            for (int n = 0; n < 25; n++) sO[n][row][c >> 1] = k[n];
        } else {
#pragma unroll
            for (int n = 0; n < 25; n++) sE[n][row][c >> 1] = k[n];
        }
    };
    if (row_ok) {
        sort_column(2 * tx + 2, ty, yo);
        sort_column(2 * tx + 3, ty, yo);
    }
    // the 4 x 5 halo columns (c = 0,1 and 130,131,132 of every row) cost one more sorting pass of ONE wave
    // (a pass costs the same for 5 active lanes as for 64); which wave takes it rotates with the block so
    // that the extra pass does not always land on the same SIMD
    if (ty == ((blockIdx.x + blockIdx.y + blockIdx.z) & (MB_Y - 1)) && tx < 5 * MB_Y) {
        const int hrow = tx / 5, h = tx - 5 * hrow;
        const int hy = blockIdx.y * MB_Y + hrow;
        if (hy < Y) {
            long long hyo[5];
#pragma unroll
            for (int q = 0; q < 5; q++) hyo[q] = (long long)mirror(hy + q - 2, Y) * X;
            sort_column(h < 2 ? h : 2 * MB_XP + h, hrow, hyo);
        }
    }
    __syncthreads();
    const int x0 = X0 + 2 * tx;
    if (!row_ok || x0 >= X) return;
    // columns x0-2 .. x0+3 are c = 2tx .. 2tx+5: even ones at sE[..][tx + 0/1/2], odd at sO[..][tx + 0/1/2]
    int s[128];
#pragma unroll
    for (int n = 0; n < 25; n++) {
        s[n] = sO[n][ty][tx];            // x0-1
        s[32 + n] = sE[n][ty][tx + 1];   // x0
        s[64 + n] = sO[n][ty][tx + 1];   // x0+1
        s[96 + n] = sE[n][ty][tx + 2];   // x0+2
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int n = 25; n < 32; n++) s[32 * b + n] = KEY_PAD;
    OEMergeI<128, 0, 64, 1>::run(s);
    OEMergeI<128, 64, 64, 1>::run(s);
    OEMergeI<128, 0, 128, 1>::run(s);
    const size_t o = ((size_t)z * Y + y) * X + x0;
    {
        int best = s[62];
#pragma unroll
        for (int i = 1; i <= 25; i++) best = min(best, max(s[36 + i], sE[25 - i][ty][tx]));  // private x0-2
        out[o] = key2f(best);
    }
    if (x0 + 1 < X) {
        int best = s[62];
#pragma unroll
        for (int i = 1; i <= 25; i++) best = min(best, max(s[36 + i], sO[25 - i][ty][tx + 2]));  // private x0+3
        out[o + 1] = key2f(best);
    }
}

// ---- flattened form: no tile quantisation, three fields per launch ------------------------------
// k_median5_lds ties a wave to 128 consecutive outputs of one row: a 131-wide level fills 51 % of its
// lanes, the five levels of a 256^3 pyramid 79 % on average.  Here the output pairs of a whole field are
// numbered consecutively, P = (z*Y + y)*XP + xp, and a workgroup takes 256 consecutive pairs wherever
// the rows break.  Slot L+1 of the exchange arrays holds the two sorted slabs of lane L (columns 2xp and
// 2xp+1), slots 0 and 257 the two columns left of lane 0 and right of lane 255 (one extra pass of one
// wave, four lanes).  The mirror boundary needs no slabs of its own: a mirrored column is a column of the
// same row, so a lane at a row start or end only picks different slots.  Same network, same keys, same
// result as k_median5_lds; `acc` adds the median to the destination (the flow update u += median(du),
// core/optical_flow_3d.py:527-529) instead of storing it.
#define MF_T 256
#define MF_W (MF_T + 2)

struct MedianDst {
    float *p[3];
    int acc;
};

// mirror() for an index at most 2 outside [0, n) of an axis with n >= 3: no modulo (the general form costs an
// emulated integer division per call, ~20 instructions, a dozen times per lane)
__device__ __forceinline__ int mirror2(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__device__ __forceinline__ void median_offsets(int z, int y, int Z, int Y, int X, unsigned (&zo)[5], unsigned (&yo)[5])
{
#pragma unroll
    for (int q = 0; q < 5; q++) {
        zo[q] = (unsigned)mirror2(z + q - 2, Z) * (unsigned)Y * (unsigned)X;
        yo[q] = (unsigned)mirror2(y + q - 2, Y) * (unsigned)X;
    }
}

// TIN = double (exact level tail of the fp64 / packed solver modes): the keys are the values ROUNDED to fp32 -- the
// result is RN32(median), because rounding is monotone -- and k_median5_refine below recovers the fp64 median from it.
template <typename TIN>
__global__ void __launch_bounds__(MF_T) __attribute__((amdgpu_waves_per_eu(3, 4)))
k_median5_flat(const TIN *__restrict__ in, long long fstride, int Z, int Y, int X, unsigned npairs, MedianDst dst)
{
    __shared__ int sK[2][25][MF_W];  // [column parity][rank][slot]
    const int L = threadIdx.x;
    const int XP = (X + 1) >> 1;
    const TIN *__restrict__ src = in + (size_t)blockIdx.y * fstride;
    auto sort_store = [&](const unsigned (&zo)[5], const unsigned (&yo)[5], int xm, int par, int slot) {
        int k[32];
#pragma unroll
        for (int n = 0; n < 25; n++) k[n] = f2key((float)src[zo[n / 5] + yo[n % 5] + (unsigned)xm]);
#pragma unroll
        for (int n = 25; n < 32; n++) k[n] = KEY_PAD;
        OESortI<32, 0, 32>::run(k);
#pragma unroll
        for (int n = 0; n < 25; n++) sK[par][n][slot] = k[n];
    };
    const unsigned P = blockIdx.x * (unsigned)MF_T + (unsigned)L;
    const bool live = P < npairs;
    int xp = 0, y = 0, z = 0;
    if (live) {
        xp = (int)(P % (unsigned)XP);
        const unsigned r = P / (unsigned)XP;
        y = (int)(r % (unsigned)Y);
        z = (int)(r / (unsigned)Y);
        unsigned zo[5], yo[5];
        median_offsets(z, y, Z, Y, X, zo, yo);
        sort_store(zo, yo, 2 * xp, 0, L + 1);
        sort_store(zo, yo, mirror2(2 * xp + 1, X), 1, L + 1);
    }
    // the columns outside the workgroup: x0-2, x0-1 of its first lane and x0+2, x0+3 of its last one.  One
    // more sorting pass of one wave (a pass costs the same for 4 active lanes as for 64); the wave rotates
    // with the workgroup so that the pass does not always land on the same SIMD.
    if ((L >> 6) == (int)((blockIdx.x + blockIdx.y) & 3u) && (L & 63) < 4) {
        const int h = L & 63;
        const unsigned PH = blockIdx.x * (unsigned)MF_T + (h < 2 ? 0u : (unsigned)(MF_T - 1));
        if (PH < npairs) {
            const int hxp = (int)(PH % (unsigned)XP);
            const unsigned r = PH / (unsigned)XP;
            unsigned zo[5], yo[5];
            median_offsets((int)(r / (unsigned)Y), (int)(r % (unsigned)Y), Z, Y, X, zo, yo);
            const int col = 2 * hxp + (h == 0 ? -2 : h == 1 ? -1 : h == 2 ? 2 : 3);
            sort_store(zo, yo, mirror2(col, X), h & 1, h < 2 ? 0 : MF_W - 1);
        }
    }
    __syncthreads();
    if (!live) return;
    // slots of the six columns x0-2 .. x0+3 (even columns in sK[0], odd ones in sK[1]); at a row start
    // -2 -> 2 and -1 -> 1, at a row end X -> X-2 and X+1 -> X-3 (whole-sample symmetric)
    const bool first = xp == 0, last = xp == XP - 1, oddX = (X & 1) != 0;
    const int own = L + 1;
    const int sl_m2 = first ? L + 2 : L;                   // x0-2 (even)
    const int sl_m1 = first ? own : L;                     // x0-1 (odd)
    const int sl_p2 = last ? (oddX ? L : own) : L + 2;     // x0+2 (even)
    const int sl_p3 = last ? L : L + 2;                    // x0+3 (odd); not read for the last pair of an odd row
    int s[128];
#pragma unroll
    for (int n = 0; n < 25; n++) {
        s[n] = sK[1][n][sl_m1];
        s[32 + n] = sK[0][n][own];
        s[64 + n] = sK[1][n][own];
        s[96 + n] = sK[0][n][sl_p2];
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int n = 25; n < 32; n++) s[32 * b + n] = KEY_PAD;
    OEMergeI<128, 0, 64, 1>::run(s);
    OEMergeI<128, 64, 64, 1>::run(s);
    OEMergeI<128, 0, 128, 1>::run(s);
    float *__restrict__ out = dst.p[blockIdx.y];
    const size_t o = ((size_t)z * Y + y) * X + 2 * xp;
    {
        int best = s[62];
#pragma unroll
        for (int i = 1; i <= 25; i++) best = min(best, max(s[36 + i], sK[0][25 - i][sl_m2]));  // private x0-2
        const float m = key2f(best);
        out[o] = dst.acc ? out[o] + m : m;
    }
    if (2 * xp + 1 < X) {
        int best = s[62];
#pragma unroll
        for (int i = 1; i <= 25; i++) best = min(best, max(s[36 + i], sK[1][25 - i][sl_p3]));  // private x0+3
        const float m = key2f(best);
        out[o + 1] = dst.acc ? out[o + 1] + m : m;
    }
}

// The fp64 median of a 5^3 window from its fp32 rounding v = RN32(median): the median is the window element x with
// RN32(x) == v whose rank among those elements is 62 - #{x : RN32(x) < v} -- almost always there is exactly one such
// element.  One thread per voxel scans its 125 window values (neighbouring lanes read neighbouring addresses; the
// window lives in L1/L2), then adds the median to the level flow the way the reference does (core/optical_flow_3d.py:
// 517-529: fp64 median, u = u + du in fp64, rounded to fp32 by the next level's resampler, util/resize_util_3D.py:116):
// u32 <- RN32(u32 + median64), ONE rounding.  (The fp32 tail -- median of fp32-rounded increments, fp32 add -- rounds
// twice; the 1-ulp differences that makes in the level flow are amplified by an ill-conditioned iteration to 2.8e-4
// voxels on BASELINE config 5: DESIGN.md section 2.)
#define MR_TX 64
#define MR_TY 4
#define MR_ZB 16  // z-slabs a workgroup marches through
__global__ void __launch_bounds__(MR_TX * MR_TY)
k_median5_refine(const double *__restrict__ in, const float *__restrict__ v32, long long fstride, int Z, int Y, int X,
                 int nzb, MedianDst dst, int dbg)
{
    // A workgroup owns 64 x 4 (x,y) columns and marches along z with a ring of five window slabs in LDS (8 rows x 68
    // values each, mirror boundary applied while loading): one global read per value and slab, the 125 window reads
    // of a voxel come from LDS.  (Reading the window straight from global memory cost 1 KB of L2 traffic per voxel:
    // 15 ms per 256^3 volume instead of 2.)
    __shared__ double tile[5][MR_TY + 4][MR_TX + 4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * MR_TX + tx;
    const int x0 = blockIdx.x * MR_TX, y0 = blockIdx.y * MR_TY;
    const int x = x0 + tx, y = y0 + ty;
    const int field = blockIdx.z / nzb, z0 = (blockIdx.z % nzb) * MR_ZB;
    const double *__restrict__ src = in + (size_t)field * fstride;
    float *__restrict__ out = dst.p[field];
    const bool live = x < X && y < Y;
    auto load_slab = [&](int zs) {
        const int slot = (zs + 10) % 5;
        const size_t zb = (size_t)mirror2(zs < -2 ? -2 : (zs > Z + 1 ? Z + 1 : zs), Z) * Y;
        for (int e = tid; e < (MR_TY + 4) * (MR_TX + 4); e += MR_TX * MR_TY) {
            const int ry = e / (MR_TX + 4), rx = e % (MR_TX + 4);
            int yy = y0 - 2 + ry, xx = x0 - 2 + rx;
            yy = mirror2(yy > Y + 1 ? Y + 1 : yy, Y);  // rows / columns of a partial tile beyond the volume are not used
            xx = mirror2(xx > X + 1 ? X + 1 : xx, X);
            tile[slot][ry][rx] = src[(zb + yy) * X + xx];
        }
    };
    for (int q = -2; q < 2; q++) load_slab(z0 + q);
    for (int z = z0; z < z0 + MR_ZB && z < Z; z++) {
        load_slab(z + 2);
        __syncthreads();
        if (live) {
            const long long t = ((long long)z * Y + y) * X + x;
            const float v = v32[(size_t)field * fstride + t];
            // pass 1, branch-free: how many window values round below v, how many round to v, and the last of those
            int lt = 0, ties = 0;
            double hit = 0.0;
#pragma unroll 1
            for (int a = 0; a < 5; a++) {
                const int slot = (z - 2 + a + 10) % 5;
#pragma unroll
                for (int b = 0; b < 5; b++)
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const double xv = tile[slot][ty + b][tx + c];
                        const float f = (float)xv;
                        lt += f < v ? 1 : 0;
                        const bool e = f == v;
                        ties += e ? 1 : 0;
                        hit = e ? xv : hit;
                    }
            }
            const int r = 62 - lt;  // rank of the median among the values that round to v
            double m;
            if (ties == 1) {
                m = hit;  // the usual case: exactly one window value rounds to v -- it is the median
            } else if (ties <= 0 || r < 0 || r >= ties || (dbg & 1)) {
                m = (double)v;  // NaN in the window: keep what the fp32 selection produced (dbg: timing experiment)
            } else {
                // several values inside one fp32 interval (flat regions; a few per cent of the voxels of a smooth field):
                // the r-th smallest of them, one pass per distinct value up to the wanted rank
                const double inf = __longlong_as_double(0x7ff0000000000000LL);
                double cur = -inf;
                int seen = 0;
                m = (double)v;
                for (int pass = 0; pass < 125; pass++) {
                    double nxt = inf;
                    int cnt = 0;
                    for (int a = 0; a < 5; a++) {
                        const int slot = (z - 2 + a + 10) % 5;
                        for (int b = 0; b < 5; b++)
#pragma unroll
                            for (int c = 0; c < 5; c++) {
                                const double xv = tile[slot][ty + b][tx + c];
                                if ((float)xv == v && xv > cur) {
                                    if (xv < nxt) {
                                        nxt = xv;
                                        cnt = 1;
                                    } else if (xv == nxt) {
                                        cnt++;
                                    }
                                }
                            }
                    }
                    if (cnt == 0) break;
                    if (r < seen + cnt) {
                        m = nxt;
                        break;
                    }
                    seen += cnt;
                    cur = nxt;
                }
            }
            out[t] = dst.acc ? (float)((double)out[t] + m) : (float)m;
        }
        __syncthreads();
    }
}

// level flow += increments without a median (levels of at most 5 voxels per axis): the same single rounding
__global__ void k_accum_round_once(float *__restrict__ u, const double *__restrict__ d, long long n)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) u[t] = (float)((double)u[t] + d[t]);
}
void launch_accum_round_once(hipStream_t st, float *u, const double *d, long long n)
{
    hipLaunchKernelGGL(k_accum_round_once, dim3(cdiv(n, 256)), dim3(256), 0, st, u, d, n);
    FR3D_LAUNCH_CHECK();
}

void launch_median5(hipStream_t st, const float *in, int Z, int Y, int X, float *out);

static bool median_flat_ok(int Z, int Y, int X, int nf)
{
    const long long n = (long long)Z * Y * X;
    return X >= 8 && Y >= 3 && Z >= 3 && n < (1ll << 32) && nf >= 1 && nf <= 3;
}

// A/B aid of the experiment build (-DFR3D_EXPERIMENTS): FR3D_MEDIAN = 1 one output per thread, 2 pairs without LDS,
// 3 row tiles.  The shipped library has no switch: mode 0.
static int median_mode()
{
#ifdef FR3D_EXPERIMENTS
    static const char *env = getenv("FR3D_MEDIAN");
    return env ? atoi(env) : 0;
#else
    return 0;
#endif
}

bool median_can_accumulate(int Z, int Y, int X)
{
    return median_mode() == 0 && median_flat_ok(Z, Y, X, 3);
}

// nf fields of one volume (field f at in + f*fstride) in one launch; out[f] = median, or out[f] += median.
void launch_median5_fields(hipStream_t st, const float *in, long long fstride, int nf, int Z, int Y, int X,
                           float *const *out, bool accumulate)
{
    const int mode = median_mode();
    if (mode == 0 && median_flat_ok(Z, Y, X, nf)) {
        MedianDst d;
        for (int f = 0; f < 3; f++) d.p[f] = f < nf ? out[f] : nullptr;
        d.acc = accumulate ? 1 : 0;
        const unsigned npairs = (unsigned)((long long)Z * Y * ((X + 1) / 2));
        hipLaunchKernelGGL(k_median5_flat<float>, dim3(cdiv((long long)npairs, MF_T), nf), dim3(MF_T), 0, st, in, fstride, Z, Y, X,
                           npairs, d);
        FR3D_LAUNCH_CHECK();
        return;
    }
    FR3D_CHECK(!accumulate, "internal: accumulating median needs the flattened kernel");
    for (int f = 0; f < nf; f++) launch_median5(st, in + (size_t)f * fstride, Z, Y, X, out[f]);
}

// Exact level tail (fp64 / packed solver modes): u[f] = RN32(u[f] + median64(in[f])) for three fp64 fields; `v32` is
// scratch for 3 * Z*Y*X floats.  Needs median_can_accumulate(Z, Y, X).
void launch_median5_fields_f64(hipStream_t st, const double *in, long long fstride, int Z, int Y, int X, float *v32,
                               float *const *u)
{
    FR3D_CHECK(median_flat_ok(Z, Y, X, 3), "internal: exact median tail needs the flattened kernel");
    MedianDst tmp, d;
    for (int f = 0; f < 3; f++) {
        tmp.p[f] = v32 + (size_t)f * fstride;
        d.p[f] = u[f];
    }
    tmp.acc = 0;
    d.acc = 1;
    const unsigned npairs = (unsigned)((long long)Z * Y * ((X + 1) / 2));
    hipLaunchKernelGGL(k_median5_flat<double>, dim3(cdiv((long long)npairs, MF_T), 3), dim3(MF_T), 0, st, in, fstride, Z, Y, X,
                       npairs, tmp);
    FR3D_LAUNCH_CHECK();
    const int nzb = cdiv(Z, MR_ZB);
    FR3D_CHECK(cdiv(Y, MR_TY) <= 65535 && 3 * nzb <= 65535, "exact median tail: axis too long");
    int dbg = 0;
#ifdef FR3D_EXPERIMENTS
    static const char *env = getenv("FR3D_MEDIAN_DBG");  // 1: skip the several-ties path (timing experiment)
    dbg = env ? atoi(env) : 0;
#endif
    hipLaunchKernelGGL(k_median5_refine, dim3(cdiv(X, MR_TX), cdiv(Y, MR_TY), 3 * nzb), dim3(MR_TX, MR_TY), 0, st, in, v32, fstride, Z, Y, X,
                       nzb, d, dbg);
    FR3D_LAUNCH_CHECK();
}

void launch_median5(hipStream_t st, const float *in, int Z, int Y, int X, float *out)
{
    const int mode = median_mode();
    if (mode == 0 && median_flat_ok(Z, Y, X, 1)) {
        float *o[1] = {out};
        launch_median5_fields(st, in, 0, 1, Z, Y, X, o, false);
    } else if (mode == 1) {
        long long total = (long long)Z * Y * X;
        hipLaunchKernelGGL(k_median5, dim3(cdiv(total, 256)), dim3(256), 0, st, in, Z, Y, X, out);
        FR3D_LAUNCH_CHECK();
    } else if (mode == 2 || Z > 65535 || cdiv(Y, MB_Y) > 65535) {
        long long total = (long long)Z * Y * ((X + 1) / 2);
        hipLaunchKernelGGL(k_median5_x2, dim3(cdiv(total, 256)), dim3(256), 0, st, in, Z, Y, X, out);
        FR3D_LAUNCH_CHECK();
    } else {
        dim3 grid(cdiv(X, 2 * MB_XP), cdiv(Y, MB_Y), Z);
        hipLaunchKernelGGL(k_median5_lds, grid, dim3(256), 0, st, in, Z, Y, X, out);
        FR3D_LAUNCH_CHECK();
    }
}

}  // namespace fr3d
