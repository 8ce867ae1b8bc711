// k_warp.hip -- K2: backward warp = scipy.ndimage.map_coordinates(order in {1,3}, mode="nearest")
// as imregister_wrapper uses it (core/optical_flow_3d.py:22-74).
//
// order 3 follows SciPy 1.15 step by step, in fp64 like SciPy:
//   1. edge-pad the volume by 12 voxels           (_interpolation.py _prepad_for_spline_filter)
//   2. cubic B-spline prefilter along z, y, x     (ni_splines.c apply_filter: gain 6, pole
//      sqrt(3)-2, mode 'nearest' -> half-sample-symmetric initialisation, in-place quirk kept)
//   3. 4x4x4 gather with fp32 coordinates         (ni_interpolation.c NI_GeometricTransform)
// Each warp is done once per pyramid level, so it is kept reference-exact (fp64 coefficients)
// rather than minimal in bytes; the bandwidth-critical kernel of the path is the SOR sweep.
#include <cstdlib>

#include "fr3d_internal.h"

namespace fr3d {

// sqrt(3.0) - 2.0 AS SciPy EVALUATES IT (ni_splines.c get_filter_poles: double arithmetic on the rounded square root):
// two ulp from the correctly rounded value of the real number, -0.26794919243112270647 = -0x1.126145e9ecd56p-2, which the
// engine used until round 3 -- invisible in fp32 outputs except on one voxel in a few million, where the fp64 tap sum
// sits on a rounding boundary (found through the verification mode at 128x160x192)
#define SPL_POLE (-0x1.126145e9ecd58p-2)

__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// ---- 1. edge pad ------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_pad_edge(const T *__restrict__ src, int cs, int co, int Z, int Y, int X, int npad,
           double *__restrict__ dst)
{
    const int PZ = Z + 2 * npad, PY = Y + 2 * npad, PX = X + 2 * npad;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)PZ * PY * PX;
    if (t >= total) return;
    int x = (int)(t % PX);
    long long r = t / PX;
    int y = (int)(r % PY);
    int z = (int)(r / PY);
    int sz = clampi(z - npad, Z), sy = clampi(y - npad, Y), sx = clampi(x - npad, X);
    dst[t] = (double)src[(((size_t)sz * Y + sy) * X + sx) * cs + co];
}

template <typename T>
void launch_pad_edge(hipStream_t st, const T *src, int cs, int co, int Z, int Y, int X, int npad,
                     double *dst)
{
    long long total = (long long)(Z + 2 * npad) * (Y + 2 * npad) * (X + 2 * npad);
    hipLaunchKernelGGL(k_pad_edge<T>, dim3(cdiv(total, 256)), dim3(256), 0, st, src, cs, co, Z, Y,
                       X, npad, dst);
    FR3D_LAUNCH_CHECK();
}
template void launch_pad_edge<float>(hipStream_t, const float *, int, int, int, int, int, int, double *);
template void launch_pad_edge<double>(hipStream_t, const double *, int, int, int, int, int, int, double *);
template void launch_pad_edge<unsigned char>(hipStream_t, const unsigned char *, int, int, int, int, int, int, double *);
template void launch_pad_edge<unsigned short>(hipStream_t, const unsigned short *, int, int, int, int, int, int, double *);
template void launch_pad_edge<short>(hipStream_t, const short *, int, int, int, int, int, int, double *);

static double zpow(int n) { return pow(SPL_POLE, (double)n); }

// ---- 2. prefilter -----------------------------------------------------------------------------
// One line in place.  `c` points at element 0, consecutive elements are `stride` apart.
// Same operation order as the CPU restatement: gain folded into the first touch of each sample.
#define PF_CH 8  // samples fetched per trip: the loads of a trip are independent of the recursion
__device__ __forceinline__ void spline_line(double *c, int n, long long stride, double z_n)
{
    const double z = SPL_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    // causal initialisation (reflect), sum truncated after 64 terms (|z|^64 < 1e-36)
    const int lim = n < 64 ? n : 64;
    double x0 = c[0] * gain;
    double acc = x0 + z_n * (c[(long long)(n - 1) * stride] * gain);
    double z_i = z;
    for (int i0 = 1; i0 < lim; i0 += PF_CH) {
        double xi[PF_CH], xr[PF_CH];
#pragma unroll
        for (int q = 0; q < PF_CH; q++) {
            // clamped index, unconditional load: a guarded load becomes branch + load + wait per sample
            const int i = i0 + q < lim ? i0 + q : lim - 1;
            xi[q] = c[(long long)i * stride];
            xr[q] = c[(long long)(n - 1 - i) * stride];
        }
#pragma unroll
        for (int q = 0; q < PF_CH; q++) {
            const int i = i0 + q;
            if (i < lim) {
                const double a = xi[q] * gain;
                const double r = (i == n - 1) ? acc : xr[q] * gain;
                acc += z_i * (a + z_n * r);
                z_i *= z;
            }
        }
    }
    acc *= z / (1.0 - z_n * z_n);
    acc += x0;
    c[0] = acc;
    double prev = acc;
    // causal sweep, PF_CH samples in flight per trip (a thread walks its line alone, and a level has
    // too few lines to hide one dependent load per sample behind other waves)
    for (int i0 = 1; i0 < n; i0 += PF_CH) {
        double v[PF_CH];
#pragma unroll
        for (int q = 0; q < PF_CH; q++) v[q] = c[(long long)(i0 + q < n ? i0 + q : n - 1) * stride];
#pragma unroll
        for (int q = 0; q < PF_CH; q++)
            if (i0 + q < n) {
                double xi = v[q] * gain;
                xi += z * prev;
                c[(long long)(i0 + q) * stride] = xi;
                prev = xi;
            }
    }
    prev *= z / (z - 1.0);
    c[(long long)(n - 1) * stride] = prev;
    for (int i0 = n - 2; i0 >= 0; i0 -= PF_CH) {
        double v[PF_CH];
#pragma unroll
        for (int q = 0; q < PF_CH; q++) v[q] = c[(long long)(i0 - q >= 0 ? i0 - q : 0) * stride];
#pragma unroll
        for (int q = 0; q < PF_CH; q++)
            if (i0 - q >= 0) {
                const double w = z * (prev - v[q]);
                c[(long long)(i0 - q) * stride] = w;
                prev = w;
            }
    }
}

// lines enumerated as (outer, inner); element 0 of a line at outer*outer_stride + inner*inner_stride
__global__ void __launch_bounds__(256)
k_prefilter_lines(double *c, long long nlines, long long inner_n, long long inner_stride,
                  long long outer_stride, int n, long long stride, double z_n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    long long inner = t % inner_n, outer = t / inner_n;
    spline_line(c + outer * outer_stride + inner * inner_stride, n, stride, z_n);
}

// x axis, n > 64: one wave per 64 rows, rows staged through a 64x32 fp64 LDS tile so that global
// accesses are 256-B row segments while each lane walks its own row in LDS.
#define PF_TW 32
__global__ void __launch_bounds__(64)
k_prefilter_x_tiled(double *c, long long nrows, int n)
{
    __shared__ double tile[64][PF_TW + 1];
    const double z = SPL_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 64;
    const int ntiles = (n + PF_TW - 1) / PF_TW;
    const int lr = lane >> 5, lc = lane & 31;  // 2 rows x 32 columns per wave-load

    auto load_tile = [&](int t) {
        // 16 row segments in flight per group (a load-then-store loop waits for each load in turn)
        int col = t * PF_TW + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                long long row = row0 + 2 * (16 * g + q) + lr;
                if (row > nrows - 1) row = nrows - 1;  // clamped, unconditional loads
                v[q] = c[row * n + (col < n ? col : n - 1)];
            }
#pragma unroll
            for (int q = 0; q < 16; q++) tile[2 * (16 * g + q) + lr][lc] = v[q];
        }
    };
    auto store_tile = [&](int t) {
        int col = t * PF_TW + lc;
        for (int r = 0; r < 64; r += 2) {
            long long row = row0 + r + lr;
            if (row < nrows && col < n) c[row * n + col] = tile[r + lr][lc];
        }
    };

    // causal initialisation from the first 64 samples (two tiles)
    double x0 = 0.0, acc = 0.0, z_i = z;
    for (int t = 0; t < 2; t++) {
        __syncthreads();
        load_tile(t);
        __syncthreads();
        for (int q = 0; q < PF_TW; q++) {
            int i = t * PF_TW + q;
            double xi = tile[lane][q] * gain;
            if (i == 0) {
                x0 = xi;
                acc = xi;
            } else {
                acc += z_i * xi;
                z_i *= z;
            }
        }
    }
    acc *= z;  // z / (1 - z_n^2) with z_n^2 == 0 in fp64 for n > 64
    acc += x0;

    // causal sweep
    double prev = acc;
    for (int t = 0; t < ntiles; t++) {
        __syncthreads();
        load_tile(t);
        __syncthreads();
        int qn = n - t * PF_TW;
        if (qn > PF_TW) qn = PF_TW;
        for (int q = 0; q < qn; q++) {
            double xi;
            if (t == 0 && q == 0) {
                xi = acc;
            } else {
                xi = tile[lane][q] * gain;
                xi += z * prev;
            }
            tile[lane][q] = xi;
            prev = xi;
        }
        __syncthreads();
        store_tile(t);
    }
    // anticausal sweep
    for (int t = ntiles - 1; t >= 0; t--) {
        __syncthreads();
        load_tile(t);
        __syncthreads();
        int qn = n - t * PF_TW;
        if (qn > PF_TW) qn = PF_TW;
        for (int q = qn - 1; q >= 0; q--) {
            double v;
            if (t == ntiles - 1 && q == qn - 1) {
                v = tile[lane][q] * (z / (z - 1.0));
            } else {
                v = z * (prev - tile[lane][q]);
            }
            tile[lane][q] = v;
            prev = v;
        }
        __syncthreads();
        store_tile(t);
    }
}

void launch_prefilter3(hipStream_t st, double *c, int PZ, int PY, int PX)
{
    // axis 0 (z): lines over (y,x), stride PY*PX
    if (PZ > 1) {
        long long nl = (long long)PY * PX;
        hipLaunchKernelGGL(k_prefilter_lines, dim3(cdiv(nl, 256)), dim3(256), 0, st, c, nl, nl,
                           1LL, 0LL, PZ, (long long)PY * PX, zpow(PZ));
    }
    // axis 1 (y): lines over (z,x), stride PX
    if (PY > 1) {
        long long nl = (long long)PZ * PX;
        hipLaunchKernelGGL(k_prefilter_lines, dim3(cdiv(nl, 256)), dim3(256), 0, st, c, nl,
                           (long long)PX, 1LL, (long long)PY * PX, PY, (long long)PX, zpow(PY));
    }
    // axis 2 (x): contiguous lines
    if (PX > 1) {
        long long nl = (long long)PZ * PY;
        if (PX > 64) {
            hipLaunchKernelGGL(k_prefilter_x_tiled, dim3(cdiv(nl, 64)), dim3(64), 0, st, c, nl, PX);
        } else {
            hipLaunchKernelGGL(k_prefilter_lines, dim3(cdiv(nl, 256)), dim3(256), 0, st, c, nl, nl,
                               (long long)PX, 0LL, PX, 1LL, zpow(PX));
        }
    }
    FR3D_LAUNCH_CHECK();
}

// ---- 2b. prefilter without the stored pad -------------------------------------------------------
// SciPy pads the volume by 12 replicated voxels and filters the padded array (steps 1-2 above).  The gather
// clips its coordinates to [0, N-1] first (core/optical_flow_3d.py:47-50), so it only ever reads coefficients
// -1 .. N+1 of an axis: here every line is still filtered over its full padded length N+24, in the same
// operation order, but the 12 pad samples of each end are the line's edge value held in a register, the pad
// LINES (copies of the edge lines, filtered to copies of their results) are not computed at all, and only the
// coefficients -2 .. N+1 are stored: z pass raw volume -> (Z+4, Y, X), y pass -> (Z+4, Y+4, X), x pass ->
// (Z+4, Y+4, X+4).  Same coefficients bit for bit as the padded form on that range (the x pass leaves out
// the |pole|^n terms of the initialisation exactly like k_prefilter_x_tiled), 95 instead of 140 bytes of
// traffic per voxel at 256^3 and 0.80 of the coefficient footprint.  Needs N >= 41 per axis (the
// initialisation sum then never reaches the in-place quirk of short lines); smaller volumes take the padded path.
#define PFC_PAD 12   // SciPy's pad
#define PFC_KEEP 2   // stored pad
#define PFC_SKIP (PFC_PAD - PFC_KEEP)

#define PFC_CH 16  // samples per trip; the loads of the NEXT trip are issued before the recursion of this one
template <typename TS>
__device__ __forceinline__ void spline_line_compact(const TS *__restrict__ s, long long sst, int N, double *__restrict__ o,
                                                    long long ost, double z_n)
{
    const double z = SPL_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const int S = N + 2 * PFC_KEEP;
    auto in = [&](int i) {  // sample i of the padded line
        int k = i - PFC_PAD;
        k = k < 0 ? 0 : (k > N - 1 ? N - 1 : k);
        return (double)s[(long long)k * sst];
    };
    const int n = N + 2 * PFC_PAD;
    // causal initialisation over the first 64 padded samples (spline_line above, n > 64)
    const double x0 = in(0) * gain;
    double acc = x0 + z_n * (in(n - 1) * gain);
    double z_i = z;
    for (int i0 = 1; i0 < 64; i0 += PFC_CH) {
        double xi[PFC_CH], xr[PFC_CH];
#pragma unroll
        for (int q = 0; q < PFC_CH; q++) {
            const int i = i0 + q < 64 ? i0 + q : 63;
            xi[q] = in(i);
            xr[q] = in(n - 1 - i);
        }
#pragma unroll
        for (int q = 0; q < PFC_CH; q++)
            if (i0 + q < 64) {
                const double a = xi[q] * gain;
                const double r = xr[q] * gain;
                acc += z_i * (a + z_n * r);
                z_i *= z;
            }
    }
    acc *= z / (1.0 - z_n * z_n);
    acc += x0;
    double prev = acc;  // coefficient 0 of the padded line
    // causal sweep: padded samples 1 .. PFC_SKIP-1 are the edge value, not stored
#pragma unroll
    for (int i = 1; i < PFC_SKIP; i++) {
        double xi = x0;
        xi += z * prev;
        prev = xi;
    }
    // A thread walks its line alone and a level has about one wave of lines per SIMD, so nothing else hides
    // the memory latency of a trip: the next trip's samples are requested before this trip's recursion runs.
    double v[PFC_CH], vn[PFC_CH];
#pragma unroll
    for (int q = 0; q < PFC_CH; q++) v[q] = in((q < S ? q : S - 1) + PFC_SKIP);
    for (int j0 = 0; j0 < S; j0 += PFC_CH) {
        if (j0 + PFC_CH < S) {
#pragma unroll
            for (int q = 0; q < PFC_CH; q++) vn[q] = in((j0 + PFC_CH + q < S ? j0 + PFC_CH + q : S - 1) + PFC_SKIP);
        }
#pragma unroll
        for (int q = 0; q < PFC_CH; q++)
            if (j0 + q < S) {
                double xi = v[q] * gain;
                xi += z * prev;
                o[(long long)(j0 + q) * ost] = xi;
                prev = xi;
            }
#pragma unroll
        for (int q = 0; q < PFC_CH; q++) v[q] = vn[q];
    }
    double tail[PFC_SKIP];  // the last PFC_SKIP padded samples: edge value again, kept for the way back
    const double xN = in(n - 1) * gain;
#pragma unroll
    for (int q = 0; q < PFC_SKIP; q++) {
        double xi = xN;
        xi += z * prev;
        tail[q] = xi;
        prev = xi;
    }
    // anticausal sweep
    prev *= z / (z - 1.0);
#pragma unroll
    for (int q = PFC_SKIP - 2; q >= 0; q--) prev = z * (prev - tail[q]);
#pragma unroll
    for (int q = 0; q < PFC_CH; q++) v[q] = o[(long long)(S - 1 - q >= 0 ? S - 1 - q : 0) * ost];
    for (int j0 = S - 1; j0 >= 0; j0 -= PFC_CH) {
        if (j0 - PFC_CH >= 0) {
#pragma unroll
            for (int q = 0; q < PFC_CH; q++) vn[q] = o[(long long)(j0 - PFC_CH - q >= 0 ? j0 - PFC_CH - q : 0) * ost];
        }
#pragma unroll
        for (int q = 0; q < PFC_CH; q++)
            if (j0 - q >= 0) {
                const double w = z * (prev - v[q]);
                o[(long long)(j0 - q) * ost] = w;
                prev = w;
            }
#pragma unroll
        for (int q = 0; q < PFC_CH; q++) v[q] = vn[q];
    }
}

// lines enumerated as (outer, inner); sample k of a source line at (outer*so + inner*si + k*ss) * cs + co
template <typename TS>
__global__ void __launch_bounds__(64)
k_prefilter_lines_compact(const TS *__restrict__ src, int cs, int co, long long so, long long si, long long ss,
                          double *__restrict__ out, long long oo, long long oi, long long os, long long nlines,
                          long long inner_n, int N, double z_n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    const long long inner = t % inner_n, outer = t / inner_n;
    spline_line_compact<TS>(src + (outer * so + inner * si) * cs + co, ss * cs, N, out + outer * oo + inner * oi, os, z_n);
}

// x axis: rows (nrows, N) -> (nrows, N+4); one wave per 64 rows, each lane walks its own row in a 64x32 LDS
// tile (global accesses stay 256-B row segments).  The next tile's row segments are requested into registers
// before the lanes run the recursion over the current tile, so their latency overlaps it.  (One tile buffer:
// with two, 34 KB of LDS per workgroup allow four workgroups per CU and a 256^3 level needs 1057.)
__global__ void __launch_bounds__(64)
k_prefilter_x_compact(const double *__restrict__ src, long long nrows, int N, double *__restrict__ out)
{
    __shared__ double tile[64][PF_TW + 1];
    const double z = SPL_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const int S = N + 2 * PFC_KEEP;
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 64;
    const int lr = lane >> 5, lc = lane & 31;
    // rows 2q + lr of the tile; the last workgroup repeats its last row (clamped, unconditional loads)
    const long long rleft = nrows - 1 - row0 - lr;
    const int qmax = rleft >= 62 ? 31 : (int)(rleft < 0 ? 0 : rleft / 2);
    const long long rbase = rleft < 0 ? nrows - 1 : row0 + lr;

    // 32 row segments of a tile: element (2q + lr, lc) = p[row*len + clamp(col0 + lc, 0, len-1)]
    auto gload = [&](double (&v)[32], const double *__restrict__ p, int len, int col0) {
        int col = col0 + lc;
        col = col < 0 ? 0 : (col > len - 1 ? len - 1 : col);
        const double *__restrict__ b = p + rbase * len + col;
        const unsigned step = 2u * (unsigned)len;
#pragma unroll
        for (int q = 0; q < 32; q++) v[q] = b[(unsigned)(q < qmax ? q : qmax) * step];
    };
    auto to_lds = [&](const double (&v)[32]) {
#pragma unroll
        for (int q = 0; q < 32; q++) tile[2 * q + lr][lc] = v[q];
    };
    auto gstore = [&](int col0) {
        const int col = col0 + lc;
        if (col < S) {
            double *__restrict__ b = out + (row0 + lr) * S + col;
            const unsigned step = 2u * (unsigned)S;
#pragma unroll
            for (int q = 0; q < 32; q++)
                if (q <= qmax && rleft >= 0) b[(unsigned)q * step] = tile[2 * q + lr][lc];
        }
    };

    double g[32];
    // causal initialisation from the first 64 padded samples (two tiles; source column = padded index - 12)
    double x0 = 0.0, acc = 0.0, z_i = z;
    gload(g, src, N, -PFC_PAD);
    for (int t = 0; t < 2; t++) {
        to_lds(g);
        __syncthreads();
        if (t == 0) gload(g, src, N, PF_TW - PFC_PAD);
        else gload(g, src, N, -PFC_KEEP);  // first tile of the causal sweep
#pragma unroll
        for (int q = 0; q < PF_TW; q++) {
            const double xi = tile[lane][q] * gain;
            if (t == 0 && q == 0) {
                x0 = xi;
                acc = xi;
            } else {
                acc += z_i * xi;
                z_i *= z;
            }
        }
        __syncthreads();
    }
    acc *= z;
    acc += x0;
    double prev = acc;
#pragma unroll
    for (int i = 1; i < PFC_SKIP; i++) {
        double xi = x0;
        xi += z * prev;
        prev = xi;
    }
    const int ntiles = (S + PF_TW - 1) / PF_TW;
    for (int t = 0; t < ntiles; t++) {  // stored column j <-> source column j - 2
        to_lds(g);
        __syncthreads();
        if (t + 1 < ntiles) gload(g, src, N, (t + 1) * PF_TW - PFC_KEEP);
        const int qn = S - t * PF_TW;
        double c[PF_TW];
#pragma unroll
        for (int q = 0; q < PF_TW; q++) c[q] = tile[lane][q];
#pragma unroll
        for (int q = 0; q < PF_TW; q++) {
            double xi = c[q] * gain;
            xi += z * prev;
            c[q] = xi;
            prev = q < qn ? xi : prev;
        }
#pragma unroll
        for (int q = 0; q < PF_TW; q++) tile[lane][q] = c[q];
        __syncthreads();
        gstore(t * PF_TW);
        __syncthreads();
    }
    double tail[PFC_SKIP];
    {
        long long row = row0 + lane;
        if (row > nrows - 1) row = nrows - 1;
        const double xN = src[row * N + (N - 1)] * gain;
#pragma unroll
        for (int q = 0; q < PFC_SKIP; q++) {
            double xi = xN;
            xi += z * prev;
            tail[q] = xi;
            prev = xi;
        }
    }
    prev *= z / (z - 1.0);
#pragma unroll
    for (int q = PFC_SKIP - 2; q >= 0; q--) prev = z * (prev - tail[q]);
    gload(g, out, S, (ntiles - 1) * PF_TW);
    for (int t = ntiles - 1; t >= 0; t--) {
        to_lds(g);
        __syncthreads();
        if (t > 0) gload(g, out, S, (t - 1) * PF_TW);
        const int qn = S - t * PF_TW;
        double c[PF_TW];
#pragma unroll
        for (int q = 0; q < PF_TW; q++) c[q] = tile[lane][q];
#pragma unroll
        for (int q = PF_TW - 1; q >= 0; q--) {
            const double v = z * (prev - c[q]);
            c[q] = v;
            prev = q < qn ? v : prev;
        }
#pragma unroll
        for (int q = 0; q < PF_TW; q++) tile[lane][q] = c[q];
        __syncthreads();
        gstore(t * PF_TW);
        __syncthreads();
    }
}

bool prefilter_compact_ok(int Z, int Y, int X)
{
#ifdef FR3D_EXPERIMENTS
    static const char *env = getenv("FR3D_PREFILTER");  // A/B aid (experiment build): "padded" forces the stored-pad form
    if (env && env[0] == 'p') return false;
#endif
    return Z >= 41 && Y >= 41 && X >= 41;
}

// vol (Z,Y,X) with channel stride/offset -> coef (Z+4, Y+4, X+4); tmp holds (Z+4)(Y+4)X doubles
template <typename T>
void launch_prefilter3_compact(hipStream_t st, const T *vol, int cs, int co, int Z, int Y, int X, double *coef, double *tmp)
{
    const int SZ = Z + 2 * PFC_KEEP, SY = Y + 2 * PFC_KEEP;
    const long long yx = (long long)Y * X;
    {   // z: lines over (y,x); raw volume -> coef viewed as (SZ, Y, X)
        hipLaunchKernelGGL((k_prefilter_lines_compact<T>), dim3(cdiv(yx, 64)), dim3(64), 0, st, vol, cs, co, 0LL, 1LL, yx,
                           coef, 0LL, 1LL, yx, yx, yx, Z, zpow(Z + 2 * PFC_PAD));
        FR3D_LAUNCH_CHECK();
    }
    {   // y: lines over (z,x); (SZ, Y, X) -> tmp (SZ, SY, X)
        const long long nl = (long long)SZ * X;
        hipLaunchKernelGGL((k_prefilter_lines_compact<double>), dim3(cdiv(nl, 64)), dim3(64), 0, st, (const double *)coef, 1, 0,
                           yx, 1LL, (long long)X, tmp, (long long)SY * X, 1LL, (long long)X, nl, (long long)X, Y,
                           zpow(Y + 2 * PFC_PAD));
        FR3D_LAUNCH_CHECK();
    }
    {   // x: rows (SZ*SY, X) -> coef (SZ, SY, X+4)
        const long long nr = (long long)SZ * SY;
        hipLaunchKernelGGL(k_prefilter_x_compact, dim3(cdiv(nr, 64)), dim3(64), 0, st, (const double *)tmp, nr, X, coef);
        FR3D_LAUNCH_CHECK();
    }
}
template void launch_prefilter3_compact<float>(hipStream_t, const float *, int, int, int, int, int, double *, double *);
template void launch_prefilter3_compact<double>(hipStream_t, const double *, int, int, int, int, int, double *, double *);
template void launch_prefilter3_compact<unsigned char>(hipStream_t, const unsigned char *, int, int, int, int, int, double *, double *);
template void launch_prefilter3_compact<unsigned short>(hipStream_t, const unsigned short *, int, int, int, int, int, double *, double *);
template void launch_prefilter3_compact<short>(hipStream_t, const short *, int, int, int, int, int, double *, double *);

// ---- 3. gather --------------------------------------------------------------------------------
// ni_splines.c get_spline_interpolation_weights(), order 3
__device__ __forceinline__ void bspline3_weights(double cc, double *w, int *start)
{
    double fl = floor(cc);
    *start = (int)fl - 1;
    double x = cc - fl;
    double z = 1.0 - x;
    // x / 6 as a correctly rounded reciprocal product (div_by_const, fr3d_internal.h: 5 instructions instead of the
    // ~12 of an fp64 division; nine of them per voxel)
    constexpr double six = 6.0, sixth = 1.0 / 6.0;
    w[1] = div_by_const(x * x * (x - 2.0) * 3.0 + 4.0, six, sixth);
    w[2] = div_by_const(z * z * (z - 2.0) * 3.0 + 4.0, six, sixth);
    w[0] = div_by_const(z * z * z, six, sixth);
    double w3 = 1.0;
    w3 -= w[0];
    w3 -= w[1];
    w3 -= w[2];
    w[3] = w3;
}

// Output conversion.  The executors warp the RAW volume (parallelization/sequential_3d.py:153-170):
// scipy.ndimage.map_coordinates allocates its output in the INPUT's dtype, so the interpolated double
// goes through NI_GeometricTransform's output cast (ni_interpolation.c CASE_INTERP_OUT_*: unsigned
// t > 0 ? t + 0.5 : 0, clamped, truncated; signed t +- 0.5, clamped, truncated; float/double plain cast),
// then into the float32 `warped` array (core/optical_flow_3d.py:59-66) and into `registered` of the
// batch's dtype.  Out-of-bounds voxels take the reference value through float32 (:69-70) and NumPy's
// float -> integer cast (truncation).  Inside the pyramid everything is float (TO = float).
template <typename TO> struct OutCast;
template <> struct OutCast<float> {
    static __device__ __forceinline__ float interp(double t) { return (float)t; }
    static __device__ __forceinline__ float oob(float f) { return f; }
};
template <> struct OutCast<double> {
    static __device__ __forceinline__ double interp(double t) { return (double)(float)t; }
    static __device__ __forceinline__ double oob(float f) { return (double)f; }
};
template <> struct OutCast<unsigned char> {
    static __device__ __forceinline__ unsigned char interp(double t)
    {
        t = t > 0 ? t + 0.5 : 0;
        t = t > 255.0 ? 255.0 : t;
        return (unsigned char)t;
    }
    static __device__ __forceinline__ unsigned char oob(float f) { return (unsigned char)f; }
};
template <> struct OutCast<unsigned short> {
    static __device__ __forceinline__ unsigned short interp(double t)
    {
        t = t > 0 ? t + 0.5 : 0;
        t = t > 65535.0 ? 65535.0 : t;
        return (unsigned short)t;
    }
    static __device__ __forceinline__ unsigned short oob(float f) { return (unsigned short)f; }
};
template <> struct OutCast<short> {
    static __device__ __forceinline__ short interp(double t)
    {
        t = t > 0 ? t + 0.5 : t - 0.5;
        t = t > 32767.0 ? 32767.0 : t;
        t = t < -32768.0 ? -32768.0 : t;
        return (short)t;
    }
    static __device__ __forceinline__ short oob(float f) { return (short)f; }
};

#ifndef WARP_ZB
#define WARP_ZB 1  // z-taps whose 16 coefficients are in flight together
#endif
template <typename TF, typename TR, typename TO>
__global__ void __launch_bounds__(256)
k_warp_cubic(const double *__restrict__ coef, int npad, const TF *__restrict__ pu,
             const TF *__restrict__ pv, const TF *__restrict__ pw, int fs, double hx, double hy,
             double hz, double rhx, double rhy, double rhz, const TR *__restrict__ ref, int rcs, int rco, int Z, int Y, int X,
             TO *__restrict__ out, int ocs, int oco)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    // core/optical_flow_3d.py:32-34 : (grid + displacement).astype(float32)
    // displacement / h as a correctly rounded reciprocal product; an infinite displacement stays infinite (out of
    // bounds -> reference value) instead of turning into the NaN the residual steps would make of it
    auto over_h = [](double d, double h, double rh) {
        const double q = div_by_const(d, h, rh);
        return fabs(d) <= 1.7976931348623157e308 ? q : d * rh;
    };
    float mx = (float)((double)x + over_h((double)pu[(size_t)t * fs], hx, rhx));
    float my = (float)((double)y + over_h((double)pv[(size_t)t * fs], hy, rhy));
    float mz = (float)((double)z + over_h((double)pw[(size_t)t * fs], hz, rhz));
    bool oob = (mx < 0.0f) || (mx >= (float)X) || (my < 0.0f) || (my >= (float)Y) ||
               (mz < 0.0f) || (mz >= (float)Z);
    if (oob) {
        out[(size_t)t * ocs + oco] = OutCast<TO>::oob((float)ref[(size_t)t * rcs + rco]);
        return;
    }
    float cx = mx > (float)(X - 1) ? (float)(X - 1) : mx;
    float cy = my > (float)(Y - 1) ? (float)(Y - 1) : my;
    float cz = mz > (float)(Z - 1) ? (float)(Z - 1) : mz;
    const int PY = Y + 2 * npad, PX = X + 2 * npad, PZ = Z + 2 * npad;
    double wz[4], wy[4], wx[4];
    int sz, sy, sx;
    bspline3_weights((double)cz + npad, wz, &sz);
    bspline3_weights((double)cy + npad, wy, &sy);
    bspline3_weights((double)cx + npad, wx, &sx);
    // The taps are summed in SciPy's order (z, y, x innermost; coefficient * wz * wy * wx), but the 16
    // coefficients of a z-tap are fetched together first: written as load-multiply-add per tap the
    // compiler emitted 64 dependent load/wait pairs per voxel.
    double acc = 0.0;
    // (the range test also catches NaN displacements, whose start index is meaningless)
    if (sx >= 0 && sy >= 0 && sz >= 0 && sx + 3 < PX && sy + 3 < PY && sz + 3 < PZ) {
        // coordinates are clipped to [0, N-1] (above), so with a pad of >= 2 the 4^3 taps never leave the
        // padded grid: no per-tap clamping, the four x-taps of a row are consecutive doubles behind ONE
        // address (the clamped form spends more VALU work on 64 tap addresses than on the 256 fp64
        // operations of the interpolation itself)
        const double *p0 = coef + ((size_t)sz * PY + sy) * PX + sx;
        const size_t sy_ = (size_t)PX, sz_ = (size_t)PY * PX;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            double c[16];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double *row = p0 + a * sz_ + b * sy_;
#pragma unroll
                for (int e = 0; e < 4; e++) c[4 * b + e] = row[e];
            }
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    double cf = c[4 * b + e];
                    cf *= wz[a];
                    cf *= wy[b];
                    cf *= wx[e];
                    acc += cf;
                }
        }
    } else {
        int xi[4], yi[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xi[e] = clampi(sx + e, PX);
            yi[e] = clampi(sy + e, PY);
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double *slab = coef + (size_t)clampi(sz + a, PZ) * PY * PX;
            double c[16];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int e = 0; e < 4; e++) c[4 * b + e] = slab[(size_t)yi[b] * PX + xi[e]];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    double cf = c[4 * b + e];
                    cf *= wz[a];
                    cf *= wy[b];
                    cf *= wx[e];
                    acc += cf;
                }
        }
    }
    out[(size_t)t * ocs + oco] = OutCast<TO>::interp(acc);
}

template <typename TF, typename TR, typename TO>
void launch_warp_cubic(hipStream_t st, const double *coef, int npad, const TF *pu, const TF *pv,
                       const TF *pw, int fs, double hx, double hy, double hz, const TR *ref,
                       int rcs, int rco, int Z, int Y, int X, TO *out, int ocs, int oco)
{
    long long total = (long long)Z * Y * X;
    hipLaunchKernelGGL((k_warp_cubic<TF, TR, TO>), dim3(cdiv(total, 256)), dim3(256), 0, st, coef,
                       npad, pu, pv, pw, fs, hx, hy, hz, 1.0 / hx, 1.0 / hy, 1.0 / hz, ref, rcs, rco, Z, Y, X, out, ocs,
                       oco);
    FR3D_LAUNCH_CHECK();
}
#define FR3D_WARP_CUBIC_INST(TF, TR, TO)                                                                       \
    template void launch_warp_cubic<TF, TR, TO>(hipStream_t, const double *, int, const TF *, const TF *, const TF *, \
                                                int, double, double, double, const TR *, int, int, int, int, int,  \
                                                TO *, int, int);
FR3D_WARP_CUBIC_INST(float, float, float)
FR3D_WARP_CUBIC_INST(double, float, float)
FR3D_WARP_CUBIC_INST(float, double, float)
FR3D_WARP_CUBIC_INST(double, double, float)
// the executor's final warp of a raw volume: float32 flow, reference f32/f64, output in the raw dtype
FR3D_WARP_CUBIC_INST(float, float, double)
FR3D_WARP_CUBIC_INST(float, double, double)
FR3D_WARP_CUBIC_INST(float, float, unsigned char)
FR3D_WARP_CUBIC_INST(float, double, unsigned char)
FR3D_WARP_CUBIC_INST(float, float, unsigned short)
FR3D_WARP_CUBIC_INST(float, double, unsigned short)
FR3D_WARP_CUBIC_INST(float, float, short)
FR3D_WARP_CUBIC_INST(float, double, short)
#undef FR3D_WARP_CUBIC_INST

// order 1: no prefilter, no padding; taps clamped like mode="nearest"
template <typename TV, typename TF, typename TR, typename TO>
__global__ void __launch_bounds__(256)
k_warp_linear(const TV *__restrict__ vol, int vcs, int vco, const TF *__restrict__ pu,
              const TF *__restrict__ pv, const TF *__restrict__ pw, int fs,
              const TR *__restrict__ ref, int Z, int Y, int X, TO *__restrict__ out, int ocs,
              int oco)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    float mx = (float)((double)x + (double)pu[(size_t)t * fs]);
    float my = (float)((double)y + (double)pv[(size_t)t * fs]);
    float mz = (float)((double)z + (double)pw[(size_t)t * fs]);
    bool oob = (mx < 0.0f) || (mx >= (float)X) || (my < 0.0f) || (my >= (float)Y) ||
               (mz < 0.0f) || (mz >= (float)Z);
    if (oob) {
        out[(size_t)t * ocs + oco] = OutCast<TO>::oob((float)ref[(size_t)t * vcs + vco]);
        return;
    }
    double cc[3] = {(double)(mz > (float)(Z - 1) ? (float)(Z - 1) : mz),
                    (double)(my > (float)(Y - 1) ? (float)(Y - 1) : my),
                    (double)(mx > (float)(X - 1) ? (float)(X - 1) : mx)};
    int s[3];
    double w[3][2];
    for (int d = 0; d < 3; d++) {
        double fl = floor(cc[d]);
        s[d] = (int)fl;
        double f = cc[d] - fl;
        w[d][0] = 1.0 - f;
        w[d][1] = 1.0 - w[d][0];
    }
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        int zi = clampi(s[0] + a, Z);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            int yi = clampi(s[1] + b, Y);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                int xi = clampi(s[2] + e, X);
                double cf = (double)vol[(((size_t)zi * Y + yi) * X + xi) * vcs + vco];
                cf *= w[0][a];
                cf *= w[1][b];
                cf *= w[2][e];
                acc += cf;
            }
        }
    }
    out[(size_t)t * ocs + oco] = OutCast<TO>::interp(acc);
}

template <typename TV, typename TF, typename TR, typename TO>
void launch_warp_linear(hipStream_t st, const TV *vol, int vcs, int vco, const TF *pu,
                        const TF *pv, const TF *pw, int fs, const TR *ref, int Z, int Y, int X,
                        TO *out, int ocs, int oco)
{
    long long total = (long long)Z * Y * X;
    hipLaunchKernelGGL((k_warp_linear<TV, TF, TR, TO>), dim3(cdiv(total, 256)), dim3(256), 0, st, vol, vcs,
                       vco, pu, pv, pw, fs, ref, Z, Y, X, out, ocs, oco);
    FR3D_LAUNCH_CHECK();
}
#define FR3D_WARP_LINEAR_INST(TV, TF, TR, TO)                                                                    \
    template void launch_warp_linear<TV, TF, TR, TO>(hipStream_t, const TV *, int, int, const TF *, const TF *,  \
                                                     const TF *, int, const TR *, int, int, int, TO *, int, int);
FR3D_WARP_LINEAR_INST(float, float, float, float)
FR3D_WARP_LINEAR_INST(float, double, float, float)
FR3D_WARP_LINEAR_INST(double, float, double, float)
FR3D_WARP_LINEAR_INST(double, double, double, float)
// raw volumes (executor tail): float32 flow, reference f32/f64, output in the raw dtype
FR3D_WARP_LINEAR_INST(float, float, double, float)
FR3D_WARP_LINEAR_INST(double, float, float, float)
FR3D_WARP_LINEAR_INST(double, float, float, double)
FR3D_WARP_LINEAR_INST(double, float, double, double)
FR3D_WARP_LINEAR_INST(unsigned char, float, float, unsigned char)
FR3D_WARP_LINEAR_INST(unsigned char, float, double, unsigned char)
FR3D_WARP_LINEAR_INST(unsigned short, float, float, unsigned short)
FR3D_WARP_LINEAR_INST(unsigned short, float, double, unsigned short)
FR3D_WARP_LINEAR_INST(short, float, float, short)
FR3D_WARP_LINEAR_INST(short, float, double, short)
#undef FR3D_WARP_LINEAR_INST

}  // namespace fr3d
