// k_resize.hip -- K1: fused Gauss (x) Keys-cubic separable resampler.
//
// Restates _resize_x3d/_y3d/_z3d of util/resize_util_3D.py:8-50: one gather of P taps along one
// axis per output voxel, fp32 products accumulated in fp32 in tap order (no FMA contraction --
// the file is compiled with -ffp-contract=off) so the result is bit-identical to the CPU path.
// HBM-bound streaming: threads run along the innermost output axis (coalesced stores; the P taps
// of neighbouring outputs overlap, so the gathers are served by L1/L2).
#include <cstdlib>

#include "fr3d_internal.h"

namespace fr3d {

// axis 2: out (n0,n1,out_len); src rows of length n2.  A thread keeps the P taps of its output
// column in registers and walks RX_ROWS rows with them (the taps depend on the column only), so the
// table is read once per RX_ROWS outputs instead of once per output.  PT > 0: the tap count is a
// template constant (every pyramid step of eta = 0.8 needs <= 16 taps), so the gathers of two rows
// are issued back to back without tap-count branches; PT == 0: generic loop for longer kernels.
#define RX_ROWS 8
#define RX_MAXP 16
template <int PT>
__global__ void __launch_bounds__(256)
k_resize_x(const float *__restrict__ src, int cs, int co, long long rows, int n2, int out_len,
           const int *__restrict__ idx, const float *__restrict__ wt, int P, float *__restrict__ dst,
           int rpb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    const long long row0 = (long long)blockIdx.y * rpb;
    const int *ii = idx + (size_t)i * P;
    const float *ww = wt + (size_t)i * P;
    if constexpr (PT > 0) {
        int tap[PT];
        float w[PT];
#pragma unroll
        for (int p = 0; p < PT; p++) {
            tap[p] = ii[p] * cs;
            w[p] = ww[p];
        }
        long long q = 0;
        for (; q + 2 <= rpb && row0 + q + 2 <= rows; q += 2) {  // two rows per trip: 2*PT gathers in flight
            const float *r0 = src + (size_t)(row0 + q) * n2 * cs + co;
            const float *r1 = r0 + (size_t)n2 * cs;
            float v0[PT], v1[PT];
#pragma unroll
            for (int p = 0; p < PT; p++) {
                v0[p] = r0[tap[p]];
                v1[p] = r1[tap[p]];
            }
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int p = 0; p < PT; p++) {  // tap order and fp32 mul/add as the reference
                a0 += v0[p] * w[p];
                a1 += v1[p] * w[p];
            }
            dst[(size_t)(row0 + q) * out_len + i] = a0;
            dst[(size_t)(row0 + q + 1) * out_len + i] = a1;
        }
        for (; q < rpb && row0 + q < rows; q++) {
            const float *r = src + (size_t)(row0 + q) * n2 * cs + co;
            float a = 0.0f;
#pragma unroll
            for (int p = 0; p < PT; p++) a += r[tap[p]] * w[p];
            dst[(size_t)(row0 + q) * out_len + i] = a;
        }
    } else {
        for (int q = 0; q < rpb; q++) {
            const long long row = row0 + q;
            if (row >= rows) break;
            const float *r = src + (size_t)row * n2 * cs + co;
            float a = 0.0f;
            for (int p = 0; p < P; p++) a += r[(size_t)ii[p] * cs] * ww[p];
            dst[(size_t)row * out_len + i] = a;
        }
    }
}

// axis 2 on a planar source with rows of at most RXL_MAXN samples: a workgroup stages RXL_ROWS whole rows in LDS
// with one contiguous, fully coalesced read (consecutive rows are adjacent in memory) and gathers the taps there.
// The register form above keeps 2*PT dependent gathers per thread in flight and waits for them four times per
// workgroup; here every global load of the workgroup is issued at once.  Same sums in the same order.
#define RXL_ROWS 8
#define RXL_MAXN 1024
template <int PT>
__global__ void __launch_bounds__(256)
k_resize_x_lds(const float *__restrict__ src, long long rows, int n2, int out_len, const int *__restrict__ idx,
               const float *__restrict__ wt, float *__restrict__ dst)
{
    extern __shared__ float rowbuf[];  // RXL_ROWS x n2
    const long long row0 = (long long)blockIdx.x * RXL_ROWS;
    const int nr = (int)(rows - row0 < RXL_ROWS ? rows - row0 : RXL_ROWS);
    const long long total = (long long)nr * n2;
    const float *__restrict__ base = src + (size_t)row0 * n2;
    for (long long e = threadIdx.x; e < total; e += 256) rowbuf[e] = base[e];
    __syncthreads();
    for (int i = threadIdx.x; i < out_len; i += 256) {
        int tap[PT];
        float w[PT];
#pragma unroll
        for (int p = 0; p < PT; p++) {
            tap[p] = idx[(size_t)i * PT + p];
            w[p] = wt[(size_t)i * PT + p];
        }
        for (int r = 0; r < nr; r++) {
            const float *row = rowbuf + r * n2;
            float a = 0.0f;
#pragma unroll
            for (int p = 0; p < PT; p++) a += row[tap[p]] * w[p];  // tap order and fp32 mul/add as the reference
            dst[(size_t)(row0 + r) * out_len + i] = a;
        }
    }
}

// axis 1 or 0 on planar data: src viewed as (outer, n, inner), dst (outer, out_len, inner).
// blockIdx.y = output index along the axis, blockIdx.z = outer: the P taps and weights of a
// workgroup are wave-uniform (scalar loads), the P gathers are independent coalesced row reads.
#ifndef RM_NOUT
#define RM_NOUT 4
#endif
template <int PT>
__global__ void __launch_bounds__(256)
k_resize_mid(const float *__restrict__ src, int n, long long inner, int out_len,
             const int *__restrict__ idx, const float *__restrict__ wt, int P,
             float *__restrict__ dst)
{
    const long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= inner) return;
    const long long o = blockIdx.z;
    const float *base = src + (size_t)o * n * inner + x;
    if constexpr (PT > 0) {
        // RM_NOUT outputs along the axis per thread: RM_NOUT*PT independent row reads in flight (neighbouring outputs
        // share most of their taps, the repeated read of a row is an L1 hit)
        float v[RM_NOUT][PT];
#pragma unroll
        for (int q = 0; q < RM_NOUT; q++) {
            const int i = min(RM_NOUT * (int)blockIdx.y + q, out_len - 1);
#pragma unroll
            for (int p = 0; p < PT; p++) v[q][p] = base[(size_t)idx[(size_t)i * PT + p] * inner];
        }
#pragma unroll
        for (int q = 0; q < RM_NOUT; q++) {
            const int i = RM_NOUT * (int)blockIdx.y + q;
            if (i >= out_len) break;
            float a = 0.0f;
#pragma unroll
            for (int p = 0; p < PT; p++) a += v[q][p] * wt[(size_t)i * PT + p];  // tap order as the reference
            dst[((size_t)o * out_len + i) * inner + x] = a;
        }
    } else {
        const int i = blockIdx.y;
        const int *ii = idx + (size_t)i * P;
        const float *ww = wt + (size_t)i * P;
        float a = 0.0f;
        // taps in groups of 8: the gathers of a group are independent and leave together, the sum keeps
        // the reference's tap order (a one-tap-per-trip loop waits for every load in turn)
        for (int p0 = 0; p0 < P; p0 += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = p0 + q < P ? base[(size_t)ii[p0 + q] * inner] : 0.0f;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (p0 + q < P) a += v[q] * ww[p0 + q];
        }
        dst[((size_t)o * out_len + i) * inner + x] = a;
    }
}

// dispatch on the tap count: 1..16 as template constants, anything longer through the generic loop
#define FR3D_TAP_SWITCH(P, CALL)                                                                          \
    switch (P) {                                                                                           \
        case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   case 4: CALL(4); break;   \
        case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;   \
        case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
        case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
        default: CALL(0); break;                                                                           \
    }

void launch_resize_pass(hipStream_t st, const float *src, int cs, int co, int n0, int n1, int n2,
                        int axis, int out_len, const int *idx, const float *wt, int P, float *dst)
{
    if (axis == 2) {
        long long rows = (long long)n0 * n1;
        long long total = rows * out_len;
        if (total == 0) return;
        (void)total;
        static const char *xl_env = getenv("FR3D_RESIZE_XLDS");  // A/B aid: 0 = register form only
        if (cs == 1 && co == 0 && n2 <= RXL_MAXN && P >= 1 && P <= RX_MAXP && !(xl_env && xl_env[0] == '0')) {
            const dim3 grid((unsigned)cdiv(rows, RXL_ROWS));
            const size_t lds = (size_t)RXL_ROWS * n2 * sizeof(float);
#define FR3D_RXL(PT) hipLaunchKernelGGL((k_resize_x_lds<PT>), grid, dim3(256), lds, st, src, rows, n2, out_len, idx, wt, dst)
            switch (P) {
                case 1: FR3D_RXL(1); break;   case 2: FR3D_RXL(2); break;   case 3: FR3D_RXL(3); break;   case 4: FR3D_RXL(4); break;
                case 5: FR3D_RXL(5); break;   case 6: FR3D_RXL(6); break;   case 7: FR3D_RXL(7); break;   case 8: FR3D_RXL(8); break;
                case 9: FR3D_RXL(9); break;   case 10: FR3D_RXL(10); break; case 11: FR3D_RXL(11); break; case 12: FR3D_RXL(12); break;
                case 13: FR3D_RXL(13); break; case 14: FR3D_RXL(14); break; case 15: FR3D_RXL(15); break; default: FR3D_RXL(16); break;
            }
#undef FR3D_RXL
            FR3D_LAUNCH_CHECK();
            return;
        }
        int rpb = RX_ROWS;
        if (cdiv(rows, rpb) > 65535) rpb = cdiv(rows, 65535);
        dim3 grid(cdiv(out_len, 256), cdiv(rows, rpb));
#define FR3D_RX(PT) hipLaunchKernelGGL((k_resize_x<PT>), grid, dim3(256), 0, st, src, cs, co, rows, n2, out_len, idx, wt, P, dst, rpb)
        FR3D_TAP_SWITCH(P, FR3D_RX)
#undef FR3D_RX
    } else {
        FR3D_CHECK(cs == 1 && co == 0, "resize: y/z passes need planar input");
        long long outer = (axis == 1) ? n0 : 1;
        int n = (axis == 1) ? n1 : n0;
        long long inner = (axis == 1) ? n2 : (long long)n1 * n2;
        long long total = outer * out_len * inner;
        if (total == 0) return;
        FR3D_CHECK(out_len <= 65535 && outer <= 65535, "resize: axis length beyond the grid limits");
        const int bx = inner >= 256 ? 256 : cdiv(inner, 64) * 64;  // short rows: no idle waves
        // template tap counts (1..16): RM_NOUT outputs along the axis per thread; generic form: one
        const dim3 grid(cdiv(inner, bx), (P >= 1 && P <= RX_MAXP) ? cdiv(out_len, RM_NOUT) : out_len, (unsigned)outer);
#define FR3D_RM(PT) hipLaunchKernelGGL((k_resize_mid<PT>), grid, dim3(bx), 0, st, src, n, inner, out_len, idx, wt, P, dst)
        FR3D_TAP_SWITCH(P, FR3D_RM)
#undef FR3D_RM
    }
    FR3D_LAUNCH_CHECK();
}

}  // namespace fr3d
