// k_preproc.hip -- SURVEY section 8 row f-1: the preprocessing that runs on every batch right before
// the flow path (motion_correction/compensate_recording_3D.py:229-254): min-max normalisation
// (util/image_processing_3D.py:12-92) followed by scipy.ndimage.gaussian_filter(mode="reflect",
// truncate=4) per channel (:95-162), all in fp64 like the reference.
//
// One kernel = one separable pass along one axis of a planar (T,Z,Y,X) fp64 array; the first pass
// also converts the caller's dtype / channels-last layout and applies (x - min) / den.  Taps are
// summed exactly as SciPy's NI_Correlate1D does for symmetric kernels: centre first, then the
// pairs from the outermost inwards (bit-identical weights apart from libm-vs-NumPy exp, 1 ulp).
#include "fr3d_internal.h"

namespace fr3d {

__device__ __forceinline__ int reflect_hs(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}
// scipy.ndimage boundary modes (NI_ExtendLine) as an index map; -1 = outside (mode "constant", cval 0)
//   FR3D_BOUNDARY_REFLECT  d c b a | a b c d | d c b a      (also "grid-mirror")
//   FR3D_BOUNDARY_CONSTANT 0 0 0 0 | a b c d | 0 0 0 0      (also "grid-constant")
//   FR3D_BOUNDARY_NEAREST  a a a a | a b c d | d d d d
//   FR3D_BOUNDARY_MIRROR   d c b | a b c d | c b a
//   FR3D_BOUNDARY_WRAP     a b c d | a b c d | a b c d      (also "grid-wrap")
__device__ __forceinline__ int extend_index(int i, int n, int mode)
{
    if (i >= 0 && i < n) return i;
    switch (mode) {
        case FR3D_BOUNDARY_CONSTANT: return -1;
        case FR3D_BOUNDARY_NEAREST: return i < 0 ? 0 : n - 1;
        case FR3D_BOUNDARY_MIRROR: {
            if (n == 1) return 0;
            const int period = 2 * n - 2;
            i %= period;
            if (i < 0) i += period;
            return i < n ? i : period - i;
        }
        case FR3D_BOUNDARY_WRAP: {
            i %= n;
            return i < 0 ? i + n : i;
        }
        default: return reflect_hs(i, n);
    }
}

// in: element (t,z,y,x) at ((t*Z+z)*Y+y)*X+x) * cs + co ; out planar fp64
template <typename TIN>
__global__ void __launch_bounds__(256)
k_gauss_pass(const TIN *__restrict__ in, int cs, int co, double nmin, double nden, int T, int Z, int Y, int X,
             int axis, const double *__restrict__ w, int radius, int mode, double *__restrict__ out)
{
    const long long total = (long long)T * Z * Y * X;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int dims[4] = {T, Z, Y, X};
    const long long strides[4] = {(long long)Z * Y * X, (long long)Y * X, (long long)X, 1};
    const int n = dims[axis];
    const long long st = strides[axis];
    const int l = (int)((e / st) % n);
    const long long base = e - (long long)l * st;
    // (outside, mode "constant": the filter pads the NORMALISED array with cval = 0)
    auto at = [&](int i) -> double {
        const double x = ((double)in[(size_t)(base + (long long)(i < 0 ? 0 : i) * st) * cs + co] - nmin) / nden;
        return i < 0 ? 0.0 : x;
    };
    const double *fw = w + radius;
    double tmp = at(l) * fw[0];
    // four tap pairs per trip: their eight loads are independent and leave together; the sum keeps
    // NI_Correlate1D's order (outermost pair first)
    for (int j0 = -radius; j0 < 0; j0 += 4) {
        double lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jj = j0 + q < 0 ? j0 + q : -1;  // clamped: always a valid tap, discarded below
            lo[q] = at(extend_index(l + jj, n, mode));
            hi[q] = at(extend_index(l - jj, n, mode));
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (j0 + q < 0) tmp += (lo[q] + hi[q]) * fw[j0 + q];
    }
    out[e] = tmp;
}

template <typename TIN>
void launch_gauss_pass(hipStream_t st, const TIN *in, int cs, int co, double nmin, double nden, int T, int Z,
                       int Y, int X, int axis, const double *w, int radius, int mode, double *out)
{
    const long long total = (long long)T * Z * Y * X;
    if (total == 0) return;
    hipLaunchKernelGGL(k_gauss_pass<TIN>, dim3(cdiv(total, 256)), dim3(256), 0, st, in, cs, co, nmin, nden, T, Z, Y,
                       X, axis, w, radius, mode, out);
    FR3D_LAUNCH_CHECK();
}
template void launch_gauss_pass<float>(hipStream_t, const float *, int, int, double, double, int, int, int, int, int, const double *, int, int, double *);
template void launch_gauss_pass<double>(hipStream_t, const double *, int, int, double, double, int, int, int, int, int, const double *, int, int, double *);
template void launch_gauss_pass<unsigned char>(hipStream_t, const unsigned char *, int, int, double, double, int, int, int, int, int, const double *, int, int, double *);
template void launch_gauss_pass<unsigned short>(hipStream_t, const unsigned short *, int, int, double, double, int, int, int, int, int, const double *, int, int, double *);
template void launch_gauss_pass<short>(hipStream_t, const short *, int, int, double, double, int, int, int, int, int, const double *, int, int, double *);

// ---- the fast passes (round 4): radius 4 = sigma 1, the OFOptions default (OF_options_3D.py:171) ----------------------
// Same arithmetic, same order of the sum as k_gauss_pass (results are bit-identical); what changes is the work per
// output: the normalisation (an fp64 division) is applied once per LOADED element instead of once per tap, mirror
// indices need no modulo (|offset| <= radius < n), a thread of a strided pass (T, Z or Y axis) keeps a register window
// and emits GP_RO outputs along the axis (12 loads for 4 outputs instead of 36), the x pass stages its row segment in
// LDS, and the LAST pass of a channel writes the caller's channels-last array itself (no separate store pass).
#define GP_R 4   // radius of the fast kernels
#define GP_RO 4  // outputs per thread along a strided axis
__device__ __forceinline__ int reflect_near(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

template <typename TIN, bool NORM>
__device__ __forceinline__ double gp_load(const TIN *__restrict__ in, long long e, int cs, int co, double nmin, double nden)
{
    const double x = (double)in[(size_t)e * cs + co];
    return NORM ? (x - nmin) / nden : x;
}
// NI_Correlate1D's order for a symmetric kernel: centre, then the pairs from the outermost inwards
__device__ __forceinline__ double gp_sum(const double *win /* 2 GP_R + 1 values, centre at GP_R */, const double *fw)
{
    double tmp = win[GP_R] * fw[0];
#pragma unroll
    for (int j = -GP_R; j < 0; j++) tmp += (win[GP_R + j] + win[GP_R - j]) * fw[j];
    return tmp;
}

// axis with element stride `inner` >= 1 and length n: the array is (outer, n, inner); thread = (outer, block of GP_RO
// positions, inner index), inner fastest (coalesced)
template <typename TIN, typename TOUT, bool NORM>
__global__ void __launch_bounds__(256)
k_gauss_strided4(const TIN *__restrict__ in, int cs, int co, double nmin, double nden, long long outer, int n, long long inner,
                 const double *__restrict__ w, TOUT *__restrict__ out, int ocs, int oco)
{
    const long long nlb = (n + GP_RO - 1) / GP_RO;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= outer * nlb * inner) return;
    const long long x = e % inner, lb = (e / inner) % nlb, o = e / (inner * nlb);
    const long long base = o * (long long)n * inner + x;
    const int l0 = (int)lb * GP_RO;
    double fwv[GP_R + 1];
#pragma unroll
    for (int j = 0; j <= GP_R; j++) fwv[j] = w[j];  // w[0..GP_R]: taps -GP_R .. 0 (the kernel is symmetric)
    double win[GP_RO + 2 * GP_R];
#pragma unroll
    for (int t = 0; t < GP_RO + 2 * GP_R; t++) {
        int l = l0 - GP_R + t;
        l = l < n + GP_R ? l : n + GP_R - 1;  // outputs beyond the axis' end are not stored: keep their taps in range
        win[t] = gp_load<TIN, NORM>(in, base + (long long)reflect_near(l, n) * inner, cs, co, nmin, nden);
    }
#pragma unroll
    for (int q = 0; q < GP_RO; q++) {
        if (l0 + q >= n) break;
        double tmp = win[q + GP_R] * fwv[GP_R];
#pragma unroll
        for (int j = -GP_R; j < 0; j++) tmp += (win[q + GP_R + j] + win[q + GP_R - j]) * fwv[GP_R + j];
        out[(size_t)(base + (long long)(l0 + q) * inner) * ocs + oco] = (TOUT)tmp;
    }
}

// innermost axis (stride 1): a workgroup takes 256 consecutive outputs of one row through an LDS segment
template <typename TIN, typename TOUT, bool NORM>
__global__ void __launch_bounds__(256)
k_gauss_x4(const TIN *__restrict__ in, int cs, int co, double nmin, double nden, long long rows, int n,
           const double *__restrict__ w, TOUT *__restrict__ out, int ocs, int oco)
{
    __shared__ double seg[256 + 2 * GP_R];
    const int nseg = (n + 255) / 256;
    const long long row = blockIdx.x / nseg;
    const int x0 = (int)(blockIdx.x % nseg) * 256;
    const long long base = row * (long long)n;
    for (int t = threadIdx.x; t < 256 + 2 * GP_R; t += 256) {
        int l = x0 - GP_R + t;
        l = l < n + GP_R ? l : n + GP_R - 1;
        seg[t] = gp_load<TIN, NORM>(in, base + reflect_near(l, n), cs, co, nmin, nden);
    }
    __syncthreads();
    const int x = x0 + (int)threadIdx.x;
    if (x >= n) return;
    double tmp = seg[threadIdx.x + GP_R] * w[GP_R];
#pragma unroll
    for (int j = -GP_R; j < 0; j++) tmp += (seg[threadIdx.x + GP_R + j] + seg[threadIdx.x + GP_R - j]) * w[GP_R + j];
    out[(size_t)(base + x) * ocs + oco] = (TOUT)tmp;
}

// one separable pass with radius GP_R along `axis` of a (T,Z,Y,X) array; false: shape not covered (caller falls back)
template <typename TIN, typename TOUT, bool NORM>
bool launch_gauss_pass4(hipStream_t st, const TIN *in, int cs, int co, double nmin, double nden, int T, int Z, int Y, int X,
                        int axis, const double *w, int radius, TOUT *out, int ocs, int oco)
{
    const int dims[4] = {T, Z, Y, X};
    const int n = dims[axis];
    if (radius != GP_R || n <= GP_R) return false;
    const long long total = (long long)T * Z * Y * X;
    if (total == 0) return true;
    if (axis == 3) {
        const long long rows = total / n, nblk = rows * ((n + 255) / 256);
        if (nblk > 2147483647LL) return false;
        hipLaunchKernelGGL((k_gauss_x4<TIN, TOUT, NORM>), dim3((unsigned)nblk), dim3(256), 0, st, in, cs, co, nmin, nden, rows, n, w,
                           out, ocs, oco);
    } else {
        long long inner = 1;
        for (int a = axis + 1; a < 4; a++) inner *= dims[a];
        const long long outer = total / (inner * n), nthr = outer * ((n + GP_RO - 1) / GP_RO) * inner;
        hipLaunchKernelGGL((k_gauss_strided4<TIN, TOUT, NORM>), dim3(cdiv(nthr, 256)), dim3(256), 0, st, in, cs, co, nmin, nden,
                           outer, n, inner, w, out, ocs, oco);
    }
    FR3D_LAUNCH_CHECK();
    return true;
}
#define GP4_INST(TIN, TOUT, NORM)                                                                                        \
    template bool launch_gauss_pass4<TIN, TOUT, NORM>(hipStream_t, const TIN *, int, int, double, double, int, int, int, int, int, \
                                                      const double *, int, TOUT *, int, int);
#define GP4_RAW(TIN) GP4_INST(TIN, double, true) GP4_INST(TIN, float, true)
GP4_RAW(float) GP4_RAW(double) GP4_RAW(unsigned char) GP4_RAW(unsigned short) GP4_RAW(short)
GP4_INST(double, double, false) GP4_INST(double, float, false)

// planar fp64 (n) -> channel c of a channels-last array of TOUT
template <typename TOUT>
__global__ void __launch_bounds__(256)
k_store_channel(const double *__restrict__ in, long long n, int C, int c, TOUT *__restrict__ out)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[(size_t)e * C + c] = (TOUT)in[e];
}

template <typename TOUT>
void launch_store_channel(hipStream_t st, const double *in, long long n, int C, int c, TOUT *out)
{
    if (n > 0) hipLaunchKernelGGL(k_store_channel<TOUT>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, n, C, c, out);
    FR3D_LAUNCH_CHECK();
}
template void launch_store_channel<float>(hipStream_t, const double *, long long, int, int, float *);
template void launch_store_channel<double>(hipStream_t, const double *, long long, int, int, double *);

}  // namespace fr3d
