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

// in: element (t,z,y,x) at ((t*Z+z)*Y+y)*X+x) * cs + co ; out planar fp64
template <typename TIN>
__global__ void __launch_bounds__(256)
k_gauss_pass(const TIN *__restrict__ in, int cs, int co, double nmin, double nden, int T, int Z, int Y, int X,
             int axis, const double *__restrict__ w, int radius, double *__restrict__ out)
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
    auto at = [&](int i) -> double { return ((double)in[(size_t)(base + (long long)i * st) * cs + co] - nmin) / nden; };
    const double *fw = w + radius;
    double tmp = at(l) * fw[0];
    // four tap pairs per trip: their eight loads are independent and leave together; the sum keeps
    // NI_Correlate1D's order (outermost pair first)
    for (int j0 = -radius; j0 < 0; j0 += 4) {
        double lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jj = j0 + q < 0 ? j0 + q : -1;  // clamped: always a valid tap, discarded below
            lo[q] = at(reflect_hs(l + jj, n));
            hi[q] = at(reflect_hs(l - jj, n));
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (j0 + q < 0) tmp += (lo[q] + hi[q]) * fw[j0 + q];
    }
    out[e] = tmp;
}

template <typename TIN>
void launch_gauss_pass(hipStream_t st, const TIN *in, int cs, int co, double nmin, double nden, int T, int Z,
                       int Y, int X, int axis, const double *w, int radius, double *out)
{
    const long long total = (long long)T * Z * Y * X;
    if (total == 0) return;
    hipLaunchKernelGGL(k_gauss_pass<TIN>, dim3(cdiv(total, 256)), dim3(256), 0, st, in, cs, co, nmin, nden, T, Z, Y,
                       X, axis, w, radius, out);
    FR3D_LAUNCH_CHECK();
}
template void launch_gauss_pass<float>(hipStream_t, const float *, int, int, double, double, int, int, int, int, int, const double *, int, double *);
template void launch_gauss_pass<double>(hipStream_t, const double *, int, int, double, double, int, int, int, int, int, const double *, int, double *);
template void launch_gauss_pass<unsigned char>(hipStream_t, const unsigned char *, int, int, double, double, int, int, int, int, int, const double *, int, double *);
template void launch_gauss_pass<unsigned short>(hipStream_t, const unsigned short *, int, int, double, double, int, int, int, int, int, const double *, int, double *);
template void launch_gauss_pass<short>(hipStream_t, const short *, int, int, double, double, int, int, int, int, int, const double *, int, double *);

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
