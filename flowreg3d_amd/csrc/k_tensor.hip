// k_tensor.hip -- K3: gradient-constancy motion tensor (core/optical_flow_3d.py:92-152).
//
// One thread per interior voxel; every derivative is recomputed from the two fp32 images with
// clamped reads (the symmetric pads / re-pads of :93-104 are clamped indices) in fp64 and in the
// reference's operation order, so the fp64 tensor is bit-identical to NumPy's before it is
// rounded once to the fp32 storage the solver streams.  Radius-2 footprint, served by L1/L2.
// Output either natural (Z,Y,X) or directly in the solver's skewed layout.
#include <cstdlib>

#include "fr3d_internal.h"
#include "k_sor_core.h"

namespace fr3d {

struct Img {
    const float *p;
    int Z, Y, X;
};

__device__ __forceinline__ int cl(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// Grid spacings of the level with their reciprocals (host side, IEEE division: correctly rounded).
struct TensorScale {
    double tx, ty, tz, hx2, hy2, hz2;        // 2h and h^2 per axis (core/optical_flow_3d.py:105-118)
    double rtx, rty, rtz, rhx2, rhy2, rhz2;  // RN(1/.) of the six divisors
    int unit;                                // hx = hy = hz = 1 (the full-resolution level): x/2 = 0.5x, x/1 = x, exactly
};
inline TensorScale tensor_scale(double hz, double hy, double hx)
{
    TensorScale t;
    t.tx = 2.0 * hx; t.ty = 2.0 * hy; t.tz = 2.0 * hz;
    t.hx2 = hx * hx; t.hy2 = hy * hy; t.hz2 = hz * hz;
    t.rtx = 1.0 / t.tx; t.rty = 1.0 / t.ty; t.rtz = 1.0 / t.tz;
    t.rhx2 = 1.0 / t.hx2; t.rhy2 = 1.0 / t.hy2; t.rhz2 = 1.0 / t.hz2;
    t.unit = hx == 1.0 && hy == 1.0 && hz == 1.0;
    return t;
}

// Division by a level constant as a correctly rounded reciprocal product (div_by_const, fr3d_internal.h): 24 of
// the 27 divisions per voxel divide by 2h or h^2.
template <bool UNIT>
__device__ __forceinline__ double divc(double a, double b, double y)
{
    if constexpr (UNIT) return a * y;  // y is 0.5 or 1: exact
    return div_by_const(a, b, y);
}

// The 19 samples of one image that the derivatives of one voxel read: centre, 6 face and 12 edge neighbours,
// every index clamped (the symmetric pads / re-pads of :93-104 are clamped indices).
struct Sten19 {
    double c, xm, xp, ym, yp, zm, zp;
    double ymxm, ymxp, ypxm, ypxp;  // (z, y-+1, x-+1)
    double zmxm, zmxp, zpxm, zpxp;  // (z-+1, y, x-+1)
    double zmym, zmyp, zpym, zpyp;  // (z-+1, y-+1, x)
};
struct StenIdx {
    size_t zb[3];     // slice bases of z-1, z, z+1 (clamped)
    unsigned yo[3];   // row offsets inside a slice of y-1, y, y+1 (clamped)
    unsigned xo[3];   // x-1, x, x+1 (clamped)
};
__device__ __forceinline__ StenIdx sten_index(int Z, int Y, int X, int z, int y, int x)
{
    StenIdx s;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        s.zb[q] = (size_t)cl(z + q - 1, Z) * ((size_t)Y * X);
        s.yo[q] = (unsigned)cl(y + q - 1, Y) * (unsigned)X;
        s.xo[q] = (unsigned)cl(x + q - 1, X);
    }
    return s;
}
__device__ __forceinline__ Sten19 sten_fetch(const float *__restrict__ p, const StenIdx &i)
{
    auto at = [&](int zi, int yi, int xi) { return (double)p[i.zb[zi] + (i.yo[yi] + i.xo[xi])]; };
    Sten19 s;
    s.c = at(1, 1, 1);
    s.xm = at(1, 1, 0); s.xp = at(1, 1, 2);
    s.ym = at(1, 0, 1); s.yp = at(1, 2, 1);
    s.zm = at(0, 1, 1); s.zp = at(2, 1, 1);
    s.ymxm = at(1, 0, 0); s.ymxp = at(1, 0, 2); s.ypxm = at(1, 2, 0); s.ypxp = at(1, 2, 2);
    s.zmxm = at(0, 1, 0); s.zmxp = at(0, 1, 2); s.zpxm = at(2, 1, 0); s.zpxp = at(2, 1, 2);
    s.zmym = at(0, 0, 1); s.zmyp = at(0, 2, 1); s.zpym = at(2, 0, 1); s.zpyp = at(2, 2, 1);
    return s;
}

// Derivatives and normalisers of one voxel (core/optical_flow_3d.py:92-132), fp64, reference operation order.
struct TensorVox {
    double fxx, fyy, fzz, fxy, fxz, fyz, fxt, fyt, fzt, rx, ry, rz;
};
template <bool UNIT>
__device__ __forceinline__ TensorVox tensor_voxel_t(const Img &f1, const Img &f2, int z, int y, int x, const TensorScale &h)
{
    const StenIdx si = sten_index(f1.Z, f1.Y, f1.X, z, y, x);
    const Sten19 a = sten_fetch(f1.p, si), b = sten_fetch(f2.p, si);
    // first derivatives of the averaged image pair at a neighbour: 0.5 * (d f1 + d f2), each a central
    // difference over 2h (:105-110); the neighbour's position is clamped like the samples
    auto avg_d = [&](double a_hi, double a_lo, double b_hi, double b_lo, double t, double rt) {
        const double g1 = divc<UNIT>(a_hi - a_lo, t, rt);
        const double g2 = divc<UNIT>(b_hi - b_lo, t, rt);
        return 0.5 * (g1 + g2);
    };
    TensorVox v;
    // fx at (z, y+-1, x), (z+-1, y, x); fy at (z+-1, y, x); ft = f2 - f1 at the six face neighbours
    const double fx_yp = avg_d(a.ypxp, a.ypxm, b.ypxp, b.ypxm, h.tx, h.rtx);
    const double fx_ym = avg_d(a.ymxp, a.ymxm, b.ymxp, b.ymxm, h.tx, h.rtx);
    const double fx_zp = avg_d(a.zpxp, a.zpxm, b.zpxp, b.zpxm, h.tx, h.rtx);
    const double fx_zm = avg_d(a.zmxp, a.zmxm, b.zmxp, b.zmxm, h.tx, h.rtx);
    const double fy_zp = avg_d(a.zpyp, a.zpym, b.zpyp, b.zpym, h.ty, h.rty);
    const double fy_zm = avg_d(a.zmyp, a.zmym, b.zmyp, b.zmym, h.ty, h.rty);
    v.fxy = divc<UNIT>(fx_yp - fx_ym, h.ty, h.rty);
    v.fxz = divc<UNIT>(fx_zp - fx_zm, h.tz, h.rtz);
    v.fyz = divc<UNIT>(fy_zp - fy_zm, h.tz, h.rtz);
    v.fzt = divc<UNIT>((b.zp - a.zp) - (b.zm - a.zm), h.tz, h.rtz);
    v.fyt = divc<UNIT>((b.yp - a.yp) - (b.ym - a.ym), h.ty, h.rty);
    v.fxt = divc<UNIT>((b.xp - a.xp) - (b.xm - a.xm), h.tx, h.rtx);
    const double fxx1 = divc<UNIT>(a.xm - 2.0 * a.c + a.xp, h.hx2, h.rhx2);
    const double fxx2 = divc<UNIT>(b.xm - 2.0 * b.c + b.xp, h.hx2, h.rhx2);
    const double fyy1 = divc<UNIT>(a.ym - 2.0 * a.c + a.yp, h.hy2, h.rhy2);
    const double fyy2 = divc<UNIT>(b.ym - 2.0 * b.c + b.yp, h.hy2, h.rhy2);
    const double fzz1 = divc<UNIT>(a.zm - 2.0 * a.c + a.zp, h.hz2, h.rhz2);
    const double fzz2 = divc<UNIT>(b.zm - 2.0 * b.c + b.zp, h.hz2, h.rhz2);
    v.fxx = 0.5 * (fxx1 + fxx2);
    v.fyy = 0.5 * (fyy1 + fyy2);
    v.fzz = 0.5 * (fzz1 + fzz2);
    double sxn = sqrt(v.fxx * v.fxx + v.fxy * v.fxy + v.fxz * v.fxz);
    double syn = sqrt(v.fxy * v.fxy + v.fyy * v.fyy + v.fyz * v.fyz);
    double szn = sqrt(v.fxz * v.fxz + v.fyz * v.fyz + v.fzz * v.fzz);
    v.rx = 1.0 / (sxn * sxn + 1e-6);
    v.ry = 1.0 / (syn * syn + 1e-6);
    v.rz = 1.0 / (szn * szn + 1e-6);
    return v;
}
__device__ __forceinline__ TensorVox tensor_voxel(const Img &f1, const Img &f2, int z, int y, int x, const TensorScale &h)
{
    return h.unit ? tensor_voxel_t<true>(f1, f2, z, y, x, h) : tensor_voxel_t<false>(f1, f2, z, y, x, h);
}
// square-root factors: J = sum_k a_k a_k^T with a_k = sqrt(reg_k) * (f_kx, f_ky, f_kz, f_kt).
// psi_data is evaluated from these (sum of three squared residuals) because the expanded
// quadratic form cancels catastrophically once J is rounded to fp32 (numerics).
__device__ __forceinline__ void tensor_factors12(const TensorVox &v, double (&A)[12])
{
    const double qx = sqrt(v.rx), qy = sqrt(v.ry), qz = sqrt(v.rz);
    A[0] = qx * v.fxx; A[1] = qx * v.fxy; A[2] = qx * v.fxz; A[3] = qx * v.fxt;
    A[4] = qy * v.fxy; A[5] = qy * v.fyy; A[6] = qy * v.fyz; A[7] = qy * v.fyt;
    A[8] = qz * v.fxz; A[9] = qz * v.fyz; A[10] = qz * v.fzz; A[11] = qz * v.fzt;
}

template <typename TA, typename TJ>
__global__ void __launch_bounds__(256)
k_motion_tensor(Img f1, Img f2, TensorScale hs, TJ *J11, TJ *J22,
                TJ *J33, TJ *J44, TJ *J12, TJ *J13, TJ *J23, TJ *J14, TJ *J24,
                TJ *J34, TA *A, long long a_stride, int skewed, int Yp, long long plane)
{
    const int Z = f1.Z, Y = f1.Y, X = f1.X;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    const TensorVox v = tensor_voxel(f1, f2, z, y, x, hs);
    const double fxx = v.fxx, fyy = v.fyy, fzz = v.fzz, fxy = v.fxy, fxz = v.fxz, fyz = v.fyz, fxt = v.fxt, fyt = v.fyt,
                 fzt = v.fzt, rx = v.rx, ry = v.ry, rz = v.rz;

    size_t o = skewed ? (size_t)sk_index(X, Yp, plane, z, y, x) : (size_t)t;
    if (J11) {
    J11[o] = (TJ)(rx * (fxx * fxx) + ry * (fxy * fxy) + rz * (fxz * fxz));
    J22[o] = (TJ)(rx * (fxy * fxy) + ry * (fyy * fyy) + rz * (fyz * fyz));
    J33[o] = (TJ)(rx * (fxz * fxz) + ry * (fyz * fyz) + rz * (fzz * fzz));
    if (J44) J44[o] = (TJ)(rx * (fxt * fxt) + ry * (fyt * fyt) + rz * (fzt * fzt));
    J12[o] = (TJ)(rx * fxx * fxy + ry * fxy * fyy + rz * fxz * fyz);
    J13[o] = (TJ)(rx * fxx * fxz + ry * fxy * fyz + rz * fxz * fzz);
    J23[o] = (TJ)(rx * fxy * fxz + ry * fyy * fyz + rz * fyz * fzz);
    J14[o] = (TJ)(rx * fxx * fxt + ry * fxy * fyt + rz * fxz * fzt);
    J24[o] = (TJ)(rx * fxy * fxt + ry * fyy * fyt + rz * fyz * fzt);
    J34[o] = (TJ)(rx * fxz * fxt + ry * fyz * fyt + rz * fzz * fzt);
    }
    if (A) {
        double a12[12];
        tensor_factors12(v, a12);
#pragma unroll
        for (int q = 0; q < 12; q++) A[q * a_stride + o] = (TA)a12[q];
    }
}

// The same factors written straight into the solver's record layout (12 values per voxel, skewed voxel
// order, compact or pitched rows): a workgroup computes a TY x 32 tile of one z-slice in the natural order
// (coalesced image reads), parks the 12 values per voxel in LDS and stores whole records along the tile's
// anti-diagonals (x + y constant = one row of the skewed layout, consecutive j = consecutive records).
// Replaces "12 natural arrays out, transposing copy in": 96 B per voxel less traffic.
#define TPX 32
template <typename TA, int TY>
__global__ void __launch_bounds__(256)
k_motion_tensor_rec(Img f1, Img f2, TensorScale hs, TA *__restrict__ dst, const Skew sk, int dbg)
{
    __shared__ typename Sto<TA>::val tile[12][TY][TPX + 2];  // pitch 34: a diagonal's elements fall into consecutive banks
    const int Y = f1.Y, X = f1.X;
    const int txn = (X + TPX - 1) / TPX;
    const int x0 = (blockIdx.x % txn) * TPX, y0 = (blockIdx.x / txn) * TY;
    const int z = blockIdx.y;
    const int lane = threadIdx.x % TPX, grp = threadIdx.x / TPX;
    for (int ly = grp; ly < TY; ly += 8) {
        const int y = y0 + ly, x = x0 + lane;
        if (y < Y && x < X && !(dbg & 1)) {
            const TensorVox v = tensor_voxel(f1, f2, z, y, x, hs);
            double a12[12];
            tensor_factors12(v, a12);
#pragma unroll
            for (int q = 0; q < 12; q++) tile[q][ly][lane] = Sto<TA>::quant(a12[q]);
        }
    }
    __syncthreads();
    for (int m = grp; m < TY; m += 8) {
        int ly, lx;
        tile_diag<TY>(lane, m, ly, lx);
        const int y = y0 + ly, x = x0 + lx;
        if (y < Y && x < X && !(dbg & 2)) {
            Rec<TA, 12> o;
#pragma unroll
            for (int q = 0; q < 12; q++) o.v[q] = tile[q][ly][lx];
            strec<TA, 12>(dst, sk_index(sk, z, y, x), o);
        }
    }
}

template <typename TA>
void launch_motion_tensor_rec(hipStream_t st, const float *f1, const float *f2, double hz, double hy, double hx, TA *dst,
                              const Skew &sk)
{
    constexpr int TY = sizeof(typename Sto<TA>::val) == 8 ? 16 : 32;  // 12 x TY x 34 values of LDS
    FR3D_CHECK(sk.Z <= 65535, "motion tensor: z axis longer than 65535");
    Img a{f1, sk.Z, sk.Y, sk.X}, b{f2, sk.Z, sk.Y, sk.X};
    int dbg = 0;
#ifdef FR3D_EXPERIMENTS
    static const char *env = getenv("FR3D_TENSOR_DBG");  // disable the arithmetic (1) / the stores (2), 16-row tiles (4)
    dbg = env ? atoi(env) : 0;
#endif
    if (dbg & 4) {
        dim3 grid(cdiv(sk.X, TPX) * cdiv(sk.Y, 16), sk.Z);
        hipLaunchKernelGGL((k_motion_tensor_rec<TA, 16>), grid, dim3(256), 0, st, a, b, tensor_scale(hz, hy, hx), dst, sk, dbg);
        return;
    }
    dim3 grid(cdiv(sk.X, TPX) * cdiv(sk.Y, TY), sk.Z);
    hipLaunchKernelGGL((k_motion_tensor_rec<TA, TY>), grid, dim3(256), 0, st, a, b, tensor_scale(hz, hy, hx), dst, sk, dbg);
    FR3D_LAUNCH_CHECK();
}
template void launch_motion_tensor_rec<float>(hipStream_t, const float *, const float *, double, double, double, float *,
                                              const Skew &);
template void launch_motion_tensor_rec<double>(hipStream_t, const float *, const float *, double, double, double, double *,
                                               const Skew &);
template void launch_motion_tensor_rec<pk42>(hipStream_t, const float *, const float *, double, double, double, pk42 *,
                                             const Skew &);

template <typename TA, typename TJ>
void launch_motion_tensor(hipStream_t st, const float *f1, const float *f2, int Z, int Y, int X,
                          double hz, double hy, double hx, TJ *const J[10], TA *A,
                          long long a_stride, const Skew *sk)
{
    long long total = (long long)Z * Y * X;
    Img a{f1, Z, Y, X}, b{f2, Z, Y, X};
    hipLaunchKernelGGL((k_motion_tensor<TA, TJ>), dim3(cdiv(total, 256)), dim3(256), 0, st, a, b, tensor_scale(hz, hy, hx),
                       J[0], J[1], J[2], J[3], J[4], J[5], J[6], J[7], J[8], J[9], A, a_stride, sk ? 1 : 0,
                       sk ? sk->Yp : 0, sk ? sk->plane : 0LL);
    FR3D_LAUNCH_CHECK();
}

template void launch_motion_tensor<float, float>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                                 double, float *const[10], float *, long long, const Skew *);
template void launch_motion_tensor<double, float>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                                  double, float *const[10], double *, long long, const Skew *);
// verification mode: the fp64 tensor entries as the reference forms them (no rounding to storage)
template void launch_motion_tensor<double, double>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                                   double, double *const[10], double *, long long, const Skew *);

}  // namespace fr3d
