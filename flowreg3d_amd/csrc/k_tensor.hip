// k_tensor.hip -- K3: gradient-constancy motion tensor (core/optical_flow_3d.py:92-152).
//
// One thread per interior voxel; every derivative is recomputed from the two fp32 images with
// clamped reads (the symmetric pads / re-pads of :93-104 are clamped indices) in fp64 and in the
// reference's operation order, so the fp64 tensor is bit-identical to NumPy's before it is
// rounded once to the fp32 storage the solver streams.  Radius-2 footprint, served by L1/L2.
// Output either natural (Z,Y,X) or directly in the solver's skewed layout.
#include "fr3d_internal.h"

namespace fr3d {

struct Img {
    const float *p;
    int Z, Y, X;
    __device__ __forceinline__ double at(int z, int y, int x) const
    {
        z = z < 0 ? 0 : (z >= Z ? Z - 1 : z);
        y = y < 0 ? 0 : (y >= Y ? Y - 1 : y);
        x = x < 0 ? 0 : (x >= X ? X - 1 : x);
        return (double)p[((size_t)z * Y + y) * X + x];
    }
};

__device__ __forceinline__ int cl(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// Derivatives and normalisers of one voxel (core/optical_flow_3d.py:92-132), fp64, reference operation order.
struct TensorVox {
    double fxx, fyy, fzz, fxy, fxz, fyz, fxt, fyt, fzt, rx, ry, rz;
};
__device__ __forceinline__ TensorVox tensor_voxel(const Img &f1, const Img &f2, int z, int y, int x, double hz,
                                                  double hy, double hx)
{
    const int Z = f1.Z, Y = f1.Y, X = f1.X;
    const double tx = 2.0 * hx, ty = 2.0 * hy, tz = 2.0 * hz;
    // first derivatives at (zz,yy,xx), position clamped (symmetric re-pad of :101-104)
    auto FX = [&](int zz, int yy, int xx) {
        zz = cl(zz, Z); yy = cl(yy, Y); xx = cl(xx, X);
        double g1 = (f1.at(zz, yy, xx + 1) - f1.at(zz, yy, xx - 1)) / tx;
        double g2 = (f2.at(zz, yy, xx + 1) - f2.at(zz, yy, xx - 1)) / tx;
        return 0.5 * (g1 + g2);
    };
    auto FY = [&](int zz, int yy, int xx) {
        zz = cl(zz, Z); yy = cl(yy, Y); xx = cl(xx, X);
        double g1 = (f1.at(zz, yy + 1, xx) - f1.at(zz, yy - 1, xx)) / ty;
        double g2 = (f2.at(zz, yy + 1, xx) - f2.at(zz, yy - 1, xx)) / ty;
        return 0.5 * (g1 + g2);
    };
    auto FT = [&](int zz, int yy, int xx) { return f2.at(zz, yy, xx) - f1.at(zz, yy, xx); };
    TensorVox v;
    v.fxy = (FX(z, y + 1, x) - FX(z, y - 1, x)) / ty;
    v.fxz = (FX(z + 1, y, x) - FX(z - 1, y, x)) / tz;
    v.fyz = (FY(z + 1, y, x) - FY(z - 1, y, x)) / tz;
    v.fzt = (FT(z + 1, y, x) - FT(z - 1, y, x)) / tz;
    v.fyt = (FT(z, y + 1, x) - FT(z, y - 1, x)) / ty;
    v.fxt = (FT(z, y, x + 1) - FT(z, y, x - 1)) / tx;
    const double hx2 = hx * hx, hy2 = hy * hy, hz2 = hz * hz;
    double a0 = f1.at(z, y, x), b0 = f2.at(z, y, x);
    double fxx1 = (f1.at(z, y, x - 1) - 2.0 * a0 + f1.at(z, y, x + 1)) / hx2;
    double fxx2 = (f2.at(z, y, x - 1) - 2.0 * b0 + f2.at(z, y, x + 1)) / hx2;
    double fyy1 = (f1.at(z, y - 1, x) - 2.0 * a0 + f1.at(z, y + 1, x)) / hy2;
    double fyy2 = (f2.at(z, y - 1, x) - 2.0 * b0 + f2.at(z, y + 1, x)) / hy2;
    double fzz1 = (f1.at(z - 1, y, x) - 2.0 * a0 + f1.at(z + 1, y, x)) / hz2;
    double fzz2 = (f2.at(z - 1, y, x) - 2.0 * b0 + f2.at(z + 1, y, x)) / hz2;
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

template <typename TA>
__global__ void __launch_bounds__(256)
k_motion_tensor(Img f1, Img f2, double hz, double hy, double hx, float *J11, float *J22,
                float *J33, float *J44, float *J12, float *J13, float *J23, float *J14, float *J24,
                float *J34, TA *A, long long a_stride, int skewed, int Yp, long long plane)
{
    const int Z = f1.Z, Y = f1.Y, X = f1.X;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    const TensorVox v = tensor_voxel(f1, f2, z, y, x, hz, hy, hx);
    const double fxx = v.fxx, fyy = v.fyy, fzz = v.fzz, fxy = v.fxy, fxz = v.fxz, fyz = v.fyz, fxt = v.fxt, fyt = v.fyt,
                 fzt = v.fzt, rx = v.rx, ry = v.ry, rz = v.rz;

    size_t o = skewed ? (size_t)sk_index(X, Yp, plane, z, y, x) : (size_t)t;
    if (J11) {
    J11[o] = (float)(rx * (fxx * fxx) + ry * (fxy * fxy) + rz * (fxz * fxz));
    J22[o] = (float)(rx * (fxy * fxy) + ry * (fyy * fyy) + rz * (fyz * fyz));
    J33[o] = (float)(rx * (fxz * fxz) + ry * (fyz * fyz) + rz * (fzz * fzz));
    if (J44) J44[o] = (float)(rx * (fxt * fxt) + ry * (fyt * fyt) + rz * (fzt * fzt));
    J12[o] = (float)(rx * fxx * fxy + ry * fxy * fyy + rz * fxz * fyz);
    J13[o] = (float)(rx * fxx * fxz + ry * fxy * fyz + rz * fxz * fzz);
    J23[o] = (float)(rx * fxy * fxz + ry * fyy * fyz + rz * fyz * fzz);
    J14[o] = (float)(rx * fxx * fxt + ry * fxy * fyt + rz * fxz * fzt);
    J24[o] = (float)(rx * fxy * fxt + ry * fyy * fyt + rz * fyz * fzt);
    J34[o] = (float)(rx * fxz * fxt + ry * fyz * fyt + rz * fzz * fzt);
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
k_motion_tensor_rec(Img f1, Img f2, double hz, double hy, double hx, TA *__restrict__ dst, const Skew sk)
{
    __shared__ TA tile[12][TY][TPX + 2];  // pitch 34: a diagonal's elements fall into consecutive banks
    const int Y = f1.Y, X = f1.X;
    const int txn = (X + TPX - 1) / TPX;
    const int x0 = (blockIdx.x % txn) * TPX, y0 = (blockIdx.x / txn) * TY;
    const int z = blockIdx.y;
    const int lane = threadIdx.x % TPX, grp = threadIdx.x / TPX;
    for (int ly = grp; ly < TY; ly += 8) {
        const int y = y0 + ly, x = x0 + lane;
        if (y < Y && x < X) {
            const TensorVox v = tensor_voxel(f1, f2, z, y, x, hz, hy, hx);
            double a12[12];
            tensor_factors12(v, a12);
#pragma unroll
            for (int q = 0; q < 12; q++) tile[q][ly][lane] = (TA)a12[q];
        }
    }
    __syncthreads();
    constexpr int ND = TY + TPX - 1;
    for (int d = grp; d < ND; d += 8) {
        const int ly = lane, lx = d - lane;
        const int y = y0 + ly, x = x0 + lx;
        if (ly < TY && lx >= 0 && lx < TPX && y < Y && x < X) {
            TA *o = dst + (size_t)sk_index(sk, z, y, x) * 12;
#pragma unroll
            for (int q = 0; q < 12; q++) o[q] = tile[q][ly][lx];
        }
    }
}

template <typename TA>
void launch_motion_tensor_rec(hipStream_t st, const float *f1, const float *f2, double hz, double hy, double hx, TA *dst,
                              const Skew &sk)
{
    constexpr int TY = sizeof(TA) == 8 ? 16 : 32;  // 12 x TY x 34 values of LDS
    FR3D_CHECK(sk.Z <= 65535, "motion tensor: z axis longer than 65535");
    Img a{f1, sk.Z, sk.Y, sk.X}, b{f2, sk.Z, sk.Y, sk.X};
    dim3 grid(cdiv(sk.X, TPX) * cdiv(sk.Y, TY), sk.Z);
    hipLaunchKernelGGL((k_motion_tensor_rec<TA, TY>), grid, dim3(256), 0, st, a, b, hz, hy, hx, dst, sk);
    FR3D_LAUNCH_CHECK();
}
template void launch_motion_tensor_rec<float>(hipStream_t, const float *, const float *, double, double, double, float *,
                                              const Skew &);
template void launch_motion_tensor_rec<double>(hipStream_t, const float *, const float *, double, double, double, double *,
                                               const Skew &);

template <typename TA>
void launch_motion_tensor(hipStream_t st, const float *f1, const float *f2, int Z, int Y, int X,
                          double hz, double hy, double hx, float *const J[10], TA *A,
                          long long a_stride, const Skew *sk)
{
    long long total = (long long)Z * Y * X;
    Img a{f1, Z, Y, X}, b{f2, Z, Y, X};
    hipLaunchKernelGGL(k_motion_tensor<TA>, dim3(cdiv(total, 256)), dim3(256), 0, st, a, b, hz, hy, hx,
                       J[0], J[1], J[2], J[3], J[4], J[5], J[6], J[7], J[8], J[9], A, a_stride, sk ? 1 : 0,
                       sk ? sk->Yp : 0, sk ? sk->plane : 0LL);
    FR3D_LAUNCH_CHECK();
}

template void launch_motion_tensor<float>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                          double, float *const[10], float *, long long, const Skew *);
template void launch_motion_tensor<double>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                           double, float *const[10], double *, long long, const Skew *);

}  // namespace fr3d
