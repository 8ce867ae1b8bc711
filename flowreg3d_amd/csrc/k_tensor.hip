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

    double fxy = (FX(z, y + 1, x) - FX(z, y - 1, x)) / ty;
    double fxz = (FX(z + 1, y, x) - FX(z - 1, y, x)) / tz;
    double fyz = (FY(z + 1, y, x) - FY(z - 1, y, x)) / tz;
    double fzt = (FT(z + 1, y, x) - FT(z - 1, y, x)) / tz;
    double fyt = (FT(z, y + 1, x) - FT(z, y - 1, x)) / ty;
    double fxt = (FT(z, y, x + 1) - FT(z, y, x - 1)) / tx;

    const double hx2 = hx * hx, hy2 = hy * hy, hz2 = hz * hz;
    double a0 = f1.at(z, y, x), b0 = f2.at(z, y, x);
    double fxx1 = (f1.at(z, y, x - 1) - 2.0 * a0 + f1.at(z, y, x + 1)) / hx2;
    double fxx2 = (f2.at(z, y, x - 1) - 2.0 * b0 + f2.at(z, y, x + 1)) / hx2;
    double fyy1 = (f1.at(z, y - 1, x) - 2.0 * a0 + f1.at(z, y + 1, x)) / hy2;
    double fyy2 = (f2.at(z, y - 1, x) - 2.0 * b0 + f2.at(z, y + 1, x)) / hy2;
    double fzz1 = (f1.at(z - 1, y, x) - 2.0 * a0 + f1.at(z + 1, y, x)) / hz2;
    double fzz2 = (f2.at(z - 1, y, x) - 2.0 * b0 + f2.at(z + 1, y, x)) / hz2;
    double fxx = 0.5 * (fxx1 + fxx2);
    double fyy = 0.5 * (fyy1 + fyy2);
    double fzz = 0.5 * (fzz1 + fzz2);

    double sxn = sqrt(fxx * fxx + fxy * fxy + fxz * fxz);
    double syn = sqrt(fxy * fxy + fyy * fyy + fyz * fyz);
    double szn = sqrt(fxz * fxz + fyz * fyz + fzz * fzz);
    double rx = 1.0 / (sxn * sxn + 1e-6);
    double ry = 1.0 / (syn * syn + 1e-6);
    double rz = 1.0 / (szn * szn + 1e-6);

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
        // square-root factors: J = sum_k a_k a_k^T with a_k = sqrt(reg_k) * (f_kx, f_ky, f_kz, f_kt).
        // psi_data is evaluated from these (sum of three squared residuals) because the expanded
        // quadratic form cancels catastrophically once J is rounded to fp32 (numerics).
        const double qx = sqrt(rx), qy = sqrt(ry), qz = sqrt(rz);
        A[0 * a_stride + o] = (TA)(qx * fxx);
        A[1 * a_stride + o] = (TA)(qx * fxy);
        A[2 * a_stride + o] = (TA)(qx * fxz);
        A[3 * a_stride + o] = (TA)(qx * fxt);
        A[4 * a_stride + o] = (TA)(qy * fxy);
        A[5 * a_stride + o] = (TA)(qy * fyy);
        A[6 * a_stride + o] = (TA)(qy * fyz);
        A[7 * a_stride + o] = (TA)(qy * fyt);
        A[8 * a_stride + o] = (TA)(qz * fxz);
        A[9 * a_stride + o] = (TA)(qz * fyz);
        A[10 * a_stride + o] = (TA)(qz * fzz);
        A[11 * a_stride + o] = (TA)(qz * fzt);
    }
}

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
}

template void launch_motion_tensor<float>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                          double, float *const[10], float *, long long, const Skew *);
template void launch_motion_tensor<double>(hipStream_t, const float *, const float *, int, int, int, double, double,
                                           double, float *const[10], double *, long long, const Skew *);

}  // namespace fr3d
