// k_sor_smooth.hip -- the a_smooth != 1 branch of the inner solver:
// K5 psi_smooth every iteration (core/level_solver_3d.py:262-311, 352-355) and the psi-weighted
// diffusion stencil of the sweep (:400-471), lexicographic-exact like k_sor.hip.
//
// What changes against the a_smooth == 1 kernel:
//  * every neighbour weight is tmp = 0.5*(psi_s[c] + psi_s[nb]) * alpha/h^2, so the u,v,w part of the
//    stencil is no longer iteration-invariant: the kernel reads u,v,w (skewed) next to du,dv,dw.
//  * psi_s of iteration t is a radius-1 function of uu = u + du as it stands BEFORE the sweep of
//    iteration t: interior = increments of iteration t-1, ghost ring = edge pad of the increments
//    of iteration t-2 (set_boundary_3d ran before sweep t-1, :379-381).  Both states must survive
//    while iteration t runs, so the increments are triple-buffered (D[t%3]) and the in-flight
//    iterations are 4 hyperplanes apart instead of 2:
//        step tau:  P-stage  psi_s^t(plane q)   for q = tau + 2 - 4t   (needs D[(t-1)%3] on q-1..q+1)
//                   sweep    iteration t, plane s = tau - 4t           (needs psi_s^t on s-1..s+1)
//    psi_s at ghost positions (needed by surface voxels) is evaluated on the fly from the same
//    two buffers.  S + 4(T-1) + 2 steps per level, two launches per step; this path favours
//    fidelity over speed (it is not on any BASELINE configuration: OFOptions.a_smooth = 1.0).
#include <cstdlib>

#include "fr3d_internal.h"

namespace fr3d {

#define SM_OMEGA 1.95

// uu component `c` at padded-grid position (k,j,i) in [-1,Z] x [-1,Y] x [-1,X] (interior coordinates)
template <typename S>
__device__ __forceinline__ double uu_at(const SmoothView<S> &v, int c, int k, int j, int i)
{
    const bool ghost = k < 0 || k >= v.Z || j < 0 || j >= v.Y || i < 0 || i >= v.X;
    const int kc = k < 0 ? 0 : (k >= v.Z ? v.Z - 1 : k);
    const int jc = j < 0 ? 0 : (j >= v.Y ? v.Y - 1 : j);
    const int ic = i < 0 ? 0 : (i >= v.X ? v.X - 1 : i);
    const size_t o = (size_t)sk_index(v.X, v.Yp, v.plane, kc, jc, ic);
    return (double)v.U[c][o] + (double)(ghost ? v.Dm2[c][o] : v.Dm1[c][o]);
}

// nonlinearity_smoothness_3d at one padded position (indices clamped to the padded array,
// differences always divided by 2h, :280-311)
template <typename S>
__device__ double psi_smooth_at(const SmoothView<S> &v, int k, int j, int i)
{
    auto cl = [](int q, int n) { return q < -1 ? -1 : (q > n ? n : q); };
    const int km = cl(k - 1, v.Z), kp = cl(k + 1, v.Z);
    const int jm = cl(j - 1, v.Y), jp = cl(j + 1, v.Y);
    const int im = cl(i - 1, v.X), ip = cl(i + 1, v.X);
    double g = 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double dx = (uu_at(v, c, k, j, ip) - uu_at(v, c, k, j, im)) / (2.0 * v.hx);
        const double dy = (uu_at(v, c, k, jp, i) - uu_at(v, c, k, jm, i)) / (2.0 * v.hy);
        const double dz = (uu_at(v, c, kp, j, i) - uu_at(v, c, km, j, i)) / (2.0 * v.hz);
        g += dx * dx;
        g += dy * dy;
        g += dz * dz;
    }
    if (g < 0.0) g = 0.0;
    // fp32 storage: psi_s is rounded to fp32 when stored or used, so a 1-ulp powf is enough (the fp64
    // pow is a ~600-instruction dependent chain per voxel and iteration); fp64 storage keeps pow
    if (sizeof(S) == 4) return v.a_smooth * (double)powf((float)(g + 1e-5), (float)(v.a_smooth - 1.0));
    return v.a_smooth * pow(g + 1e-5, v.a_smooth - 1.0);
}

// P-stage: psi_s^t on hyperplane q for every in-flight t (grid.z)
template <typename S>
__global__ void __launch_bounds__(256)
k_psi_smooth(const SmoothArgs<S> a, int tau, int t_lo)
{
    const int t = t_lo + blockIdx.z;
    const int q = tau + 2 - SM_LAG * t;
    if (t >= a.iterations || q < 0 || q >= a.S_planes) return;
    const int k = blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.view.Z || j >= a.view.Y) return;
    const int i = q - k - j;
    if (i < 0 || i >= a.view.X) return;
    SmoothView<S> v = a.view;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        v.Dm1[c] = a.D[(t + 2) % 3][c];  // (t-1) mod 3
        v.Dm2[c] = a.D[(t + 1) % 3][c];  // (t-2) mod 3
    }
    a.Ps[(size_t)sk_index(v.X, v.Yp, v.plane, k, j, i)] = (S)psi_smooth_at(v, k, j, i);
}

// sweep: iteration t on hyperplane s = tau - 4t
template <typename S, int C>
__global__ void __launch_bounds__(256)
k_sor_smooth(const SmoothArgs<S> a, int tau, int t_lo)
{
    const int t = t_lo + blockIdx.z;
    const int s = tau - SM_LAG * t;
    if (t >= a.iterations || s < 0 || s >= a.S_planes) return;
    const int Z = a.view.Z, Y = a.view.Y, X = a.view.X;
    const int k = blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Z || j >= Y) return;
    const int i = s - k - j;
    if (i < 0 || i >= X) return;
    SmoothView<S> v = a.view;
    S *const *Dn = a.D[t % 3];                 // new values (this iteration)
    const S *const *Do = a.D[(t + 2) % 3];     // old values (iteration t-1)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        v.Dm1[c] = a.D[(t + 2) % 3][c];
        v.Dm2[c] = a.D[(t + 1) % 3][c];
    }
    const size_t c0 = (size_t)sk_index(X, v.Yp, v.plane, k, j, i);
    const double d0[3] = {(double)Do[0][c0], (double)Do[1][c0], (double)Do[2][c0]};
    const double u0[3] = {(double)v.U[0][c0], (double)v.U[1][c0], (double)v.U[2][c0]};
    const double ps_c = (double)a.Ps[c0];

    // neighbours in the reference's order: k-1, k+1, j-1, j+1, i-1, i+1 (:401-471)
    const int nk[6] = {k - 1, k + 1, k, k, k, k};
    const int nj[6] = {j, j, j - 1, j + 1, j, j};
    const int ni[6] = {i, i, i, i, i - 1, i + 1};
    const double sc[6] = {a.az, a.az, a.ay, a.ay, a.ax, a.ax};
    const bool newer[6] = {true, false, true, false, true, false};  // minus side already swept
    // psi_s and the neighbour terms of the six neighbours.  Inside the volume: loads at the (clamped)
    // neighbour position, all unconditional.  Ghost neighbours need psi_s evaluated on the fly, which
    // is expensive and diverges; per axis at most one of the two neighbours is a ghost (both only
    // when that axis has length 1), so it is evaluated once per axis instead of once per neighbour.
    double psn[6], term[6][3];
    bool inside[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        inside[q] = nk[q] >= 0 && nk[q] < Z && nj[q] >= 0 && nj[q] < Y && ni[q] >= 0 && ni[q] < X;
        const int kc = min(max(nk[q], 0), Z - 1), jc = min(max(nj[q], 0), Y - 1), ic = min(max(ni[q], 0), X - 1);
        const size_t o = (size_t)sk_index(X, v.Yp, v.plane, kc, jc, ic);
        psn[q] = (double)a.Ps[o];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double nb = (double)v.U[c][o] + (double)(newer[q] ? Dn[c][o] : Do[c][o]) - u0[c];
            // ghost: u is edge-padded (u_nb = u_c) and du holds the Neumann copy of the voxel's own
            // previous increment (set_boundary_3d ran right before this sweep)
            term[q][c] = inside[q] ? nb : (u0[c] + d0[c]) - u0[c];
        }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        const int qm = 2 * ax, qp = 2 * ax + 1;
        if (!inside[qm] || !inside[qp]) {
            const int g = !inside[qm] ? qm : qp;
            const double val = psi_smooth_at(v, nk[g], nj[g], ni[g]);
            if (!inside[qm]) psn[qm] = val;
            else psn[qp] = val;
            if (!inside[qm] && !inside[qp]) psn[qp] = psi_smooth_at(v, nk[qp], nj[qp], ni[qp]);  // axis of length 1
        }
    }
    double num[3] = {0.0, 0.0, 0.0}, den = 0.0;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const double tmp = 0.5 * (ps_c + psn[q]) * sc[q];
#pragma unroll
        for (int c = 0; c < 3; c++) num[c] += tmp * term[q][c];
        den += tmp;
    }

    // data term: frozen system between psi_data updates (same construction as k_sor.hip, without L)
    double M11, M22, M33, M12, M13, M23, bu, bv, bw;
    if ((t % a.update_lag) == 0) {
        M11 = M22 = M33 = M12 = M13 = M23 = bu = bv = bw = 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            double f[12];
#pragma unroll
            for (int q = 0; q < 12; q++) f[q] = (double)a.A[q * FR3D_MAX_CHANNELS + c][c0];
            double wt = (double)a.weight[c][c0];
            const double adc = a.a_data[c];
            if (adc != 1.0) {
                double val = 0.0;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const double r = f[4 * e] * d0[0] + f[4 * e + 1] * d0[1] + f[4 * e + 2] * d0[2] + f[4 * e + 3];
                    val += r * r;
                }
                wt *= adc * pow(val + 1e-6, adc - 1.0);
            }
            wt = (double)(S)wt;
            M11 += wt * (f[0] * f[0] + f[4] * f[4] + f[8] * f[8]);
            M22 += wt * (f[1] * f[1] + f[5] * f[5] + f[9] * f[9]);
            M33 += wt * (f[2] * f[2] + f[6] * f[6] + f[10] * f[10]);
            M12 += wt * (f[0] * f[1] + f[4] * f[5] + f[8] * f[9]);
            M13 += wt * (f[0] * f[2] + f[4] * f[6] + f[8] * f[10]);
            M23 += wt * (f[1] * f[2] + f[5] * f[6] + f[9] * f[10]);
            bu += wt * (f[0] * f[3] + f[4] * f[7] + f[8] * f[11]);
            bv += wt * (f[1] * f[3] + f[5] * f[7] + f[9] * f[11]);
            bw += wt * (f[2] * f[3] + f[6] * f[7] + f[10] * f[11]);
        }
        const double vals[9] = {M11, M22, M33, M12, M13, M23, bu, bv, bw};
#pragma unroll
        for (int q = 0; q < 9; q++) a.M[q][c0] = (S)vals[q];
        M11 = (double)(S)M11; M22 = (double)(S)M22; M33 = (double)(S)M33;
        M12 = (double)(S)M12; M13 = (double)(S)M13; M23 = (double)(S)M23;
        bu = (double)(S)bu; bv = (double)(S)bv; bw = (double)(S)bw;
    } else {
        M11 = (double)a.M[0][c0]; M22 = (double)a.M[1][c0]; M33 = (double)a.M[2][c0];
        M12 = (double)a.M[3][c0]; M13 = (double)a.M[4][c0]; M23 = (double)a.M[5][c0];
        bu = (double)a.M[6][c0]; bv = (double)a.M[7][c0]; bw = (double)a.M[8][c0];
    }
    const double den_u = den + M11, den_v = den + M22, den_w = den + M33;
    double n2 = num[0] - (bu + M12 * d0[1] + M13 * d0[2]);
    const double du1 = (1.0 - SM_OMEGA) * d0[0] + SM_OMEGA * (den_u != 0.0 ? n2 / den_u : 0.0);
    n2 = num[1] - (bv + M12 * du1 + M23 * d0[2]);
    const double dv1 = (1.0 - SM_OMEGA) * d0[1] + SM_OMEGA * (den_v != 0.0 ? n2 / den_v : 0.0);
    n2 = num[2] - (bw + M13 * du1 + M23 * dv1);
    const double dw1 = (1.0 - SM_OMEGA) * d0[2] + SM_OMEGA * (den_w != 0.0 ? n2 / den_w : 0.0);
    Dn[0][c0] = (S)du1;
    Dn[1][c0] = (S)dv1;
    Dn[2][c0] = (S)dw1;
}

template <typename S>
long long launch_sor_smooth(hipStream_t st, const SmoothArgs<S> &a)
{
    const int T = a.iterations, Sp = a.S_planes;
    if (T <= 0) return 0;
    const int Z = a.view.Z, Y = a.view.Y;
    const dim3 block(64, 4);
    const int gx = cdiv(Y, 64), gy = cdiv(Z, 4);
    long long launches = 0;
    const int last = (Sp - 1) + SM_LAG * (T - 1);
    for (int tau = -2; tau <= last; tau++) {
        // P-stage: t with 0 <= tau + 2 - 4t < Sp
        {
            int hi = (tau + 2) / SM_LAG;
            if (tau + 2 < 0) hi = -1;
            int lo_num = tau + 2 - (Sp - 1);
            int lo = lo_num <= 0 ? 0 : (lo_num + SM_LAG - 1) / SM_LAG;
            if (hi > T - 1) hi = T - 1;
            if (lo <= hi) {
                hipLaunchKernelGGL(k_psi_smooth<S>, dim3(gx, gy, hi - lo + 1), block, 0, st, a, tau, lo);
                launches++;
            }
        }
        if (tau >= 0) {
            int hi = tau / SM_LAG;
            int lo_num = tau - (Sp - 1);
            int lo = lo_num <= 0 ? 0 : (lo_num + SM_LAG - 1) / SM_LAG;
            if (hi > T - 1) hi = T - 1;
            if (lo <= hi) {
                dim3 grid(gx, gy, hi - lo + 1);
                switch (a.C) {
                    case 1: hipLaunchKernelGGL((k_sor_smooth<S, 1>), grid, block, 0, st, a, tau, lo); break;
                    case 2: hipLaunchKernelGGL((k_sor_smooth<S, 2>), grid, block, 0, st, a, tau, lo); break;
                    case 3: hipLaunchKernelGGL((k_sor_smooth<S, 3>), grid, block, 0, st, a, tau, lo); break;
                    case 4: hipLaunchKernelGGL((k_sor_smooth<S, 4>), grid, block, 0, st, a, tau, lo); break;
                    default: throw Error("SOR kernel is instantiated for 1..4 channels");
                }
                launches++;
            }
        }
    }
    return launches;
}
template long long launch_sor_smooth<float>(hipStream_t, const SmoothArgs<float> &);
template long long launch_sor_smooth<double>(hipStream_t, const SmoothArgs<double> &);

}  // namespace fr3d
