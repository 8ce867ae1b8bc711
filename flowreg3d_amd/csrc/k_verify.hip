// k_verify.hip -- kernels of the VERIFICATION mode (fr3d_get_displacement_verify): the reference's arithmetic,
// operation by operation, on the engine's data path.
//
// What it is for: the shipped solver modes reformulate the update (square-root factors, frozen 3x3 system, fused
// multiply-adds, fp32 or packed storage) and agree with the CPU path to ~1e-5 voxels -- a tolerance, behind which a
// schedule or indexing slip that only shows at full size could hide.  This mode keeps the engine's STRUCTURE -- the
// compact skewed layout and its tables, the hyperplane launch schedule and tile decode of k_sor.hip, the resampler,
// prefilter, gather, tensor and median stages of the pyramid -- and evaluates the solver exactly as
// core/level_solver_3d.py:356-377,472-540 writes it: fp64 everywhere, the expanded quadratic form of psi_data, the
// per-channel accumulation order, true divisions, no FMA contraction, psi through the portable pow that the `ppow`
// build of the CPU restatement of the reference uses too (portable_pow.h).  GPU and CPU then agree BIT FOR BIT on the whole
// get_displacement (tests/test_gpu_verify_mode.py), at full size: every difference of the shipped modes against the
// CPU path is rounding, measured, not argued.  Speed is irrelevant here (about 5x the fp64-storage mode).
#include <algorithm>

#include "fr3d_internal.h"
#include "k_sor_core.h"
#include "portable_pow.h"

namespace fr3d {

#define VF_BX 64

// one voxel update in the reference's order; neighbours: planes s-1 hold the values of iteration t, planes s+1 those
// of iteration t-1 (hyperplane schedule = lexicographic Gauss-Seidel), a missing neighbour is the Neumann ghost
// (set_boundary_3d :246-259: the voxel's own old increment; add_boundary core/optical_flow_3d.py:88: the voxel's own u)
__global__ void __launch_bounds__(256)
k_sor_verify(const VerifyArgs a, int tau, int t_lo, int nent, const SorEntry *__restrict__ ent, const int *__restrict__ lut)
{
    const int Z = a.sk.Z, Y = a.sk.Y, X = a.sk.X;
    const int b = blockIdx.x;
    int lo = lut[b >> SOR_LUT_SHIFT];
    while (lo + 1 < nent && ent[lo + 1].pre <= b) lo++;
    const SorEntry en = ent[lo];
    const int local = b - en.pre;
    const int n = __builtin_amdgcn_readfirstlane((int)threadIdx.z);
    if (n >= sor_entry_nit(en)) return;
    const int t = t_lo + sor_entry_toff(en) + n;
    const int s = tau - 2 * t;
    const int k = (en.kb0 + local / en.njb) * (int)blockDim.y + __builtin_amdgcn_readfirstlane((int)threadIdx.y) - n;
    if (k < 0 || k >= Z) return;
    const int r = s - k;
    if (r < 0 || r > X + Y - 2) return;
    const int jm0 = sk_jm(X, r);
    const int jj = (local % en.njb) * VF_BX + threadIdx.x;
    const int j = jj + jm0;
    const int i = r - j;
    if (j >= Y || i < 0) return;

    const long long pm = a.sk.pb[s], p0 = a.sk.pb[s + 1], pp = a.sk.pb[s + 2];
    const long long cm = a.sk.cp[r], c_ = a.sk.cp[r + 1], cpn = a.sk.cp[r + 2];
    const long long b0 = p0 - c_, bm = pm - cm, bzm = pm - c_, bp = pp - cpn, bzp = pp - c_;
    const long long c0 = b0 + jj;
    const int d1 = jm0 - sk_jm(X, r - 1), d2 = jm0 - sk_jm(X, r + 1);
    const long long im = i > 0 ? bm + jj + d1 : c0;
    const long long ip = i < X - 1 ? bp + jj + d2 : c0;
    const long long jm = j > 0 ? bm + jj + d1 - 1 : c0;
    const long long jp = j < Y - 1 ? bp + jj + d2 + 1 : c0;
    const long long km = k > 0 ? bzm + jj : c0;
    const long long kp = k < Z - 1 ? bzp + jj : c0;

    const Rec<double, 3> U0 = ldrec<double, 3>(a.U, c0), D0 = ldrec<double, 3>(a.D, c0);
    Rec<double, 3> Uim = ldrec<double, 3>(a.U, im), Uip = ldrec<double, 3>(a.U, ip);
    Rec<double, 3> Ujm = ldrec<double, 3>(a.U, jm), Ujp = ldrec<double, 3>(a.U, jp);
    Rec<double, 3> Ukm = ldrec<double, 3>(a.U, km), Ukp = ldrec<double, 3>(a.U, kp);
    if (a.Ug) {
        // a ghost ring that is not the edge pad: u of a missing neighbour comes from the caller's padded arrays
        const long long gn = X + 2, gm = (long long)(Y + 2) * gn, gp = (long long)(Z + 2) * gm;
        auto ghost_u = [&](int kk, int jj, int ii) {  // padded coordinates
            Rec<double, 3> r;
            const long long o = (long long)kk * gm + (long long)jj * gn + ii;
            for (int c = 0; c < 3; c++) r.v[c] = a.Ug[(long long)c * gp + o];
            return r;
        };
        if (i == 0) Uim = ghost_u(k + 1, j + 1, 0);
        if (i == X - 1) Uip = ghost_u(k + 1, j + 1, X + 1);
        if (j == 0) Ujm = ghost_u(k + 1, 0, i + 1);
        if (j == Y - 1) Ujp = ghost_u(k + 1, Y + 1, i + 1);
        if (k == 0) Ukm = ghost_u(0, j + 1, i + 1);
        if (k == Z - 1) Ukp = ghost_u(Z + 1, j + 1, i + 1);
    }
    const Rec<double, 3> Dim = ldrec<double, 3>(a.D, im), Dip = ldrec<double, 3>(a.D, ip);
    const Rec<double, 3> Djm = ldrec<double, 3>(a.D, jm), Djp = ldrec<double, 3>(a.D, jp);
    const Rec<double, 3> Dkm = ldrec<double, 3>(a.D, km), Dkp = ldrec<double, 3>(a.D, kp);
    double du = D0.v[0], dv = D0.v[1], dw = D0.v[2];

    const double eps = 1e-6, OMEGA = 1.95;
    const double ax = a.ax, ay = a.ay, az = a.az;
    double denom_u = 0.0, denom_v = 0.0, denom_w = 0.0;
    double num_u = 0.0, num_v = 0.0, num_w = 0.0;
    if (a.Ps) {
        // level_solver_3d.py:400-471 (a_smooth != 1): z-, z+, y-, y+, x-, x+; the neighbour weight is the mean of psi_smooth
        // at the voxel and at the neighbour (ghost positions included: psi_smooth lives on the padded grid)
        const long long pn = X + 2, pm = (long long)(Y + 2) * pn;
        const long long pc = (long long)(k + 1) * pm + (long long)(j + 1) * pn + (i + 1);
        const double psc = a.Ps[pc];
        auto term = [&](const Rec<double, 3> &Un, const Rec<double, 3> &Dn, double psn, double sc) {
            const double tmp = 0.5 * (psc + psn) * sc;
            num_u += tmp * (Un.v[0] + Dn.v[0] - U0.v[0]);
            denom_u += tmp;
            num_v += tmp * (Un.v[1] + Dn.v[1] - U0.v[1]);
            denom_v += tmp;
            num_w += tmp * (Un.v[2] + Dn.v[2] - U0.v[2]);
            denom_w += tmp;
        };
        term(Ukm, Dkm, a.Ps[pc - pm], az);
        term(Ukp, Dkp, a.Ps[pc + pm], az);
        term(Ujm, Djm, a.Ps[pc - pn], ay);
        term(Ujp, Djp, a.Ps[pc + pn], ay);
        term(Uim, Dim, a.Ps[pc - 1], ax);
        term(Uip, Dip, a.Ps[pc + 1], ax);
    } else {
    // level_solver_3d.py:472-493 (a_smooth == 1), x then y then z, each sum left to right
    num_u += ax * (Uip.v[0] + Dip.v[0] + Uim.v[0] + Dim.v[0] - 2 * U0.v[0]);
    denom_u += 2 * ax;
    num_v += ax * (Uip.v[1] + Dip.v[1] + Uim.v[1] + Dim.v[1] - 2 * U0.v[1]);
    denom_v += 2 * ax;
    num_w += ax * (Uip.v[2] + Dip.v[2] + Uim.v[2] + Dim.v[2] - 2 * U0.v[2]);
    denom_w += 2 * ax;
    num_u += ay * (Ujp.v[0] + Djp.v[0] + Ujm.v[0] + Djm.v[0] - 2 * U0.v[0]);
    denom_u += 2 * ay;
    num_v += ay * (Ujp.v[1] + Djp.v[1] + Ujm.v[1] + Djm.v[1] - 2 * U0.v[1]);
    denom_v += 2 * ay;
    num_w += ay * (Ujp.v[2] + Djp.v[2] + Ujm.v[2] + Djm.v[2] - 2 * U0.v[2]);
    denom_w += 2 * ay;
    num_u += az * (Ukp.v[0] + Dkp.v[0] + Ukm.v[0] + Dkm.v[0] - 2 * U0.v[0]);
    denom_u += 2 * az;
    num_v += az * (Ukp.v[1] + Dkp.v[1] + Ukm.v[1] + Dkm.v[1] - 2 * U0.v[1]);
    denom_v += 2 * az;
    num_w += az * (Ukp.v[2] + Dkp.v[2] + Ukm.v[2] + Dkm.v[2] - 2 * U0.v[2]);
    denom_w += 2 * az;
    }

    // psi_data (:356-377) from the increments of iteration t-1 on update iterations, the stored value otherwise;
    // ww = weight [* psi]
    const bool upd = ((t + a.t_base) % a.update_lag) == 0;
    double ww[FR3D_MAX_CHANNELS];
    double J12[FR3D_MAX_CHANNELS], J13[FR3D_MAX_CHANNELS], J23[FR3D_MAX_CHANNELS], J14[FR3D_MAX_CHANNELS],
        J24[FR3D_MAX_CHANNELS], J34[FR3D_MAX_CHANNELS];
    for (int c = 0; c < a.C; c++) {
        const Rec<double, 10> J = ldrec<double, 10>(a.J[c], c0);  // J11,J22,J33,J44,J12,J13,J23,J14,J24,J34
        double w_ = (double)a.w[c][c0];
        const double adc = a.a_data[c];
        if (adc != 1.0) {
            double psi;
            if (upd) {
                double val = J.v[0] * du * du + J.v[1] * dv * dv + J.v[2] * dw * dw + 2.0 * J.v[4] * du * dv +
                             2.0 * J.v[5] * du * dw + 2.0 * J.v[6] * dv * dw + 2.0 * J.v[7] * du + 2.0 * J.v[8] * dv +
                             2.0 * J.v[9] * dw + J.v[3];
                if (val < 0.0) val = 0.0;
                psi = adc * fr3d_ppow(val + eps, adc - 1.0);
                a.psi[c][c0] = psi;
            } else {
                psi = a.psi[c][c0];
            }
            w_ *= psi;
        }
        ww[c] = w_;
        denom_u += w_ * J.v[0];
        denom_v += w_ * J.v[1];
        denom_w += w_ * J.v[2];
        J12[c] = J.v[4]; J13[c] = J.v[5]; J23[c] = J.v[6]; J14[c] = J.v[7]; J24[c] = J.v[8]; J34[c] = J.v[9];
    }
    double num_u2 = num_u;
    for (int c = 0; c < a.C; c++) num_u2 -= ww[c] * (J14[c] + J12[c] * dv + J13[c] * dw);
    const double du_kp1 = denom_u != 0.0 ? num_u2 / denom_u : 0.0;
    du = (1.0 - OMEGA) * du + OMEGA * du_kp1;
    double num_v2 = num_v;
    for (int c = 0; c < a.C; c++) num_v2 -= ww[c] * (J24[c] + J12[c] * du + J23[c] * dw);
    const double dv_kp1 = denom_v != 0.0 ? num_v2 / denom_v : 0.0;
    dv = (1.0 - OMEGA) * dv + OMEGA * dv_kp1;
    double num_w2 = num_w;
    for (int c = 0; c < a.C; c++) num_w2 -= ww[c] * (J34[c] + J13[c] * du + J23[c] * dv);
    const double dw_kp1 = denom_w != 0.0 ? num_w2 / denom_w : 0.0;
    dw = (1.0 - OMEGA) * dw + OMEGA * dw_kp1;
    Rec<double, 3> out;
    out.v[0] = du; out.v[1] = dv; out.v[2] = dw;
    strec<double, 3>(a.D, c0, out);
}

long long launch_sor_verify(hipStream_t st, const VerifyArgs &a, const SorChainSched &sc)
{
    FR3D_CHECK(a.sk.pb && a.sk.cp, "internal: the verification sweep runs on the compact skewed layout");
    FR3D_CHECK(a.C >= 1 && a.C <= FR3D_MAX_CHANNELS, "verification sweep: channel count out of range");
    FR3D_CHECK(VF_BX * sc.by * sc.nch <= 256, "internal: verification sweep workgroups hold at most 256 threads");
    long long launches = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        if (sc.ntiles[l] <= 0) continue;
        hipLaunchKernelGGL(k_sor_verify, dim3(sc.ntiles[l]), dim3(VF_BX, sc.by, sc.nch), 0, st, a, sc.tau[l], sc.t_lo[l],
                           sc.nent[l], sc.entries + sc.first[l], sc.lut + sc.lut_first[l]);
        FR3D_LAUNCH_CHECK();
        launches++;
    }
    return launches;
}

// nonlinearity_smoothness_3d (level_solver_3d.py:262-311) on the padded grid, one thread per padded voxel: uu = u + du with
// the padded arrays' contents as the reference has them when it evaluates psi_smooth of iteration t -- interior: the
// increments of iteration t-1; ghost ring: the edge pad of the increments of iteration t-2 (set_boundary_3d ran before the
// sweep of t-1) over the edge pad of u (add_boundary, core/optical_flow_3d.py:88) -- central differences with indices
// clamped to the padded array, always divided by 2h, the nine squares summed in the reference's order, portable pow.
__global__ void __launch_bounds__(256)
k_psi_smooth_verify(const Skew sk, const double *__restrict__ U, const double *__restrict__ D, const double *__restrict__ Dm2,
                    double a_smooth, double hx, double hy, double hz, double *__restrict__ Ps, const double *__restrict__ Ug)
{
    const int Z = sk.Z, Y = sk.Y, X = sk.X;
    const int P = Z + 2, M = Y + 2, N = X + 2;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)P * M * N) return;
    const int i = (int)(e % N), j = (int)((e / N) % M), k = (int)(e / ((long long)N * M));
    // uu at padded position (kk, jj, ii)
    auto uu = [&](int kk, int jj, int ii, double (&o)[3]) {
        const bool ghost = kk == 0 || kk == P - 1 || jj == 0 || jj == M - 1 || ii == 0 || ii == N - 1;
        const int kc = kk - 1 < 0 ? 0 : (kk - 1 > Z - 1 ? Z - 1 : kk - 1);
        const int jc = jj - 1 < 0 ? 0 : (jj - 1 > Y - 1 ? Y - 1 : jj - 1);
        const int ic = ii - 1 < 0 ? 0 : (ii - 1 > X - 1 ? X - 1 : ii - 1);
        const long long r = sk_index(sk, kc, jc, ic);
        Rec<double, 3> u = ldrec<double, 3>(U, r);
        const Rec<double, 3> d = ldrec<double, 3>(ghost ? Dm2 : D, r);
        if (ghost && Ug)  // a caller-chosen ghost ring of u,v,w (padded natural arrays, component-major)
            for (int c = 0; c < 3; c++) u.v[c] = Ug[((long long)c * P + kk) * M * N + (long long)jj * N + ii];
        for (int c = 0; c < 3; c++) o[c] = u.v[c] + d.v[c];
    };
    const int ixm = i > 0 ? i - 1 : 0, ixp = i < N - 1 ? i + 1 : N - 1;
    const int jym = j > 0 ? j - 1 : 0, jyp = j < M - 1 ? j + 1 : M - 1;
    const int kzm = k > 0 ? k - 1 : 0, kzp = k < P - 1 ? k + 1 : P - 1;
    double xp[3], xm[3], yp[3], ym[3], zp[3], zm[3];
    uu(k, j, ixp, xp); uu(k, j, ixm, xm);
    uu(k, jyp, i, yp); uu(k, jym, i, ym);
    uu(kzp, j, i, zp); uu(kzm, j, i, zm);
    double g = 0.0;
    bool first = true;
    for (int c = 0; c < 3; c++) {
        const double dx = (xp[c] - xm[c]) / (2 * hx);
        const double dy = (yp[c] - ym[c]) / (2 * hy);
        const double dz = (zp[c] - zm[c]) / (2 * hz);
        if (first) { g = dx * dx; first = false; }
        else g = g + dx * dx;
        g = g + dy * dy;
        g = g + dz * dz;
    }
    if (g < 0.0) g = 0.0;
    Ps[e] = a_smooth * fr3d_ppow(g + 1e-5, a_smooth - 1.0);
}

void launch_psi_smooth_verify(hipStream_t st, const Skew &sk, const double *U, const double *D, const double *Dm2, double a_smooth,
                              double hx, double hy, double hz, double *Ps, const double *Ug)
{
    const long long n = (long long)(sk.Z + 2) * (sk.Y + 2) * (sk.X + 2);
    hipLaunchKernelGGL(k_psi_smooth_verify, dim3(cdiv(n, 256)), dim3(256), 0, st, sk, U, D, Dm2, a_smooth, hx, hy, hz, Ps, Ug);
    FR3D_LAUNCH_CHECK();
}

// ---- fp64 tail of a level ---------------------------------------------------------------------------------------

// scipy.ndimage.median_filter(size=5^3, mode="mirror") on fp64 (core/optical_flow_3d.py:517-526): the 63rd smallest of
// the 125 window values (any correct selection returns the same value)
__device__ __forceinline__ int mirror_idx(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}
__global__ void __launch_bounds__(64)
k_median5_f64(const double *__restrict__ in, int Z, int Y, int X, double *__restrict__ out)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)Z * Y * X) return;
    const int x = (int)(tid % X), y = (int)((tid / X) % Y), z = (int)(tid / ((long long)X * Y));
    double buf[125];
    int zi[5], yi[5], xi[5];
    for (int q = 0; q < 5; q++) {
        zi[q] = mirror_idx(z + q - 2, Z);
        yi[q] = mirror_idx(y + q - 2, Y);
        xi[q] = mirror_idx(x + q - 2, X);
    }
    int cnt = 0;
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            for (int c = 0; c < 5; c++) buf[cnt++] = in[((size_t)zi[a] * Y + yi[b]) * X + xi[c]];
    // quickselect of rank 62
    int lo = 0, hi = 124;
    const int kth = 62;
    while (lo < hi) {
        const double piv = buf[(lo + hi) >> 1];
        int p = lo, q = hi;
        while (p <= q) {
            while (buf[p] < piv) p++;
            while (buf[q] > piv) q--;
            if (p <= q) {
                const double tmp = buf[p]; buf[p] = buf[q]; buf[q] = tmp;
                p++; q--;
            }
        }
        if (kth <= q) hi = q;
        else if (kth >= p) lo = p;
        else break;
    }
    out[tid] = buf[kth];
}
void launch_median5_f64(hipStream_t st, const double *in, int Z, int Y, int X, double *out)
{
    const long long n = (long long)Z * Y * X;
    hipLaunchKernelGGL(k_median5_f64, dim3(cdiv(n, 64)), dim3(64), 0, st, in, Z, Y, X, out);
    FR3D_LAUNCH_CHECK();
}

template <typename TS, typename TD>
__global__ void k_cast(const TS *__restrict__ src, long long n, TD *__restrict__ dst)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = (TD)src[t];
}
template <typename TS, typename TD>
void launch_cast(hipStream_t st, const TS *src, long long n, TD *dst)
{
    hipLaunchKernelGGL((k_cast<TS, TD>), dim3(cdiv(n, 256)), dim3(256), 0, st, src, n, dst);
    FR3D_LAUNCH_CHECK();
}
template void launch_cast<float, double>(hipStream_t, const float *, long long, double *);
template void launch_cast<double, float>(hipStream_t, const double *, long long, float *);

__global__ void k_axpy_f64(double *__restrict__ y, const double *__restrict__ x, long long n)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = y[t] + x[t];
}
void launch_axpy_f64(hipStream_t st, double *y, const double *x, long long n)
{
    hipLaunchKernelGGL(k_axpy_f64, dim3(cdiv(n, 256)), dim3(256), 0, st, y, x, n);
    FR3D_LAUNCH_CHECK();
}

// test hook: the portable pow on the device (tests compare it bit for bit with the same source compiled by gcc)
__global__ void k_ppow(const double *__restrict__ x, const double *__restrict__ y, long long n, double *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = fr3d_ppow(x[t], y[t]);
}
void launch_ppow(hipStream_t st, const double *x, const double *y, long long n, double *out)
{
    hipLaunchKernelGGL(k_ppow, dim3(cdiv(n, 256)), dim3(256), 0, st, x, y, n, out);
    FR3D_LAUNCH_CHECK();
}

// (Z,Y,X,3) interleaved fp64 flow from three planar arrays
__global__ void k_pack3_f64(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, long long n,
                            double *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        out[3 * t + 0] = a[t];
        out[3 * t + 1] = b[t];
        out[3 * t + 2] = c[t];
    }
}
void launch_pack3_f64(hipStream_t st, const double *a, const double *b, const double *c, long long n, double *out)
{
    hipLaunchKernelGGL(k_pack3_f64, dim3(cdiv(n, 256)), dim3(256), 0, st, a, b, c, n, out);
    FR3D_LAUNCH_CHECK();
}

}  // namespace fr3d
