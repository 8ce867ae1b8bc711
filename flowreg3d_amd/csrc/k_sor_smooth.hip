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
//        step n:  P-stage  psi_s^t(plane n - 4t)      (needs D[(t-1)%3] on the planes next to it)
//                 sweep    iteration t, plane n - 2 - 4t  (needs psi_s^t on the planes next to it)
//    psi_s at ghost positions (needed by surface voxels) is evaluated on the fly from the same two
//    buffers.  One launch per step on the lag-4 tile schedule of k_sor.hip: P-stage tiles, sweep tiles
//    and -- packed into dense workgroups of their own, because their ghost handling is expensive and
//    would otherwise diverge in the first and last wave of every row -- the surface voxels of both.
//    fp64 arithmetic (the (u + du) - u differences cancel in fp32); not on any BASELINE configuration
//    (OFOptions.a_smooth = 1.0), but get_displacement's own default.
//  * operands are records like in k_sor.hip (u,v,w / du,dv,dw: 3 values, system: 9, factors: 12 per voxel
//    contiguous, compact skewed rows), so a neighbour costs three wide loads instead of seven dword loads, and
//    the volumes of a lock-step batch share the launches (blockIdx.y = volume of the batch).
#include <cstdlib>
#include <cstring>

#include "fr3d_internal.h"
#include "k_sor_core.h"

namespace fr3d {

#define SM_OMEGA 1.95

// uu = u + du (three components) at padded-grid position (k,j,i) in [-1,Z] x [-1,Y] x [-1,X] (interior coordinates)
template <typename S>
__device__ __forceinline__ void uu3_at(const SmoothView<S> &v, int k, int j, int i, double (&uu)[3])
{
    const bool ghost = k < 0 || k >= v.Z || j < 0 || j >= v.Y || i < 0 || i >= v.X;
    const int kc = k < 0 ? 0 : (k >= v.Z ? v.Z - 1 : k);
    const int jc = j < 0 ? 0 : (j >= v.Y ? v.Y - 1 : j);
    const int ic = i < 0 ? 0 : (i >= v.X ? v.X - 1 : i);
    const long long o = sk_index(v.sk, kc, jc, ic);
    // both buffers are read and the value is selected: selecting the POINTER (v.Dm2 vs v.Dm1) would cost a
    // 64-bit select per access and no load less
    const Rec<S, 3> u = ldrec<S, 3>(v.U, o), d1 = ldrec<S, 3>(v.Dm1, o), d2 = ldrec<S, 3>(v.Dm2, o);
#pragma unroll
    for (int c = 0; c < 3; c++) uu[c] = (double)u.v[c] + (ghost ? (double)d2.v[c] : (double)d1.v[c]);
}

// nonlinearity_smoothness_3d at one padded position (indices clamped to the padded array,
// differences always divided by 2h, :280-311)
template <typename S>
__device__ __forceinline__ double psi_smooth_at(const SmoothView<S> &v, int k, int j, int i)
{
    auto cl = [](int q, int n) { return q < -1 ? -1 : (q > n ? n : q); };
    const int km = cl(k - 1, v.Z), kp = cl(k + 1, v.Z);
    const int jm = cl(j - 1, v.Y), jp = cl(j + 1, v.Y);
    const int im = cl(i - 1, v.X), ip = cl(i + 1, v.X);
    double xp[3], xm[3], yp[3], ym[3], zp[3], zm[3];
    uu3_at(v, k, j, ip, xp);
    uu3_at(v, k, j, im, xm);
    uu3_at(v, k, jp, i, yp);
    uu3_at(v, k, jm, i, ym);
    uu3_at(v, kp, j, i, zp);
    uu3_at(v, km, j, i, zm);
    double g = 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double dx = div_by_const(xp[c] - xm[c], v.tx, v.rtx);
        const double dy = div_by_const(yp[c] - ym[c], v.ty, v.rty);
        const double dz = div_by_const(zp[c] - zm[c], v.tz, v.rtz);
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

// a.D[m][c] for a wave-uniform m in 0..2, chosen with selects: indexing the kernel argument with a
// run-time value would move the pointer table to scratch memory and turn every access through it
// into a flat load
template <typename S>
__device__ __forceinline__ S *pick_buffer(const SmoothArgs<S> &a, int m)
{
    return (m == 0 ? a.D[0] : (m == 1 ? a.D[1] : a.D[2])) + (size_t)blockIdx.y * a.vsD;
}
// the geometry + this volume's u,v,w; Dm1/Dm2 are set by the callers
template <typename S>
__device__ __forceinline__ SmoothView<S> volume_view(const SmoothArgs<S> &a)
{
    SmoothView<S> v = a.view;
    v.U = a.view.U + (size_t)blockIdx.y * a.vsU;
    return v;
}

// Thread -> voxel mapping shared by both kernels: the tile schedule of k_sor.hip with SM_LAG planes
// between iterations (blockIdx.x enumerates the 64 x blockDim.y tiles of all in-flight iterations,
// rows are the left-aligned skewed rows, lanes run along j).  Returns false for lanes without a voxel.
struct SmoothPos {
    int t, s, k, j, i;
    size_t c0, xm, xp, ym, yp, zm, zp;  // element offsets of the voxel and its six neighbours
};
// offsets of voxel (k,j) of hyperplane s; false if (k,j) is not a voxel of that plane
template <typename S>
__device__ __forceinline__ bool smooth_offsets(const SmoothArgs<S> &a, int s, int k, int jj_or_j, bool is_jj, SmoothPos &p)
{
    const int Z = a.view.Z, Y = a.view.Y, X = a.view.X;
    const Skew &sk = a.view.sk;
    if (k >= Z) return false;
    const int r = s - k;
    const int jm0 = sk_jm(X, r);
    const int jj = is_jj ? jj_or_j : jj_or_j - jm0;
    p.s = s;
    p.k = k;
    p.j = jj + jm0;
    p.i = r - p.j;
    if (r < 0 || p.j >= Y || p.i < 0) return false;
    // row starts: own, (s-1,k), (s-1,k-1), (s+1,k), (s+1,k+1) -- compact rows (pb/cp tables, see Skew and
    // k_sor_step) or pitched rows; the rows k-1 / k+1 of the neighbouring planes have the same i+j as this row
    long long b0, bm, bzm, bp, bzp;
    if (sk.pb) {
        const long long pm = sk.pb[s], p0 = sk.pb[s + 1], pp = sk.pb[s + 2];
        const long long cm = sk.cp[r], c_ = sk.cp[r + 1], cpn = sk.cp[r + 2];
        b0 = p0 - c_; bm = pm - cm; bzm = pm - c_; bp = pp - cpn; bzp = pp - c_;
    } else {
        b0 = (long long)s * sk.plane + (long long)k * sk.Yp;
        bm = b0 - sk.plane; bzm = bm - sk.Yp; bp = b0 + sk.plane; bzp = bp + sk.Yp;
    }
    const long long d1 = jm0 - sk_jm(X, r - 1), d2 = jm0 - sk_jm(X, r + 1);
    p.c0 = (size_t)(b0 + jj);
    p.xm = (size_t)(bm + jj + d1);
    p.xp = (size_t)(bp + jj + d2);
    p.ym = (size_t)(bm + jj + d1 - 1);
    p.yp = (size_t)(bp + jj + d2 + 1);
    p.zm = (size_t)(bzm + jj);
    p.zp = (size_t)(bzp + jj);
    return true;
}
template <typename S>
__device__ __forceinline__ bool on_surface(const SmoothArgs<S> &a, const SmoothPos &p)
{
    return p.k == 0 || p.k == a.view.Z - 1 || p.j == 0 || p.j == a.view.Y - 1 || p.i == 0 || p.i == a.view.X - 1;
}

// One part of a step: the tiles (interior voxels) or the surface-voxel workgroups of the iterations
// t_lo .. t_lo+nt-1, iteration t working on hyperplane tau - SM_LAG*t.
struct StepPart {
    int tau, t_lo, nt, ntiles;
    const SorEntry *ent;
    const int *lut;
};
// interior tile `b` of a part -> voxel of this lane (false: no voxel here, or a surface voxel, which
// belongs to the surface workgroups)
template <typename S>
__device__ __forceinline__ bool locate_tile(const SmoothArgs<S> &a, int b, const StepPart &P, SmoothPos &p)
{
    int lo = P.lut[b >> SOR_LUT_SHIFT];
    while (lo + 1 < P.nt && P.ent[lo + 1].pre <= b) lo++;
    const SorEntry en = P.ent[lo];
    const int local = b - en.pre;
    p.t = P.t_lo + lo;
    // a wave is one row (blockDim.x == 64): its row number is wave-uniform, which keeps the row starts in SGPRs
    const int k = (en.kb0 + local / en.njb) * (int)blockDim.y + __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int jj = (local % en.njb) * 64 + threadIdx.x;
    if (!smooth_offsets(a, P.tau - SM_LAG * p.t, k, jj, true, p)) return false;
    return !on_surface(a, p);
}
// surface workgroup `b` of a part (cb workgroups of 256 lanes per iteration) -> surface voxel of this lane
template <typename S>
__device__ __forceinline__ bool locate_surface(const SmoothArgs<S> &a, int b, int cb, const StepPart &P,
                                               const int *__restrict__ meta, const int *__restrict__ kj, SmoothPos &p)
{
    p.t = P.t_lo + b / cb;
    const int s = P.tau - SM_LAG * p.t;
    if (s < 0 || s >= a.S_planes) return false;
    const int n = (b % cb) * 256 + threadIdx.y * 64 + threadIdx.x;
    if (n >= meta[2 * s + 1]) return false;
    const int v = kj[meta[2 * s] + n];
    return smooth_offsets(a, s, v >> 16, v & 0xffff, false, p);
}

// P-stage: psi_s^t of one voxel.  INTERIOR: no clamping, no ghosts -- psi_smooth_at with the six
// neighbours at row-uniform offsets; otherwise the generic evaluation with clamped indices.
template <typename S, bool INTERIOR>
__device__ __forceinline__ void psi_voxel(const SmoothArgs<S> &a, const SmoothPos &p)
{
    SmoothView<S> v = volume_view(a);
    v.Dm1 = pick_buffer(a, (p.t + 2) % 3);  // (t-1) mod 3
    v.Dm2 = pick_buffer(a, (p.t + 1) % 3);  // (t-2) mod 3
    double ps;
    if constexpr (INTERIOR) {
        const S *U = v.U, *D = v.Dm1;
        const Rec<S, 3> uxp = ldrec<S, 3>(U, p.xp), dxp = ldrec<S, 3>(D, p.xp), uxm = ldrec<S, 3>(U, p.xm), dxm = ldrec<S, 3>(D, p.xm);
        const Rec<S, 3> uyp = ldrec<S, 3>(U, p.yp), dyp = ldrec<S, 3>(D, p.yp), uym = ldrec<S, 3>(U, p.ym), dym = ldrec<S, 3>(D, p.ym);
        const Rec<S, 3> uzp = ldrec<S, 3>(U, p.zp), dzp = ldrec<S, 3>(D, p.zp), uzm = ldrec<S, 3>(U, p.zm), dzm = ldrec<S, 3>(D, p.zm);
        double g = 0.0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double xp = (double)uxp.v[c] + (double)dxp.v[c], xm = (double)uxm.v[c] + (double)dxm.v[c];
            const double yp = (double)uyp.v[c] + (double)dyp.v[c], ym = (double)uym.v[c] + (double)dym.v[c];
            const double zp = (double)uzp.v[c] + (double)dzp.v[c], zm = (double)uzm.v[c] + (double)dzm.v[c];
            const double dx = div_by_const(xp - xm, v.tx, v.rtx);
            const double dy = div_by_const(yp - ym, v.ty, v.rty);
            const double dz = div_by_const(zp - zm, v.tz, v.rtz);
            g += dx * dx;
            g += dy * dy;
            g += dz * dz;
        }
        if (g < 0.0) g = 0.0;
#ifdef FR3D_EXPERIMENTS
        if (a.dbg & 16) ps = g; else  // no pow
#endif
        if (sizeof(S) == 4) ps = v.a_smooth * (double)powf((float)(g + 1e-5), (float)(v.a_smooth - 1.0));
        else ps = v.a_smooth * pow(g + 1e-5, v.a_smooth - 1.0);
    } else {
        ps = psi_smooth_at(v, p.k, p.j, p.i);  // clamped indices and ghost values
    }
    a.Ps[(size_t)blockIdx.y * a.vsP + p.c0] = (S)ps;
}

// sweep: iteration t on hyperplane s = tau - 4t
template <typename S, int C, bool INTERIOR>
__device__ __forceinline__ void sweep_voxel(const SmoothArgs<S> &a, const SmoothPos &p)
{
    const int Z = a.view.Z, Y = a.view.Y, X = a.view.X;
    const int t = p.t, k = p.k, j = p.j, i = p.i;
    SmoothView<S> v = volume_view(a);
    S *Dn = pick_buffer(a, t % 3);              // new values (this iteration)
    const S *Do = pick_buffer(a, (t + 2) % 3);  // old values (iteration t-1)
    v.Dm1 = Do;
    v.Dm2 = pick_buffer(a, (t + 1) % 3);
    const S *Ps = a.Ps + (size_t)blockIdx.y * a.vsP;
    const size_t c0 = p.c0;
    const Rec<S, 3> d0r = ldrec<S, 3>(Do, c0), u0r = ldrec<S, 3>(v.U, c0);
    const double d0[3] = {(double)d0r.v[0], (double)d0r.v[1], (double)d0r.v[2]};
    const double u0[3] = {(double)u0r.v[0], (double)u0r.v[1], (double)u0r.v[2]};
    const double ps_c = (double)Ps[c0];

    // neighbours in the reference's order: k-1, k+1, j-1, j+1, i-1, i+1 (:401-471)
    const int nk[6] = {k - 1, k + 1, k, k, k, k};
    const int nj[6] = {j, j, j - 1, j + 1, j, j};
    const int ni[6] = {i, i, i, i, i - 1, i + 1};
    const size_t off[6] = {p.zm, p.zp, p.ym, p.yp, p.xm, p.xp};
    const bool inside[6] = {INTERIOR || k > 0, INTERIOR || k < Z - 1, INTERIOR || j > 0,
                            INTERIOR || j < Y - 1, INTERIOR || i > 0, INTERIOR || i < X - 1};
    const double sc[6] = {a.az, a.az, a.ay, a.ay, a.ax, a.ax};
    const bool newer[6] = {true, false, true, false, true, false};  // minus side already swept
    // psi_s and the neighbour terms.  Inside the volume: loads at the neighbour's offset (a ghost
    // reads c0 instead), all unconditional.  Ghost neighbours need psi_s evaluated on the fly, which is
    // expensive and diverges; per axis at most one of the two neighbours is a ghost (both only when
    // that axis has length 1), so it is evaluated once per axis instead of once per neighbour.
    double psn[6], term[6][3];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const size_t o = inside[q] ? off[q] : c0;
        psn[q] = (double)Ps[o];
        const Rec<S, 3> un = ldrec<S, 3>(v.U, o), dn = ldrec<S, 3>(newer[q] ? (const S *)Dn : Do, o);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double nb = (double)un.v[c] + (double)dn.v[c] - u0[c];
            // ghost: u is edge-padded (u_nb = u_c) and du holds the Neumann copy of the voxel's own
            // previous increment (set_boundary_3d ran right before this sweep)
            term[q][c] = inside[q] ? nb : (u0[c] + d0[c]) - u0[c];
        }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        const int qm = 2 * ax, qp = 2 * ax + 1;
        if (!inside[qm] || !inside[qp]) {
            const bool gm = !inside[qm];  // which neighbour of the axis is the ghost (selects, no indexed arrays)
            const double val = psi_smooth_at(v, gm ? nk[qm] : nk[qp], gm ? nj[qm] : nj[qp], gm ? ni[qm] : ni[qp]);
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
        const int nch = C > 0 ? C : a.C;  // C == 0: channel count at run time (more than 4 channels)
#pragma unroll
        for (int c = 0; c < nch; c++) {
            const Rec<S, 12> fr = ldrec<S, 12>(a.A[c] + (size_t)blockIdx.y * a.vsA, c0);
            double f[12];
#pragma unroll
            for (int q = 0; q < 12; q++) f[q] = (double)fr.v[q];
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
        const Rec<S, 9> mr = {{(S)M11, (S)M22, (S)M33, (S)M12, (S)M13, (S)M23, (S)bu, (S)bv, (S)bw}};
        strec<S, 9>(a.M + (size_t)blockIdx.y * a.vsM, c0, mr);
        M11 = (double)(S)M11; M22 = (double)(S)M22; M33 = (double)(S)M33;
        M12 = (double)(S)M12; M13 = (double)(S)M13; M23 = (double)(S)M23;
        bu = (double)(S)bu; bv = (double)(S)bv; bw = (double)(S)bw;
    } else {
        const Rec<S, 9> mr = ldrec<S, 9>(a.M + (size_t)blockIdx.y * a.vsM, c0);
        M11 = (double)mr.v[0]; M22 = (double)mr.v[1]; M33 = (double)mr.v[2];
        M12 = (double)mr.v[3]; M13 = (double)mr.v[4]; M23 = (double)mr.v[5];
        bu = (double)mr.v[6]; bv = (double)mr.v[7]; bw = (double)mr.v[8];
    }
    const double den_u = den + M11, den_v = den + M22, den_w = den + M33;
    double n2 = num[0] - (bu + M12 * d0[1] + M13 * d0[2]);
    const double du1 = (1.0 - SM_OMEGA) * d0[0] + SM_OMEGA * (den_u != 0.0 ? n2 / den_u : 0.0);
    n2 = num[1] - (bv + M12 * du1 + M23 * d0[2]);
    const double dv1 = (1.0 - SM_OMEGA) * d0[1] + SM_OMEGA * (den_v != 0.0 ? n2 / den_v : 0.0);
    n2 = num[2] - (bw + M13 * du1 + M23 * dv1);
    const double dw1 = (1.0 - SM_OMEGA) * d0[2] + SM_OMEGA * (den_w != 0.0 ? n2 / den_w : 0.0);
    const Rec<S, 3> out = {{(S)du1, (S)dv1, (S)dw1}};
    strec<S, 3>(Dn, c0, out);
}

// One step n, four kinds of workgroups: P-stage tiles (psi_s^t on plane n - 4t, interior
// voxels), sweep tiles (iteration t on plane n - 2 - 4t, interior voxels), and for each of the two the
// surface voxels of the same planes packed 256 to a workgroup.  Within a step all four are independent:
// the sweep reads psi_s of planes finished in earlier steps, the P-stage increments swept in earlier steps.
// 4 waves per SIMD (128 VGPRs, ~150 B of scratch per lane): 273 ms per 256^3 volume against 297 ms at the
// 159 VGPRs / 3 waves the compiler picks by itself (fp32 storage, batch 8; on pitched rows the same kernel
// measured 254 / 274 ms, 286 ms at 5 waves and 447 ms at 6 -- the compact rows cost 7 % here and save 2.4x memory)
#ifndef SM_WPE
#define SM_WPE 4
#endif
#define SM_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SM_WPE, SM_WPE)))
#ifdef FR3D_EXPERIMENTS  // both stages in one kernel: the form of rounds 2-3, kept for FR3D_SM_DBG decompositions
template <typename S, int C>
__global__ void __launch_bounds__(256) SM_WPE_ATTR
k_smooth_step(const SmoothArgs<S> a, StepPart P, StepPart W, int cb, const int *__restrict__ meta,
              const int *__restrict__ kj)
{
    int b = blockIdx.x;
    SmoothPos p;
#ifdef FR3D_EXPERIMENTS  // FR3D_SM_DBG: 1 / 2 = no P-stage / sweep tiles, 4 / 8 = no P-stage / sweep surface workgroups
#define SM_SKIP(bit) if (a.dbg & (bit)) return;
#else
#define SM_SKIP(bit)
#endif
    if (b < P.ntiles) {
        SM_SKIP(1)
        if (locate_tile(a, b, P, p)) psi_voxel<S, true>(a, p);
        return;
    }
    b -= P.ntiles;
    if (b < W.ntiles) {
        SM_SKIP(2)
        if (locate_tile(a, b, W, p)) sweep_voxel<S, C, true>(a, p);
        return;
    }
    b -= W.ntiles;
    if (b < P.nt * cb) {
        SM_SKIP(4)
        if (locate_surface(a, b, cb, P, meta, kj, p)) psi_voxel<S, false>(a, p);
        return;
    }
    b -= P.nt * cb;
    SM_SKIP(8)
    if (locate_surface(a, b, cb, W, meta, kj, p)) sweep_voxel<S, C, false>(a, p);
}
#endif

// The stages as kernels of their own (same stream, round 4): the P-stage needs 64 VGPRs (8 waves per SIMD) where the
// sweep needs 110-128 (4 waves), and in one kernel both ran at the sweep's occupancy.
// (7 waves per SIMD -- 72 VGPRs, what the surface form asks for -- measured the same as 8)
// fp64 storage: 4 waves per SIMD (116 VGPRs; at 8 the kernel spills 196 B per lane: 1790 -> 1695 ms on the two-channel
// 256 x 512 x 512 case)
template <typename S>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(S) == 8 ? 4 : 8, sizeof(S) == 8 ? 4 : 8)))
k_smooth_psi_only(const SmoothArgs<S> a, StepPart P, int cb, const int *__restrict__ meta, const int *__restrict__ kj)
{
    int b = blockIdx.x;
    SmoothPos p;
    if (b < P.ntiles) {
        if (locate_tile(a, b, P, p)) psi_voxel<S, true>(a, p);
        return;
    }
    b -= P.ntiles;
    if (locate_surface(a, b, cb, P, meta, kj, p)) psi_voxel<S, false>(a, p);
}
// SURFACE: the surface workgroups instead of the tiles (a launch of its own: the tiles then run without the surface form's
// scratch, and the short launch fills in under the other engine lane -- two lanes 4.22 -> 4.38 volumes/s; the P-stage's
// tiles and surface workgroups in separate launches measured 268 against 251 ms and stay together)
// (the tiles at 5 waves per SIMD -- 96 VGPRs, 80 B of scratch -- measured 319 against 250 ms)
// The surface workgroups at 3 waves per SIMD: their form asks for 153 VGPRs, and under the tiles' cap of 128 it spilled
// 132 B per lane (249 -> 238 ms per volume).
// the sweep's tiles: 4 waves per SIMD for one channel with 4-byte storage (110 VGPRs), 3 for fp64 storage and for several
// channels (136-168 VGPRs: under the cap of 128 they spilled up to 228 B per lane; two channels with fp64 storage,
// 256 x 512 x 512: 2287 -> 1825 ms of solver time per volume, 512^3 one channel: 3920 -> 3234)
#define SM_TILE_WPE(S, C) ((sizeof(S) == 8 || (C) != 1) ? 3 : SM_WPE)
template <typename S, int C, bool SURFACE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SURFACE ? 3 : SM_TILE_WPE(S, C), SURFACE ? 3 : SM_TILE_WPE(S, C))))
k_smooth_sweep_only(const SmoothArgs<S> a, StepPart W, int cb, const int *__restrict__ meta, const int *__restrict__ kj)
{
    SmoothPos p;
    if (!SURFACE) {
        if (locate_tile(a, blockIdx.x, W, p)) sweep_voxel<S, C, true>(a, p);
    } else {
        if (locate_surface(a, blockIdx.x, cb, W, meta, kj, p)) sweep_voxel<S, C, false>(a, p);
    }
}

// One schedule (SM_LAG planes between iterations) serves both stages: step n runs the P-stage with
// tau = n and the sweep with tau = n - 2, i.e. psi_s^t is two planes ahead of sweep t.
template <typename S>
long long launch_sor_smooth(hipStream_t st, const SmoothArgs<S> &a, const SorSched &sc)
{
    if (a.iterations <= 0) return 0;
    FR3D_CHECK(sc.lag == SM_LAG && sc.bnd_kj && sc.bnd_meta, "internal: smooth solver needs the lag-4 schedule");
    const dim3 block(64, sc.by);
    FR3D_CHECK(sc.by == 4, "internal: smooth solver workgroups are 64 x 4");
    const int cb = cdiv(std::max(sc.bnd_max, 1), 256);
    long long launches = 0;
    const int last = (int)sc.launch_of_tau.size() - 1;
    auto part = [&](int tau) {
        StepPart p{0, 0, 0, 0, nullptr, nullptr};
        if (tau < 0 || tau > last || sc.launch_of_tau[tau] < 0) return p;
        const int l = sc.launch_of_tau[tau];
        p.tau = sc.tau[l]; p.t_lo = sc.t_lo[l]; p.nt = sc.nt[l]; p.ntiles = sc.ntiles[l];
        p.ent = sc.entries + sc.first[l];
        p.lut = sc.lut + sc.lut_first[l];
        return p;
    };
    bool two = true;
#ifdef FR3D_EXPERIMENTS
    static const char *one_env = getenv("FR3D_SMOOTH");  // "one": both stages in one kernel (the form of rounds 2-3; FR3D_SM_DBG)
    two = !(one_env && !strcmp(one_env, "one"));
#endif
    const int nv = a.nvol > 0 ? a.nvol : 1;
    for (int n = 0; n <= last + 2; n++) {
        const StepPart P = part(n), W = part(n - 2);
        const int blocks = P.ntiles + W.ntiles + (P.nt + W.nt) * cb;
        if (blocks <= 0) continue;
        if (two) {
            // three launches per step: the P-stage at 8 waves per SIMD, the sweep's tiles and its surface workgroups at 4
            // (250 against 274 ms per 256^3 volume for one launch per step)
            const int bp = P.ntiles + P.nt * cb;
            if (bp > 0) hipLaunchKernelGGL((k_smooth_psi_only<S>), dim3(bp, nv), block, 0, st, a, P, cb, sc.bnd_meta, sc.bnd_kj);
#define FR3D_SM_SWEEP(CC)                                                                                                       \
    {                                                                                                                            \
        if (W.ntiles > 0)                                                                                                        \
            hipLaunchKernelGGL((k_smooth_sweep_only<S, CC, false>), dim3(W.ntiles, nv), block, 0, st, a, W, cb, sc.bnd_meta, sc.bnd_kj); \
        if (W.nt * cb > 0)                                                                                                       \
            hipLaunchKernelGGL((k_smooth_sweep_only<S, CC, true>), dim3(W.nt * cb, nv), block, 0, st, a, W, cb, sc.bnd_meta, sc.bnd_kj); \
    }
            switch (a.C) {
                case 1: FR3D_SM_SWEEP(1); break;
                case 2: FR3D_SM_SWEEP(2); break;
                case 3: FR3D_SM_SWEEP(3); break;
                case 4: FR3D_SM_SWEEP(4); break;
                default:
                    FR3D_CHECK(a.C >= 1 && a.C <= FR3D_MAX_CHANNELS, "SOR kernel: channel count out of range");
                    FR3D_SM_SWEEP(0);
                    break;
            }
#undef FR3D_SM_SWEEP
            FR3D_LAUNCH_CHECK();
            launches++;
            continue;
        }
#ifdef FR3D_EXPERIMENTS
        const dim3 grid(blocks, nv);
        switch (a.C) {
            case 1: hipLaunchKernelGGL((k_smooth_step<S, 1>), grid, block, 0, st, a, P, W, cb, sc.bnd_meta, sc.bnd_kj); break;
            case 2: hipLaunchKernelGGL((k_smooth_step<S, 2>), grid, block, 0, st, a, P, W, cb, sc.bnd_meta, sc.bnd_kj); break;
            case 3: hipLaunchKernelGGL((k_smooth_step<S, 3>), grid, block, 0, st, a, P, W, cb, sc.bnd_meta, sc.bnd_kj); break;
            case 4: hipLaunchKernelGGL((k_smooth_step<S, 4>), grid, block, 0, st, a, P, W, cb, sc.bnd_meta, sc.bnd_kj); break;
            default:
                FR3D_CHECK(a.C >= 1 && a.C <= FR3D_MAX_CHANNELS, "SOR kernel: channel count out of range");
                hipLaunchKernelGGL((k_smooth_step<S, 0>), grid, block, 0, st, a, P, W, cb, sc.bnd_meta, sc.bnd_kj);
                break;
        }
        FR3D_LAUNCH_CHECK();
        launches++;
#endif
    }
    return launches;
}
template long long launch_sor_smooth<float>(hipStream_t, const SmoothArgs<float> &, const SorSched &);
template long long launch_sor_smooth<double>(hipStream_t, const SmoothArgs<double> &, const SorSched &);

#ifdef FR3D_EXPERIMENTS
// ---- fused form (round 4): P-stage and sweep tiles in ONE 512-thread workgroup ---------------------------------------
// With v = 2t for the P-stage of iteration t and v = 2t + 1 for its sweep, "virtual iteration" v works on hyperplane
// tau - 2v in launch tau -- the plane sweep's own pipeline with 2T iterations -- and the chain schedule of k_sor.hip
// (build_sor_chain_schedule, rows x chain positions = 4 x 2) pairs two consecutive v in one workgroup: position n takes
// rows k - n of plane s0 - 2n, so both positions read the SAME rows of the odd plane between them (the P-stage of plane q
// and the sweep of plane q - 2 share the rows of plane q - 1; a pair that starts at an odd v shares plane q - 3 between
// the sweep of t and the P-stage of t + 1).  The second reader finds them in the CU's L1 / the XCD's L2.  Dependences are
// those of the split form (psi_s^t two planes ahead of sweep t, sweep t two planes ahead of psi_s^{t+1}); the surface
// voxels keep their dense workgroups (512 lanes here).  Same per-voxel functions: bit-identical results.
// Measured (profiles/r04/smooth_fusion.md): HBM traffic 342 -> 299.5 B per update, and 7 % SLOWER on one lane (level on
// two): experiment build only (FR3D_SMOOTH=fused|paired), the split form ships.
template <typename S>
__device__ __forceinline__ bool locate_surface512(const SmoothArgs<S> &a, int b, int cb, const StepPart &P,
                                                  const int *__restrict__ meta, const int *__restrict__ kj, SmoothPos &p)
{
    p.t = P.t_lo + b / cb;
    const int s = P.tau - SM_LAG * p.t;
    if (s < 0 || s >= a.S_planes) return false;
    const int n = (b % cb) * 512 + (threadIdx.z * 4 + threadIdx.y) * 64 + threadIdx.x;
    if (n >= meta[2 * s + 1]) return false;
    const int v = kj[meta[2 * s] + n];
    return smooth_offsets(a, s, v >> 16, v & 0xffff, false, p);
}

// PAIRED: the two chain positions of a tile run as two 256-thread workgroups of their own, 8 workgroup ids apart --
// workgroups are dealt round-robin over the 8 XCDs, so the pair shares an L2 and is dispatched at about the same time --
// instead of one 512-thread workgroup (whose P-stage waves finish early and hold their slots until the sweep waves are done)
template <typename S, int C, bool PAIRED>
__global__ void __launch_bounds__(PAIRED ? 256 : 512) SM_WPE_ATTR
k_smooth_fused(const SmoothArgs<S> a, int tau, int v_lo, int nent, const SorEntry *__restrict__ ent,
               const int *__restrict__ lut, int ntiles, StepPart P, StepPart W, int cb, const int *__restrict__ meta,
               const int *__restrict__ kj)
{
    int b = blockIdx.x;
    SmoothPos p;
    const int nwg = PAIRED ? ((ntiles + 7) / 8) * 16 : ntiles;  // workgroup ids of the tiles
    if (b < nwg) {
        int n = __builtin_amdgcn_readfirstlane((int)threadIdx.z);
        if (PAIRED) {
            n = (b & 15) >> 3;
            b = (b >> 4) * 8 + (b & 7);
            if (b >= ntiles) return;
        }
        int lo = lut[b >> SOR_LUT_SHIFT];
        while (lo + 1 < nent && ent[lo + 1].pre <= b) lo++;
        const SorEntry en = ent[lo];
        const int local = b - en.pre;
        if (n >= sor_entry_nit(en)) return;
        const int v = v_lo + sor_entry_toff(en) + n;
        const int s = tau - 2 * v;
        const int k = (en.kb0 + local / en.njb) * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y) - n;
        if (k < 0 || s < 0 || s >= a.S_planes) return;
        const int jj = (local % en.njb) * 64 + threadIdx.x;
        p.t = v >> 1;
        if (!smooth_offsets(a, s, k, jj, true, p)) return;
        if (on_surface(a, p)) return;
        if (v & 1) sweep_voxel<S, C, true>(a, p);
        else psi_voxel<S, true>(a, p);
        return;
    }
    b -= nwg;
    if (b < P.nt * cb) {
        if (PAIRED ? locate_surface(a, b, cb, P, meta, kj, p) : locate_surface512(a, b, cb, P, meta, kj, p)) psi_voxel<S, false>(a, p);
        return;
    }
    b -= P.nt * cb;
    if (PAIRED ? locate_surface(a, b, cb, W, meta, kj, p) : locate_surface512(a, b, cb, W, meta, kj, p)) sweep_voxel<S, C, false>(a, p);
}

template <typename S>
long long launch_sor_smooth_fused(hipStream_t st, const SmoothArgs<S> &a, const SorSched &sc, const SorChainSched &ch, bool paired)
{
    if (a.iterations <= 0) return 0;
    FR3D_CHECK(sc.lag == SM_LAG && sc.bnd_kj && sc.bnd_meta, "internal: smooth solver needs the lag-4 schedule");
    FR3D_CHECK(ch.by == 4 && ch.nch == 2, "internal: the fused smooth solver runs 64 x 4 x 2 workgroups");
    const dim3 block(64, 4, paired ? 1 : 2);
    const int cb = cdiv(std::max(sc.bnd_max, 1), paired ? 256 : 512);
    long long launches = 0;
    const int last = (int)sc.launch_of_tau.size() - 1;
    auto part = [&](int tau) {
        StepPart p{0, 0, 0, 0, nullptr, nullptr};
        if (tau < 0 || tau > last || sc.launch_of_tau[tau] < 0) return p;
        const int l = sc.launch_of_tau[tau];
        p.tau = sc.tau[l]; p.t_lo = sc.t_lo[l]; p.nt = sc.nt[l]; p.ntiles = sc.ntiles[l];
        p.ent = sc.entries + sc.first[l];
        p.lut = sc.lut + sc.lut_first[l];
        return p;
    };
    for (size_t l = 0; l < ch.tau.size(); l++) {
        const int tau = ch.tau[l];
        const StepPart P = part(tau), W = part(tau - 2);
        const int nwg = paired ? ((ch.ntiles[l] + 7) / 8) * 16 : ch.ntiles[l];
        const int blocks = nwg + (P.nt + W.nt) * cb;
        if (blocks <= 0) continue;
        const dim3 grid(blocks, a.nvol > 0 ? a.nvol : 1);
#define FR3D_SM_FUSED(CC)                                                                                                   \
    if (paired)                                                                                                              \
        hipLaunchKernelGGL((k_smooth_fused<S, CC, true>), grid, block, 0, st, a, tau, ch.t_lo[l], ch.nent[l],                \
                           ch.entries + ch.first[l], ch.lut + ch.lut_first[l], ch.ntiles[l], P, W, cb, sc.bnd_meta, sc.bnd_kj); \
    else                                                                                                                     \
        hipLaunchKernelGGL((k_smooth_fused<S, CC, false>), grid, block, 0, st, a, tau, ch.t_lo[l], ch.nent[l],               \
                           ch.entries + ch.first[l], ch.lut + ch.lut_first[l], ch.ntiles[l], P, W, cb, sc.bnd_meta, sc.bnd_kj)
        switch (a.C) {
            case 1: FR3D_SM_FUSED(1); break;
            case 2: FR3D_SM_FUSED(2); break;
            case 3: FR3D_SM_FUSED(3); break;
            case 4: FR3D_SM_FUSED(4); break;
            default:
                FR3D_CHECK(a.C >= 1 && a.C <= FR3D_MAX_CHANNELS, "SOR kernel: channel count out of range");
                FR3D_SM_FUSED(0);
                break;
        }
#undef FR3D_SM_FUSED
        FR3D_LAUNCH_CHECK();
        launches++;
    }
    return launches;
}
template long long launch_sor_smooth_fused<float>(hipStream_t, const SmoothArgs<float> &, const SorSched &, const SorChainSched &, bool);
template long long launch_sor_smooth_fused<double>(hipStream_t, const SmoothArgs<double> &, const SorSched &, const SorChainSched &, bool);
#endif  // FR3D_EXPERIMENTS


}  // namespace fr3d
