// k_sor.hip -- K4+K6+K7: the lagged-nonlinearity SOR inner solver
// (core/level_solver_3d.py:314-546, a_smooth == 1 branch :472-493), lexicographic-exact.
//
// The reference sweeps voxels in lexicographic k->j->i order (true Gauss-Seidel: a voxel sees the
// NEW values of (k-1,j,i),(k,j-1,i),(k,j,i-1) and the OLD values of the +1 neighbours).  All voxels
// on a hyperplane s = i+j+k are mutually independent under that order, and iteration t+1 may
// process hyperplane s-2 while iteration t processes s.  One kernel launch therefore advances every
// in-flight iteration by one hyperplane:  launch tau handles {(t, s = tau - 2t)} -- up to
// `iterations` hyperplanes at once -- and reproduces the sequential sweep exactly (same
// neighbour states), with S + 2(iterations-1) launches per level instead of iterations*S.
//
// Data sits in the skewed layout of fr3d_internal.h (compact rows, one record per voxel and operand group), so a
// wave reads/writes contiguous j-runs and the six neighbours are row-uniform offsets.  The algorithmic traffic of
// the reference's update is 9C tensor entries + C (w*psi) + 3 Laplacian terms + 3 increments read, 3 written:
// (10C+9) values = 76 B for C = 1 with fp32 storage, 152 B with fp64 storage -- the figure bench.py prices the
// kernel against.  The kernel itself streams less: between psi updates the per-voxel 3x3 system (6+3 values,
// channels summed) is frozen, so ordinary iterations read a 9-value and seven 3-value records and write one
// 3-value record; psi-update iterations (every update_lag-th) read the 12C square-root factors + C weights + 3 L
// + the increments and write 9 + 3.
//
// Fusions: the Neumann ghost copy set_boundary_3d (:246-259) becomes "a missing neighbour is the
// voxel's own old value"; the psi_data update (:356-377) is pointwise in the old increment, so it
// is evaluated inside the sweep on iterations with t % update_lag == 0 and stored as w*psi;
// the u,v,w part of the stencil is iteration-invariant and precomputed once (k_laplace).
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>

#include "fr3d_internal.h"
#include "k_sor_core.h"

namespace fr3d {

#define SOR_BX 64
#define SOR_MAX_THREADS 512  // 64 lanes x rows x chain positions of a workgroup (8 waves: 2 x 4 or 4 x 2)

template <typename R, typename S, int C>
__global__ void __launch_bounds__(SOR_MAX_THREADS)
k_sor_step(const SorArgsT<S> a, int tau, int t_lo, int nent, const SorEntry *__restrict__ ent,
           const int *__restrict__ lut, int xg)
{
    const int Z = a.sk.Z, Y = a.sk.Y, X = a.sk.X;
    // blockIdx.x enumerates the tiles of all groups of this launch (schedule built on the host)
    const int vol = blockIdx.y;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids with equal id % 8 share one and its L2),
    // so in blockIdx order the row blocks of a plane that share halo rows never meet in one L2.  Within every run of
    // 8 xg tiles an XCD takes xg CONSECUTIVE tiles of the list (a few neighbouring row blocks), while all XCDs stay in
    // the same neighbourhood of memory; the tail of the list keeps its order; xg = 0: blockIdx order.
    // (One contiguous eighth of the list per XCD loses 5-7 % at 256^3: profiles/r03/sor_xcd_swizzle_ab.txt.)
    // (the XCD follows the LINEAR workgroup id, blockIdx.y * gridDim.x + blockIdx.x: the volumes of a lock-step batch
    // after the first start on a shifted XCD whenever the tile count is not a multiple of 8)
    int b = blockIdx.x;
    if (xg > 0) {
        const int run = 8 * xg, full = (int)(gridDim.x / run) * run;
        if (b < full) {
            const int off = (int)((blockIdx.y * gridDim.x) & 7u), p = b % run;
            b = (b / run) * run + ((p + off) & 7) * xg + (p >> 3);
        }
    }
    // find the group: the table gives the entry of the first tile of this tile group, a short
    // forward scan does the rest (a bisection costs ~7 dependent scalar loads before the first
    // vector load can be issued)
    int lo = lut[b >> SOR_LUT_SHIFT];
    while (lo + 1 < nent && ent[lo + 1].pre <= b) lo++;
    const SorEntry en = ent[lo];
    const int local = b - en.pre;
    // a wave is one row of one chain position (blockDim.x == 64): row and iteration are wave-uniform, which
    // keeps the row starts in SGPRs
    const int n = __builtin_amdgcn_readfirstlane((int)threadIdx.z);
    if (n >= sor_entry_nit(en)) return;
    const int t = t_lo + sor_entry_toff(en) + n;
    const int s = tau - 2 * t;
    const int k = (en.kb0 + local / en.njb) * (int)blockDim.y + __builtin_amdgcn_readfirstlane((int)threadIdx.y) - n;
    if (k < 0 || k >= Z) return;
    const int r = s - k;                       // i + j of this row
    if (r < 0 || r > X + Y - 2) return;
    const int jm0 = sk_jm(X, r);               // first valid j of the row (left-aligned storage)
    const int jj = (local % en.njb) * SOR_BX + threadIdx.x;
    const int j = jj + jm0;
    const int i = r - j;
    if (j >= Y || i < 0) return;               // i < X holds by construction of jm0

    // voxel (record) indices of the voxel and of its six neighbours inside one volume's arrays: row start
    // (wave-uniform) + lane.  Compact layout: rows of the planes s-1, s, s+1 start at pb[.] - cp[.] (see Skew);
    // the rows k-1 and k+1 of the neighbouring planes have the same i+j as this row, hence the same cp entry.
    // A ghost neighbour holds the voxel's own old value (set_boundary_3d): a missing neighbour
    // re-reads c0 (not yet overwritten), so all seven increment records are loaded unconditionally and
    // leave together instead of hiding behind exec-mask branches that wait for du0.
    long long b0, bm, bzm, bp, bzp;  // row starts: own, (s-1,k), (s-1,k-1), (s+1,k), (s+1,k+1)
    if (a.sk.pb) {
        const long long pm = a.sk.pb[s], p0 = a.sk.pb[s + 1], pp = a.sk.pb[s + 2];
        const long long cm = a.sk.cp[r], c_ = a.sk.cp[r + 1], cpn = a.sk.cp[r + 2];
        b0 = p0 - c_; bm = pm - cm; bzm = pm - c_; bp = pp - cpn; bzp = pp - c_;
    } else {
        const long long plane = a.sk.plane, Yp = a.sk.Yp;
        b0 = (long long)s * plane + (long long)k * Yp;
        bm = b0 - plane; bzm = bm - Yp; bp = b0 + plane; bzp = bp + Yp;
    }
    const long long c0 = b0 + jj;
    const int d1 = jm0 - sk_jm(X, r - 1), d2 = jm0 - sk_jm(X, r + 1);
    S *const D = a.d + vol * a.vsD;
    [[maybe_unused]] const long long xm = i > 0 ? bm + jj + d1 : c0;
    [[maybe_unused]] const long long xp = i < X - 1 ? bp + jj + d2 : c0;
    [[maybe_unused]] const long long ym = j > 0 ? bm + jj + d1 - 1 : c0;
    [[maybe_unused]] const long long yp = j < Y - 1 ? bp + jj + d2 + 1 : c0;
    const long long zm = k > 0 ? bzm + jj : c0;
    const long long zp = k < Z - 1 ? bzp + jj : c0;
    const bool upd = (t % a.update_lag) == 0;
    R du1, dv1, dw1;
#ifdef FR3D_SOR_PHASED
    constexpr bool phased = FR3D_SOR_PHASED != 0;
#else
    // Packed storage: a psi-update wave builds its system first and fetches the neighbours afterwards, an ordinary
    // wave has all its loads fenced into one group -- 82 instead of 92 VGPRs and 0.550 against 0.513 of the roofline
    // at 512^3 (same box, profiles/r03/sor_variants_ab.txt).  float / double storage: all loads up front, which
    // measured the same (fp32 storage) or 3 % better (fp64 storage) than the phase-ordered form.
    constexpr bool phased = std::is_same<S, pk42>::value;
#endif
    if constexpr (phased) {
    Rec<S, 3> q0 = ldrec<S, 3>(D, c0);
    Rec<S, 3> qxm, qxp, qym, qyp, qzm, qzp;
    R m[9];
    if (upd) {
        pin(q0);
        sor_system<R, S, C>(a, true, true, vol * a.vsM, vol * a.vsA, vol * a.vsL, c0, (R)q0.v[0], (R)q0.v[1], (R)q0.v[2], m);
        qxm = ldrec<S, 3>(D, xm); qxp = ldrec<S, 3>(D, xp);
        qym = ldrec<S, 3>(D, ym); qyp = ldrec<S, 3>(D, yp);
        qzm = ldrec<S, 3>(D, zm); qzp = ldrec<S, 3>(D, zp);
        pin(qxm, qxp, qym, qyp, qzm, qzp);
    } else {
        Rec<S, 9> mr = ldrec<S, 9>(a.M + vol * a.vsM, c0);
        qxm = ldrec<S, 3>(D, xm); qxp = ldrec<S, 3>(D, xp);
        qym = ldrec<S, 3>(D, ym); qyp = ldrec<S, 3>(D, yp);
        qzm = ldrec<S, 3>(D, zm); qzp = ldrec<S, 3>(D, zp);
        pin(q0, mr, qxm, qxp, qym, qyp, qzm, qzp);
#pragma unroll
        for (int q = 0; q < 9; q++) m[q] = (R)mr.v[q];
    }
    const R du0 = (R)q0.v[0], dv0 = (R)q0.v[1], dw0 = (R)q0.v[2];
    const R su_x = (R)qxm.v[0] + (R)qxp.v[0], sv_x = (R)qxm.v[1] + (R)qxp.v[1], sw_x = (R)qxm.v[2] + (R)qxp.v[2];
    const R su_y = (R)qym.v[0] + (R)qyp.v[0], sv_y = (R)qym.v[1] + (R)qyp.v[1], sw_y = (R)qym.v[2] + (R)qyp.v[2];
    const R su_z = (R)qzm.v[0] + (R)qzp.v[0], sv_z = (R)qzm.v[1] + (R)qzp.v[1], sw_z = (R)qzm.v[2] + (R)qzp.v[2];
#ifdef FR3D_EXPERIMENTS
    if (a.dbg & 16) {
        const R tiny = (R)1e-30;
        du1 = du0 + tiny * (su_x + su_y + su_z + m[0] + m[3] + m[6]);
        dv1 = dv0 + tiny * (sv_x + sv_y + sv_z + m[1] + m[4] + m[7]);
        dw1 = dw0 + tiny * (sw_x + sw_y + sw_z + m[2] + m[5] + m[8]);
    } else
#endif
    sor_relax_sel<R>(m, a.ax, a.ay, a.az, su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z, du0, dv0, dw0, du1, dv1,
                     dw1);
    } else {
    // All seven increment records are requested before anything else; a psi-update wave adds its factor loads
    // behind them.
    const Rec<S, 3> q0 = ldrec<S, 3>(D, c0);
    // Lane-shared neighbour rows (fp64 storage): (k,j,i-1) and (k,j-1,i) are records jj+d1 and jj+d1-1 of the SAME row
    // (s-1,k), (k,j,i+1) and (k,j+1,i) records jj+d2 and jj+d2+1 of row (s+1,k).  Each of the two rows is loaded once
    // per lane and the y-neighbour is the adjacent lane's x-neighbour (ds_bpermute; the active lanes of a wave are a
    // prefix: lane 0 and the last active lane fetch their missing record themselves): 5 wave-wide record loads instead
    // of 7.  The bytes that cross the fabric are the same; with 24-B records it measures +7.5 % at 512^3 (0.485 ->
    // 0.522), with 12-B records -6...-10 % and with packed 16-B records nothing (profiles/r03/sor_lane_share_ab.txt),
    // so only S = double takes it.
#ifdef FR3D_SOR_SHFL
    constexpr bool lane_share = FR3D_SOR_SHFL != 0;
#else
    // (one channel only: with more channels the psi branch's registers, not the loads, bound the wave -- 0.611 with
    // lane sharing against 0.629 without at 256^3, two channels)
    constexpr bool lane_share = std::is_same<S, double>::value && C == 1;
#endif
    Rec<S, 3> qxm, qxp, qym, qyp, qzm, qzp;
    if constexpr (lane_share) {
        // unconditional row loads: lanes whose neighbour does not exist read a record they do not use -- kept inside the slab
        const long long lastrec = a.sk.total - 1;
        auto inslab = [&](long long v) { return v < 0 ? 0LL : (v > lastrec ? lastrec : v); };
        const Rec<S, 3> rowm = ldrec<S, 3>(D, inslab(bm + jj + d1)), rowp = ldrec<S, 3>(D, inslab(bp + jj + d2));
        qzm = ldrec<S, 3>(D, zm);
        qzp = ldrec<S, 3>(D, zp);
        const unsigned long long act = __ballot(1);
        const int lane = (int)threadIdx.x;
        const bool last = ((act >> lane) >> 1) == 0;  // no active lane above this one
        Rec<S, 3> edge_m = q0, edge_p = q0;
        if (lane == 0 && j > 0) edge_m = ldrec<S, 3>(D, bm + jj + d1 - 1);
        if (last && j < Y - 1) edge_p = ldrec<S, 3>(D, bp + jj + d2 + 1);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const typename Sto<S>::val up = __shfl_up(rowm.v[q], 1), dn = __shfl_down(rowp.v[q], 1);
            qxm.v[q] = i > 0 ? rowm.v[q] : q0.v[q];
            qxp.v[q] = i < X - 1 ? rowp.v[q] : q0.v[q];
            qym.v[q] = j > 0 ? (lane == 0 ? edge_m.v[q] : up) : q0.v[q];
            qyp.v[q] = j < Y - 1 ? (last ? edge_p.v[q] : dn) : q0.v[q];
        }
    } else {
        qxm = ldrec<S, 3>(D, xm); qxp = ldrec<S, 3>(D, xp);
        qym = ldrec<S, 3>(D, ym); qyp = ldrec<S, 3>(D, yp);
        qzm = ldrec<S, 3>(D, zm); qzp = ldrec<S, 3>(D, zp);
    }
    const R du0 = (R)q0.v[0], dv0 = (R)q0.v[1], dw0 = (R)q0.v[2];
    const R su_x = (R)qxm.v[0] + (R)qxp.v[0], sv_x = (R)qxm.v[1] + (R)qxp.v[1], sw_x = (R)qxm.v[2] + (R)qxp.v[2];
    const R su_y = (R)qym.v[0] + (R)qyp.v[0], sv_y = (R)qym.v[1] + (R)qyp.v[1], sw_y = (R)qym.v[2] + (R)qyp.v[2];
    const R su_z = (R)qzm.v[0] + (R)qzp.v[0], sv_z = (R)qzm.v[1] + (R)qzp.v[1], sw_z = (R)qzm.v[2] + (R)qzp.v[2];
    R m[9];
    sor_system<R, S, C>(a, upd, true, vol * a.vsM, vol * a.vsA, vol * a.vsL, c0, du0, dv0, dw0, m);
#ifdef FR3D_EXPERIMENTS
    if (a.dbg & 16) {  // memory skeleton: every load and store of the update, none of its arithmetic (results are garbage)
        const R tiny = (R)1e-30;
        du1 = du0 + tiny * (su_x + su_y + su_z + m[0] + m[3] + m[6]);
        dv1 = dv0 + tiny * (sv_x + sv_y + sv_z + m[1] + m[4] + m[7]);
        dw1 = dw0 + tiny * (sw_x + sw_y + sw_z + m[2] + m[5] + m[8]);
    } else
#endif
    sor_relax<R>(m, a.ax, a.ay, a.az, su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z, du0, dv0, dw0, du1, dv1,
                 dw1);
    }
    Rec<S, 3> out;
    out.v[0] = Sto<S>::quant(du1);
    out.v[1] = Sto<S>::quant(dv1);
    out.v[2] = Sto<S>::quant(dw1);
#ifdef FR3D_EXPERIMENTS
    if (a.dbg & 1) {
        using V = typename Sto<S>::val;
        out.v[0] = (V)(float)du1; out.v[1] = (V)(float)dv1; out.v[2] = (V)(float)dw1;
    }
    if (a.dbg & 128) bfp_round(out.v, 0, 3, (a.dbg >> 8) & 0xff);  // du, dv, dw with one exponent
#endif
    strec<S, 3>(D, c0, out);
}

// Tiles an XCD takes in a row (k_sor_step's xg).  Measured, alternating on one box (profiles/r03/sor_xcd_group_ab.txt):
// at 512^3 (rows of up to 512 lanes, 8 tiles per row block) 32 takes the sweep from 0.525 to 0.558 of the roofline with
// packed storage and from 0.518 to 0.538 with fp64 storage (8: 0.547, 16: 0.55, 64: 0.555); fp32 storage: no gain;
// at 256^3 every group size LOSES 4 % (0.505 -> 0.485 packed, 0.484 -> 0.462 fp32 storage).  So: levels whose rows
// reach 384 lanes, packed and fp64 storage.
template <typename S>
static int sor_xcd_group(const Skew &sk)
{
    int g = (!std::is_same<S, float>::value && std::min(sk.X, sk.Y) >= 384) ? 32 : 0;
#ifdef FR3D_EXPERIMENTS
    if (const char *env = getenv("FR3D_SOR_XCD_G"))
        if (atoi(env) >= 0) g = atoi(env);  // negative: the rule above
#endif
    return g;
}

template <typename R, typename S>
static void launch_step(hipStream_t st, const SorArgsT<S> &a, int tau, int t_lo, const SorChainSched &sc, size_t l)
{
    const int ntiles = sc.ntiles[l];
    if (ntiles <= 0) return;
    const SorEntry *ent = sc.entries + sc.first[l];
    const int *lut = sc.lut + sc.lut_first[l];
    const int nent = sc.nent[l];
    const dim3 grid(ntiles, a.nvol > 0 ? a.nvol : 1), block(SOR_BX, sc.by, sc.nch);
    const int xg = sor_xcd_group<S>(a.sk);
#define FR3D_SOR_CASE(CH)                                                                                     \
    case CH: hipLaunchKernelGGL((k_sor_step<R, S, CH>), grid, block, 0, st, a, tau, t_lo, nent, ent, lut, xg); break;
    switch (a.C) {
        FR3D_SOR_CASE(1)
        FR3D_SOR_CASE(2)
        FR3D_SOR_CASE(3)
        FR3D_SOR_CASE(4)
        // 5..FR3D_MAX_CHANNELS channels: one instantiation with the channel loop bound read at run time
        // (the loop already handles one channel at a time, k_sor_core.h; level_solver_3d.py:356-377 loops over any C)
        default:
            FR3D_CHECK(a.C >= 1 && a.C <= FR3D_MAX_CHANNELS, "SOR kernel: channel count out of range");
            hipLaunchKernelGGL((k_sor_step<R, S, 0>), grid, block, 0, st, a, tau, t_lo, nent, ent, lut, xg);
            break;
    }
#undef FR3D_SOR_CASE
    FR3D_LAUNCH_CHECK();
}

void sor_tile_shape(const Skew &sk, int &by, int &nch)
{
    (void)sk;
    by = 2;   // 64 lanes x 2 rows
    nch = 1;  // chain positions: see DESIGN.md section 4 (chains cut the fetched bytes by 9 % and lost 10 % of the rate)
#ifdef FR3D_EXPERIMENTS
    if (const char *env = getenv("FR3D_SOR_SHAPE")) {  // "<rows>x<chain>", e.g. 2x4
        int b = 0, c = 0;
        if (sscanf(env, "%dx%d", &b, &c) == 2 && (b == 1 || b == 2 || b == 4 || b == 8) && c >= 1 && c <= 8) {
            by = b;
            nch = c;
        }
    }
#endif
    while (SOR_BX * by * nch > SOR_MAX_THREADS) nch--;
}

// rows [klo, khi] of hyperplane s that hold voxels, and the longest of them
static inline bool plane_rows(const Skew &sk, int s, int &klo, int &khi)
{
    klo = std::max(0, s - (sk.X - 1) - (sk.Y - 1));
    khi = std::min(sk.Z - 1, s);
    return s >= 0 && s < sk.S && klo <= khi;
}
static inline int plane_maxlen(const Skew &sk, int s, int klo, int khi)
{
    // len(r) = min(Y-1, r) - jm(r) + 1 rises to min(X,Y) and falls again: evaluate the ends and the plateau
    int maxlen = 0;
    for (int k = klo; k <= khi; k++) {
        const int r = s - k;
        const int len = std::min(sk.Y - 1, r) - sk_jm(sk.X, r) + 1;
        if (len > maxlen) maxlen = len;
    }
    return maxlen;
}

// host side of the schedule: entries and group tables of every launch (no device calls)
void make_chain_entries(const Skew &sk, int T, int by, int nch, SorChainSched &sc, std::vector<SorEntry> &ent,
                        std::vector<int> &lut)
{
    sc.by = by;
    sc.nch = nch;
    if (T <= 0) return;
    FR3D_CHECK(T <= 32767, "SOR schedule: more than 32767 iterations");
    FR3D_CHECK(by >= 1 && nch >= 1 && SOR_BX * by * nch <= SOR_MAX_THREADS, "SOR schedule: bad tile shape");
    const int S = sk.S;
    const int last = (S - 1) + 2 * (T - 1);
    for (int tau = 0; tau <= last; tau++) {
        int t_lo = tau - (S - 1);
        t_lo = t_lo <= 0 ? 0 : (t_lo + 1) / 2;
        int t_hi = tau / 2;
        if (t_hi > T - 1) t_hi = T - 1;
        if (t_lo > t_hi) continue;
        sc.tau.push_back(tau);
        sc.t_lo.push_back(t_lo);
        sc.first.push_back((int)ent.size());
        int pre = 0, nent = 0;
        for (int t = t_lo; t <= t_hi;) {
            // the group: a run of up to nch consecutive iterations
            const int n = std::min(nch, t_hi - t + 1);
            // tile rows in the coordinates of chain position 0: position q handles row k - q of plane s0 - 2q
            int kmin = INT_MAX, kmax = INT_MIN, maxlen = 0;
            for (int q = 0; q < n; q++) {
                int klo, khi;
                const int s = tau - 2 * (t + q);
                if (!plane_rows(sk, s, klo, khi)) continue;
                kmin = std::min(kmin, klo + q);
                kmax = std::max(kmax, khi + q);
                maxlen = std::max(maxlen, plane_maxlen(sk, s, klo, khi));
            }
            if (maxlen > 0) {
                SorEntry e;
                e.pre = pre;
                e.kb0 = kmin / by;
                e.njb = cdiv(maxlen, SOR_BX);
                e.tn = (t - t_lo) | (n << 16);
                pre += (kmax / by - kmin / by + 1) * e.njb;
                ent.push_back(e);
                nent++;
            }
            t += n;
        }
        sc.nent.push_back(nent);
        sc.ntiles.push_back(pre);
        // group table of this launch
        sc.lut_first.push_back((int)lut.size());
        const size_t e0 = (size_t)sc.first.back();
        int cur = 0;
        for (int g = 0; (g << SOR_LUT_SHIFT) < pre; g++) {
            const int b0 = g << SOR_LUT_SHIFT;
            while (cur + 1 < nent && ent[e0 + cur + 1].pre <= b0) cur++;
            lut.push_back(cur);
        }
    }
}

SorChainSched build_sor_chain_schedule(const Skew &sk, int T, int by, int nch)
{
    SorChainSched sc;
    std::vector<SorEntry> ent;
    std::vector<int> lut;
    make_chain_entries(sk, T, by, nch, sc, ent, lut);
    if (T <= 0) return sc;
    FR3D_HIP(hipMalloc((void **)&sc.entries, std::max<size_t>(ent.size(), 1) * sizeof(SorEntry)));
    FR3D_HIP(hipMemcpy(sc.entries, ent.data(), ent.size() * sizeof(SorEntry), hipMemcpyHostToDevice));
    FR3D_HIP(hipMalloc((void **)&sc.lut, std::max<size_t>(lut.size(), 1) * sizeof(int)));
    FR3D_HIP(hipMemcpy(sc.lut, lut.data(), lut.size() * sizeof(int), hipMemcpyHostToDevice));
    return sc;
}

// Host replay of the kernel's index arithmetic over a whole schedule (no device involved): every voxel update
// (t, k, j, i) must be issued exactly once, by the launch tau = i + j + k + 2t.  Returns the number of violations
// (0 = consistent) and the number of updates issued.
long long check_chain_schedule(int Z, int Y, int X, int T, int by, int nch, long long *n_updates)
{
    const Skew sk = make_skew(Z, Y, X);
    SorChainSched sc;
    std::vector<SorEntry> ent;
    std::vector<int> lut;
    make_chain_entries(sk, T, by, nch, sc, ent, lut);
    const size_t nv = (size_t)Z * Y * X;
    std::vector<unsigned char> seen((size_t)std::max(T, 0) * nv, 0);
    long long bad = 0, total = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        const SorEntry *en_ = ent.data() + sc.first[l];
        const int *lu = lut.data() + sc.lut_first[l];
        const int nent = sc.nent[l], tau = sc.tau[l], t_lo = sc.t_lo[l];
        for (int b = 0; b < sc.ntiles[l]; b++) {
            int lo = lu[b >> SOR_LUT_SHIFT];
            while (lo + 1 < nent && en_[lo + 1].pre <= b) lo++;
            const SorEntry en = en_[lo];
            const int local = b - en.pre;
            if (local < 0) { bad++; continue; }
            for (int n = 0; n < nch; n++)
                for (int ty = 0; ty < by; ty++) {
                    if (n >= sor_entry_nit(en)) continue;
                    const int t = t_lo + sor_entry_toff(en) + n;
                    if (t < 0 || t >= T) { bad++; continue; }
                    const int s = tau - 2 * t;
                    const int k = (en.kb0 + local / en.njb) * by + ty - n;
                    if (k < 0 || k >= Z) continue;
                    const int r = s - k;
                    if (r < 0 || r > X + Y - 2) continue;
                    const int jm0 = sk_jm(X, r);
                    for (int lane = 0; lane < SOR_BX; lane++) {
                        const int jj = (local % en.njb) * SOR_BX + lane;
                        const int j = jj + jm0, i = r - j;
                        if (j >= Y || i < 0) continue;
                        if (i >= X) { bad++; continue; }
                        unsigned char &c = seen[(size_t)t * nv + ((size_t)k * Y + j) * X + i];
                        if (c) bad++;
                        c = 1;
                        total++;
                    }
                }
        }
    }
    for (unsigned char c : seen) bad += c ? 0 : 1;
    if (n_updates) *n_updates = total;
    return bad;
}

void free_sor_chain_schedule(SorChainSched &s)
{
    if (s.entries) (void)hipFree(s.entries);
    if (s.lut) (void)hipFree(s.lut);
    s.entries = nullptr;
    s.lut = nullptr;
}

SorSched build_sor_schedule(const Skew &sk, int T, int by, int lag)
{
    SorSched sc;
    sc.by = by;
    sc.lag = lag;
    const int S = sk.S, Z = sk.Z, Y = sk.Y, X = sk.X;
    std::vector<SorEntry> ent;
    std::vector<int> lut;
    if (T <= 0) return sc;
    const int last = (S - 1) + lag * (T - 1);
    sc.launch_of_tau.assign((size_t)last + 1, -1);
    for (int tau = 0; tau <= last; tau++) {
        int t_lo = tau - (S - 1);
        t_lo = t_lo <= 0 ? 0 : (t_lo + lag - 1) / lag;
        int t_hi = tau / lag;
        if (t_hi > T - 1) t_hi = T - 1;
        if (t_lo > t_hi) continue;
        sc.launch_of_tau[tau] = (int)sc.tau.size();
        sc.tau.push_back(tau);
        sc.t_lo.push_back(t_lo);
        sc.nt.push_back(t_hi - t_lo + 1);
        sc.first.push_back((int)ent.size());
        int pre = 0;
        for (int t = t_lo; t <= t_hi; t++) {
            const int s = tau - lag * t;
            // valid rows of hyperplane s: k in [klo,khi]; row (s,k) holds r = s-k, j in
            // [jm(r), min(Y-1,r)], stored left-aligned, so tiles start at 0 and the row count of
            // j-tiles is set by the longest row
            const int klo = std::max(0, s - (X - 1) - (Y - 1)), khi = std::min(Z - 1, s);
            SorEntry e;
            e.pre = pre;
            e.tn = (t - t_lo) | (1 << 16);
            e.kb0 = 0;
            e.njb = 1;
            if (klo <= khi) {
                int maxlen = 0;
                for (int k = klo; k <= khi; k++) {
                    const int r = s - k;
                    const int len = std::min(Y - 1, r) - sk_jm(X, r) + 1;
                    if (len > maxlen) maxlen = len;
                }
                if (maxlen > 0) {
                    const int kb0 = klo / by, kb1 = khi / by;
                    e.kb0 = kb0;
                    e.njb = cdiv(maxlen, SOR_BX);
                    pre += (kb1 - kb0 + 1) * e.njb;
                }
            }
            ent.push_back(e);
        }
        sc.ntiles.push_back(pre);
        // group table of this launch
        sc.lut_first.push_back((int)lut.size());
        const size_t e0 = (size_t)sc.first.back();
        const int nte = t_hi - t_lo + 1;
        int cur = 0;
        for (int g = 0; (g << SOR_LUT_SHIFT) < pre; g++) {
            const int b0 = g << SOR_LUT_SHIFT;
            while (cur + 1 < nte && ent[e0 + cur + 1].pre <= b0) cur++;
            lut.push_back(cur);
        }
    }
    FR3D_HIP(hipMalloc((void **)&sc.entries, ent.size() * sizeof(SorEntry)));
    FR3D_HIP(hipMemcpy(sc.entries, ent.data(), ent.size() * sizeof(SorEntry), hipMemcpyHostToDevice));
    FR3D_HIP(hipMalloc((void **)&sc.lut, std::max<size_t>(lut.size(), 1) * sizeof(int)));
    FR3D_HIP(hipMemcpy(sc.lut, lut.data(), lut.size() * sizeof(int), hipMemcpyHostToDevice));
    if (lag != 2) {
        FR3D_CHECK(Z <= 65535 && Y <= 65535, "a_smooth != 1 solver: axis longer than 65535");
        std::vector<int> kj;
        sc.bnd_first.assign(S, 0);
        sc.bnd_count.assign(S, 0);
        for (int s = 0; s < S; s++) {
            sc.bnd_first[s] = (int)kj.size();
            const int klo = std::max(0, s - (X - 1) - (Y - 1)), khi = std::min(Z - 1, s);
            for (int k = klo; k <= khi; k++) {
                const int r = s - k;
                const int jlo = sk_jm(X, r), jhi = std::min(Y - 1, r);
                for (int j = jlo; j <= jhi; j++) {
                    const int i = r - j;
                    if (k == 0 || k == Z - 1 || j == 0 || j == Y - 1 || i == 0 || i == X - 1) kj.push_back((k << 16) | j);
                    else if (j < jhi - 1) j = jhi - 1;  // interior of a row: jump to its last voxels
                }
            }
            sc.bnd_count[s] = (int)kj.size() - sc.bnd_first[s];
            sc.bnd_max = std::max(sc.bnd_max, sc.bnd_count[s]);
        }
        FR3D_HIP(hipMalloc((void **)&sc.bnd_kj, std::max<size_t>(kj.size(), 1) * sizeof(int)));
        FR3D_HIP(hipMemcpy(sc.bnd_kj, kj.data(), kj.size() * sizeof(int), hipMemcpyHostToDevice));
        std::vector<int> meta(2 * (size_t)S);
        for (int s = 0; s < S; s++) {
            meta[2 * s] = sc.bnd_first[s];
            meta[2 * s + 1] = sc.bnd_count[s];
        }
        FR3D_HIP(hipMalloc((void **)&sc.bnd_meta, meta.size() * sizeof(int)));
        FR3D_HIP(hipMemcpy(sc.bnd_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return sc;
}

void free_sor_schedule(SorSched &s)
{
    if (s.entries) (void)hipFree(s.entries);
    if (s.lut) (void)hipFree(s.lut);
    if (s.bnd_kj) (void)hipFree(s.bnd_kj);
    if (s.bnd_meta) (void)hipFree(s.bnd_meta);
    s.bnd_kj = nullptr;
    s.bnd_meta = nullptr;
    s.entries = nullptr;
    s.lut = nullptr;
}

template <typename S>
long long launch_sor(hipStream_t st, const SorArgsT<S> &a_in, bool fp64, const SorChainSched &sc)
{
    SorArgsT<S> a = a_in;
    a.dbg = 0;
#ifdef FR3D_EXPERIMENTS
    if (const char *dbg_env = getenv("FR3D_SOR_DBG")) a.dbg = atoi(dbg_env);  // numerics experiments, see SorArgsT::dbg
#endif
    long long launches = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        if (sc.ntiles[l] <= 0) continue;
        if constexpr (Sto<S>::wide) {
            launch_step<double, S>(st, a, sc.tau[l], sc.t_lo[l], sc, l);
        } else {
            if (fp64) launch_step<double, S>(st, a, sc.tau[l], sc.t_lo[l], sc, l);
            else launch_step<float, S>(st, a, sc.tau[l], sc.t_lo[l], sc, l);
        }
        launches++;
    }
    return launches;
}
template long long launch_sor<float>(hipStream_t, const SorArgsT<float> &, bool, const SorChainSched &);
template long long launch_sor<double>(hipStream_t, const SorArgsT<double> &, bool, const SorChainSched &);
template long long launch_sor<pk42>(hipStream_t, const SorArgsT<pk42> &, bool, const SorChainSched &);

// ---- layout conversion and the iteration-invariant stencil part -------------------------------

// NREC natural planar arrays -> one skewed array of NREC-value records (and back), through an LDS tile of
// TY x 32 voxels of one z-slice: the natural side moves as 128-B row segments, the skewed side as whole records
// along the tile's anti-diagonals (x+y constant => same hyperplane and row, consecutive j => consecutive
// records), so both sides are coalesced.
#define PKX 32
template <typename TS, typename TD, int NREC, int TY, bool TO_SKEW>
__global__ void __launch_bounds__(256)
k_skew_pack(const TS *__restrict__ src, long long src_stride, TD *__restrict__ dst, const Skew sk)
{
    const int Y = sk.Y, X = sk.X;
    // pitch 34: element (ly, d - ly) of a diagonal sits at 33*ly + d -> consecutive banks for consecutive ly
    // (tile_diag pairs a short diagonal with the one 32 further on: every step fills its 32 lanes)
    __shared__ TD tile[NREC][TY][PKX + 2];
    const int txn = (X + PKX - 1) / PKX;
    const int x0 = (blockIdx.x % txn) * PKX, y0 = (blockIdx.x / txn) * TY;
    const int z = blockIdx.y;
    const int lane = threadIdx.x % PKX, grp = threadIdx.x / PKX;  // 8 groups of 32
    if constexpr (TO_SKEW) {
#pragma unroll
        for (int a = 0; a < NREC; a++) {
            TS v[(TY + 7) / 8];
#pragma unroll
            for (int q = 0; q < (TY + 7) / 8; q++) {
                const int ly = grp + 8 * q, y = y0 + ly, x = x0 + lane;
                v[q] = (ly < TY && y < Y && x < X) ? src[a * src_stride + ((size_t)z * Y + y) * X + x] : (TS)0;
            }
#pragma unroll
            for (int q = 0; q < (TY + 7) / 8; q++)
                if (grp + 8 * q < TY) tile[a][grp + 8 * q][lane] = (TD)v[q];
        }
        __syncthreads();
        for (int m = grp; m < TY; m += 8) {
            int ly, lx;
            tile_diag<TY>(lane, m, ly, lx);
            const int y = y0 + ly, x = x0 + lx;
            if (y < Y && x < X) {
                TD *o = dst + (size_t)sk_index(sk, z, y, x) * NREC;
#pragma unroll
                for (int a = 0; a < NREC; a++) o[a] = tile[a][ly][lx];
            }
        }
    } else {
        for (int m = grp; m < TY; m += 8) {
            int ly, lx;
            tile_diag<TY>(lane, m, ly, lx);
            const int y = y0 + ly, x = x0 + lx;
            if (y < Y && x < X) {
                const Rec<TS, NREC> o = ldrec<TS, NREC>(src, sk_index(sk, z, y, x));
#pragma unroll
                for (int a = 0; a < NREC; a++) tile[a][ly][lx] = (TD)o.v[a];
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NREC; a++)
#pragma unroll
            for (int q = 0; q < (TY + 7) / 8; q++) {
                const int ly = grp + 8 * q, y = y0 + ly, x = x0 + lane;
                if (ly < TY && y < Y && x < X) dst[a * src_stride + ((size_t)z * Y + y) * X + x] = tile[a][ly][lane];
            }
    }
}

template <typename TS, typename TD, int NREC, bool TO_SKEW>
static void launch_pack_t(hipStream_t st, const TS *src, long long stride, TD *dst, const Skew &sk)
{
    // LDS per workgroup: NREC * TY * 34 values; 12 fp64 values per voxel need the 16-row tile
    constexpr int TY = (sizeof(TD) * NREC > 48) ? 16 : 32;
    FR3D_CHECK(sk.Z <= 65535, "skew transposes: z axis longer than 65535");
    dim3 grid(cdiv(sk.X, PKX) * cdiv(sk.Y, TY), sk.Z);
    hipLaunchKernelGGL((k_skew_pack<TS, TD, NREC, TY, TO_SKEW>), grid, dim3(256), 0, st, src, stride, dst, sk);
    FR3D_LAUNCH_CHECK();
}

template <typename TS, typename TD>
void launch_skew_pack(hipStream_t st, const TS *src, long long src_stride, TD *dst, int nrec, const Skew &sk)
{
    if (nrec == 1) launch_pack_t<TS, TD, 1, true>(st, src, src_stride, dst, sk);
    else if (nrec == 3) launch_pack_t<TS, TD, 3, true>(st, src, src_stride, dst, sk);
    else if (nrec == 12) launch_pack_t<TS, TD, 12, true>(st, src, src_stride, dst, sk);
    else if (nrec == 10) {
        if constexpr (std::is_same<TS, double>::value && std::is_same<TD, double>::value)  // verification mode: tensor entries
            launch_pack_t<TS, TD, 10, true>(st, src, src_stride, dst, sk);
        else throw Error("internal: records of 10 values are fp64");
    } else throw Error("internal: records of 1, 3, 10 or 12 values");
}
template <typename TS, typename TD>
void launch_unskew_unpack(hipStream_t st, const TS *src, TD *dst, long long dst_stride, int nrec, const Skew &sk)
{
    if (nrec == 3) launch_pack_t<TS, TD, 3, false>(st, src, dst_stride, dst, sk);
    else throw Error("internal: records of 3 values");
}
template void launch_skew_pack<float, float>(hipStream_t, const float *, long long, float *, int, const Skew &);
template void launch_skew_pack<double, double>(hipStream_t, const double *, long long, double *, int, const Skew &);
template void launch_skew_pack<float, double>(hipStream_t, const float *, long long, double *, int, const Skew &);
template void launch_unskew_unpack<float, float>(hipStream_t, const float *, float *, long long, int, const Skew &);
template void launch_unskew_unpack<double, float>(hipStream_t, const double *, float *, long long, int, const Skew &);
template void launch_unskew_unpack<pk42, float>(hipStream_t, const pk42 *, float *, long long, int, const Skew &);
template void launch_unskew_unpack<double, double>(hipStream_t, const double *, double *, long long, int, const Skew &);
template void launch_unskew_unpack<pk42, double>(hipStream_t, const pk42 *, double *, long long, int, const Skew &);

// L = ax*(u_ip + u_im - 2u) + ay*(...) + az*(...) with edge-padded u (add_boundary,
// core/optical_flow_3d.py:88), evaluated in fp64 from the fp32-exact level flow.
// The same three terms written as one record per voxel in the skewed voxel order of `sk` (compact or pitched),
// through an LDS tile like k_skew_pack / k_motion_tensor_rec.
template <typename TL>
__global__ void __launch_bounds__(256)
k_laplace_rec(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w, const Skew sk,
              double ax, double ay, double az, TL *__restrict__ dst)
{
    __shared__ typename Sto<TL>::val tile[3][32][PKX + 2];
    const int Z = sk.Z, Y = sk.Y, X = sk.X;
    const int txn = (X + PKX - 1) / PKX;
    const int x0 = (blockIdx.x % txn) * PKX, y0 = (blockIdx.x / txn) * 32;
    const int z = blockIdx.y;
    const int lane = threadIdx.x % PKX, grp = threadIdx.x / PKX;
    const long long sx = 1, sy = X, sz = (long long)Y * X;
    const float *f[3] = {u, v, w};
    for (int ly = grp; ly < 32; ly += 8) {
        const int y = y0 + ly, x = x0 + lane;
        if (y < Y && x < X) {
            const long long t = ((long long)z * Y + y) * X + x;
            const long long xm = x > 0 ? -sx : 0, xp = x < X - 1 ? sx : 0;
            const long long ym = y > 0 ? -sy : 0, yp = y < Y - 1 ? sy : 0;
            const long long zm = z > 0 ? -sz : 0, zp = z < Z - 1 ? sz : 0;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float *q = f[d] + t;
                double c = (double)q[0];
                double acc = ax * ((double)q[xp] + (double)q[xm] - 2.0 * c);
                acc += ay * ((double)q[yp] + (double)q[ym] - 2.0 * c);
                acc += az * ((double)q[zp] + (double)q[zm] - 2.0 * c);
                tile[d][ly][lane] = Sto<TL>::quant(acc);
            }
        }
    }
    __syncthreads();
    for (int m = grp; m < 32; m += 8) {
        int ly, lx;
        tile_diag<32>(lane, m, ly, lx);
        const int y = y0 + ly, x = x0 + lx;
        if (y < Y && x < X) {
            Rec<TL, 3> o;
            o.v[0] = tile[0][ly][lx];
            o.v[1] = tile[1][ly][lx];
            o.v[2] = tile[2][ly][lx];
            strec<TL, 3>(dst, sk_index(sk, z, y, x), o);
        }
    }
}

template <typename TL>
void launch_laplace_rec(hipStream_t st, const float *u, const float *v, const float *w, const Skew &sk, double ax,
                        double ay, double az, TL *dst)
{
    FR3D_CHECK(sk.Z <= 65535, "laplace: z axis longer than 65535");
    dim3 grid(cdiv(sk.X, PKX) * cdiv(sk.Y, 32), sk.Z);
    hipLaunchKernelGGL(k_laplace_rec<TL>, grid, dim3(256), 0, st, u, v, w, sk, ax, ay, az, dst);
    FR3D_LAUNCH_CHECK();
}
template void launch_laplace_rec<float>(hipStream_t, const float *, const float *, const float *, const Skew &, double,
                                        double, double, float *);
template void launch_laplace_rec<double>(hipStream_t, const float *, const float *, const float *, const Skew &, double,
                                         double, double, double *);
template void launch_laplace_rec<pk42>(hipStream_t, const float *, const float *, const float *, const Skew &, double,
                                       double, double, pk42 *);

}  // namespace fr3d
