// k_sor_pair.hip -- the a_smooth == 1 SOR sweep (core/level_solver_3d.py:314-546), two hyperplanes
// of one iteration per launch.  Same arithmetic, same lexicographic Gauss-Seidel order and therefore
// bit-identical results to k_sor_step (k_sor.hip, k_sor_core.h); what changes is how often the
// increments travel through HBM.
//
// k_sor_step relaxes ONE hyperplane s = i+j+k of every in-flight iteration per launch and reads, per
// voxel update, its own old increment and six neighbours: the planes s-1 (new) and s+1 (old) are
// fetched next to the voxel's own plane, 3 plane-reads per plane relaxed.  Here a workgroup owns a
// tile (ROWS rows x 64 lanes of the skewed layout) for the planes s AND s+1 of iteration t:
//   phase 1  relaxes plane s   (minus-neighbours: plane s-1 from HBM, plus-neighbours: plane s+1 old),
//            keeps the new increments in LDS and stores the tile's own voxels;
//   barrier
//   phase 2  relaxes plane s+1 (minus-neighbours: phase 1's values from LDS, plus-neighbours:
//            plane s+2 old from HBM).
// 4 plane-reads per 2 planes relaxed instead of 6.  A voxel (k,j,i') of plane s+1 needs the new values
// of (k,j,i'-1), (k,j-1,i') and (k-1,j,i') -- in lane coordinates of the left-aligned rows: the same
// row at lanes jj+dl and jj+dl-1 (dl = 0/1, row-uniform) and row k-1 at lane jj -- so phase 1 also
// relaxes, without storing them, one halo row above the tile (wave 0) and one halo lane per row
// (left of the tile where dl = 0, right of it where dl = 1; the last wave, one lane per row).  The
// neighbouring tile computes the same values from the same inputs, so nothing depends on which
// workgroup runs first.
// Because planes s and s+1 are rewritten while neighbouring tiles still read their old values, the
// increments are double-buffered by iteration parity: iteration t reads "old" from D[(t+1)&1] and
// "new" (plane s-1) from D[t&1], and writes D[t&1].  In-flight iterations are 3 planes apart
// (iteration t relaxes planes 2n-3t and 2n-3t+1 in launch n; its plus-neighbours on plane 2n-3t+2 were
// finished by iteration t-1 in launch n-1): (S + 3(T-1)) / 2 launches per level instead of S + 2(T-1).
#include <algorithm>

#include "fr3d_internal.h"
#include "k_sor_core.h"

namespace fr3d {

template <typename R, typename S, int C, typename I, int ROWS>
__global__ void __launch_bounds__(64 * (ROWS + 2))
k_sor_pair(const SorArgsT<S> a, int n, int t_lo, int nt, const SorEntry *__restrict__ ent,
           const int *__restrict__ lut)
{
    __shared__ S xch[3][ROWS + 1][66];
    const int Z = a.sk.Z, Y = a.sk.Y, X = a.sk.X, Yp = a.sk.Yp;
    const long long plane = a.sk.plane;
    const int vol = blockIdx.y;
    const int b = blockIdx.x;
    int lo = lut[b >> SOR_LUT_SHIFT];
    while (lo + 1 < nt && ent[lo + 1].pre <= b) lo++;
    const SorEntry en = ent[lo];
    const int local = b - en.pre;
    const int t = t_lo + lo;
    const int s = 2 * n - 3 * t;                       // phase-1 plane; phase 2 relaxes s + 1
    const int k0 = (en.kb0 + local / en.njb) * ROWS;   // first row the tile owns
    const int L0 = (local % en.njb) * 64;              // first lane the tile owns
    const int w = threadIdx.y, lane = threadIdx.x;

    const I esz = (I)sizeof(S);
    const I pl = (I)plane * esz, row = (I)Yp * esz;
    const bool odd = t & 1;
    S *const nU = (odd ? a.dB[0] : a.d[0]) + vol * a.vsD;  // written by iteration t (and read: plane s-1)
    S *const nV = (odd ? a.dB[1] : a.d[1]) + vol * a.vsD;
    S *const nW = (odd ? a.dB[2] : a.d[2]) + vol * a.vsD;
    const S *const oU = (odd ? a.d[0] : a.dB[0]) + vol * a.vsD;  // iteration t-1
    const S *const oV = (odd ? a.d[1] : a.dB[1]) + vol * a.vsD;
    const S *const oW = (odd ? a.d[2] : a.dB[2]) + vol * a.vsD;
    const bool upd = ((a.t_base + t) % a.update_lag) == 0;
    const long long vM = vol * a.vsM, vA = vol * a.vsA, vL = vol * a.vsL;

    // ---- phase 2's operands first: own old value, plus-neighbours on plane s+2 and (ordinary iterations)
    // the frozen system do not depend on phase 1, so their loads go out together with phase 1's and the
    // workgroup pays one memory round trip instead of two
    const bool p2wave = w >= 1 && w <= ROWS;
    bool ex2 = false;
    I c2 = 0;
    int i2 = 0, j2 = 0, dl2 = 0;
    const int k2 = k0 - 1 + w;
    S q_d0[3] = {0, 0, 0}, q_xp[3] = {0, 0, 0}, q_yp[3] = {0, 0, 0}, q_zp[3] = {0, 0, 0}, q_m[9];  // storage type: half the registers in the fp64-arithmetic mode
    if (p2wave) {
        const int jj = L0 + lane, r = s + 1 - k2;
        const int jm0 = sk_jm(X, r);
        j2 = jj + jm0;
        i2 = r - j2;
        ex2 = k2 < Z && r >= 0 && j2 < Y && i2 >= 0;
        if (ex2) {
            c2 = (I)((long long)(s + 1) * plane + (long long)k2 * Yp + jj) * esz;
            const I d2 = (I)(jm0 - sk_jm(X, r + 1)) * esz;
            dl2 = jm0 - sk_jm(X, r - 1);
            const I xp = (i2 < X - 1) ? c2 + pl + d2 : c2;
            const I yp = (j2 < Y - 1) ? c2 + pl + d2 + esz : c2;
            const I zp = (k2 < Z - 1) ? c2 + pl + row : c2;
            q_d0[0] = ldb(oU, c2); q_d0[1] = ldb(oV, c2); q_d0[2] = ldb(oW, c2);
            q_xp[0] = ldb(oU, xp); q_xp[1] = ldb(oV, xp); q_xp[2] = ldb(oW, xp);
            q_yp[0] = ldb(oU, yp); q_yp[1] = ldb(oV, yp); q_yp[2] = ldb(oW, yp);
            q_zp[0] = ldb(oU, zp); q_zp[1] = ldb(oV, zp); q_zp[2] = ldb(oW, zp);
            if (!upd) {
#pragma unroll
                for (int q = 0; q < 9; q++) q_m[q] = ldb(a.M[q] + vM, c2);
            }
        }
    }

    // ---- phase 1: plane s, the tile's rows plus the halo row (wave 0) and the halo lanes (last wave)
    {
        int k, jj;
        bool active = true;
        const bool core = w >= 1 && w <= ROWS;
        if (w <= ROWS) {
            k = k0 - 1 + w;
            jj = L0 + lane;
        } else {
            k = k0 + lane;
            const int rr = s - k;
            jj = (sk_jm(X, rr + 1) - sk_jm(X, rr)) ? L0 + 64 : L0 - 1;
            active = lane < ROWS;
        }
        const int r = s - k;
        const int jm0 = sk_jm(X, r);
        const int j = jj + jm0, i = r - j;
        if (active && k >= 0 && k < Z && r >= 0 && jj >= 0 && j < Y && i >= 0) {
            const I c0 = (I)((long long)s * plane + (long long)k * Yp + jj) * esz;
            const I d1 = (I)(jm0 - sk_jm(X, r - 1)) * esz, d2 = (I)(jm0 - sk_jm(X, r + 1)) * esz;
            const bool hxm = i > 0, hym = j > 0, hzm = k > 0;
            // every load is unconditional: a missing plus-neighbour re-reads the voxel's own old value,
            // a missing minus-neighbour reads a valid address of the new buffer and is replaced below
            const I xm = hxm ? c0 - pl + d1 : c0, ym = hym ? c0 - pl + d1 - esz : c0, zm = hzm ? c0 - pl - row : c0;
            const I xp = (i < X - 1) ? c0 + pl + d2 : c0;
            const I yp = (j < Y - 1) ? c0 + pl + d2 + esz : c0;
            const I zp = (k < Z - 1) ? c0 + pl + row : c0;
            const R du0 = (R)ldb(oU, c0), dv0 = (R)ldb(oV, c0), dw0 = (R)ldb(oW, c0);
            const R uxm = (R)ldb(nU, xm), vxm = (R)ldb(nV, xm), wxm = (R)ldb(nW, xm);
            const R uym = (R)ldb(nU, ym), vym = (R)ldb(nV, ym), wym = (R)ldb(nW, ym);
            const R uzm = (R)ldb(nU, zm), vzm = (R)ldb(nV, zm), wzm = (R)ldb(nW, zm);
            const R uxp = (R)ldb(oU, xp), vxp = (R)ldb(oV, xp), wxp = (R)ldb(oW, xp);
            const R uyp = (R)ldb(oU, yp), vyp = (R)ldb(oV, yp), wyp = (R)ldb(oW, yp);
            const R uzp = (R)ldb(oU, zp), vzp = (R)ldb(oV, zp), wzp = (R)ldb(oW, zp);
            R m[9];
            sor_system<R, S, C, I>(a, upd, core, vM, vA, vL, c0, du0, dv0, dw0, m);
            const R su_x = (hxm ? uxm : du0) + uxp, sv_x = (hxm ? vxm : dv0) + vxp, sw_x = (hxm ? wxm : dw0) + wxp;
            const R su_y = (hym ? uym : du0) + uyp, sv_y = (hym ? vym : dv0) + vyp, sw_y = (hym ? wym : dw0) + wyp;
            const R su_z = (hzm ? uzm : du0) + uzp, sv_z = (hzm ? vzm : dv0) + vzp, sw_z = (hzm ? wzm : dw0) + wzp;
            R du1, dv1, dw1;
            sor_relax<R>(m, a.ax, a.ay, a.az, su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z, du0, dv0, dw0, du1,
                         dv1, dw1);
            const int rs = k - k0 + 1, col = jj - L0 + 1;
            xch[0][rs][col] = (S)du1;
            xch[1][rs][col] = (S)dv1;
            xch[2][rs][col] = (S)dw1;
            if (core) {
                stb(nU, c0, (S)du1);
                stb(nV, c0, (S)dv1);
                stb(nW, c0, (S)dw1);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: plane s + 1, the tile's own rows and lanes
    if (ex2) {
        const R du0 = (R)q_d0[0], dv0 = (R)q_d0[1], dw0 = (R)q_d0[2];
        R m[9];
        if (upd) {
            sor_system<R, S, C, I>(a, true, true, vM, vA, vL, c2, du0, dv0, dw0, m);
        } else {
#pragma unroll
            for (int q = 0; q < 9; q++) m[q] = (R)q_m[q];
        }
        const int rs = w, col = lane + 1;  // = k - k0 + 1, jj - L0 + 1
        const bool hxm = i2 > 0, hym = j2 > 0, hzm = k2 > 0;
        const R uxm = hxm ? (R)xch[0][rs][col + dl2] : du0, vxm = hxm ? (R)xch[1][rs][col + dl2] : dv0,
                wxm = hxm ? (R)xch[2][rs][col + dl2] : dw0;
        const R uym = hym ? (R)xch[0][rs][col + dl2 - 1] : du0, vym = hym ? (R)xch[1][rs][col + dl2 - 1] : dv0,
                wym = hym ? (R)xch[2][rs][col + dl2 - 1] : dw0;
        const R uzm = hzm ? (R)xch[0][rs - 1][col] : du0, vzm = hzm ? (R)xch[1][rs - 1][col] : dv0,
                wzm = hzm ? (R)xch[2][rs - 1][col] : dw0;
        R du1, dv1, dw1;
        sor_relax<R>(m, a.ax, a.ay, a.az, uxm + (R)q_xp[0], vxm + (R)q_xp[1], wxm + (R)q_xp[2], uym + (R)q_yp[0],
                     vym + (R)q_yp[1], wym + (R)q_yp[2], uzm + (R)q_zp[0], vzm + (R)q_zp[1], wzm + (R)q_zp[2], du0, dv0,
                     dw0, du1, dv1, dw1);
        stb(nU, c2, (S)du1);
        stb(nV, c2, (S)dv1);
        stb(nW, c2, (S)dw1);
    }
}

// Launch schedule of the pair sweep: launch n relaxes, for every in-flight iteration t, the planes
// s = 2n - 3t and s + 1.  Per (launch, iteration) the bounding box of the tiles that hold voxels of
// either plane, tiles = ROWS rows x 64 lanes; same entry / group-table format as build_sor_schedule.
SorSched build_sor_pair_schedule(const Skew &sk, int T, int rows)
{
    SorSched sc;
    sc.by = rows;
    sc.lag = 3;
    const int S = sk.S, Z = sk.Z, Y = sk.Y, X = sk.X;
    std::vector<SorEntry> ent;
    std::vector<int> lut;
    if (T <= 0) return sc;
    const int last = (S - 1 + 3 * (T - 1)) / 2;
    for (int n = 0; n <= last; n++) {
        // s = 2n - 3t must lie in [-1, S-1]
        int t_lo = 2 * n - (S - 1);
        t_lo = t_lo <= 0 ? 0 : (t_lo + 2) / 3;
        int t_hi = (2 * n + 1) / 3;
        if (t_hi > T - 1) t_hi = T - 1;
        if (t_lo > t_hi) continue;
        sc.tau.push_back(n);
        sc.t_lo.push_back(t_lo);
        sc.nt.push_back(t_hi - t_lo + 1);
        sc.first.push_back((int)ent.size());
        int pre = 0;
        for (int t = t_lo; t <= t_hi; t++) {
            const int s = 2 * n - 3 * t;
            int klo_u = Z, khi_u = -1, maxlen = 0;
            for (int p = s; p <= s + 1; p++) {
                if (p < 0 || p > S - 1) continue;
                const int klo = std::max(0, p - (X - 1) - (Y - 1)), khi = std::min(Z - 1, p);
                if (klo > khi) continue;
                klo_u = std::min(klo_u, klo);
                khi_u = std::max(khi_u, khi);
                for (int k = klo; k <= khi; k++) {
                    const int r = p - k;
                    maxlen = std::max(maxlen, std::min(Y - 1, r) - sk_jm(X, r) + 1);
                }
            }
            SorEntry e;
            e.pre = pre;
            e.pad0 = e.pad1 = 0;
            e.kb0 = 0;
            e.njb = 1;
            if (klo_u <= khi_u && maxlen > 0) {
                const int kb0 = klo_u / rows, kb1 = khi_u / rows;
                e.kb0 = (short)kb0;
                e.njb = (short)cdiv(maxlen, 64);
                pre += (kb1 - kb0 + 1) * e.njb;
            }
            ent.push_back(e);
        }
        sc.ntiles.push_back(pre);
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
    FR3D_HIP(hipMalloc((void **)&sc.entries, std::max<size_t>(ent.size(), 1) * sizeof(SorEntry)));
    FR3D_HIP(hipMemcpy(sc.entries, ent.data(), ent.size() * sizeof(SorEntry), hipMemcpyHostToDevice));
    FR3D_HIP(hipMalloc((void **)&sc.lut, std::max<size_t>(lut.size(), 1) * sizeof(int)));
    FR3D_HIP(hipMemcpy(sc.lut, lut.data(), lut.size() * sizeof(int), hipMemcpyHostToDevice));
    return sc;
}

template <typename R, typename S, int ROWS>
static void launch_pair_step(hipStream_t st, const SorArgsT<S> &a, int n, int t_lo, int nt, int ntiles,
                             const SorEntry *ent, const int *lut)
{
    dim3 grid(ntiles, a.nvol > 0 ? a.nvol : 1), block(64, ROWS + 2);
    const bool narrow = (unsigned long long)a.sk.total * sizeof(S) < (1ull << 32);
#define FR3D_PAIR_CASE(CH)                                                                                          \
    case CH:                                                                                                        \
        if (narrow) hipLaunchKernelGGL((k_sor_pair<R, S, CH, unsigned, ROWS>), grid, block, 0, st, a, n, t_lo, nt, ent, lut); \
        else hipLaunchKernelGGL((k_sor_pair<R, S, CH, size_t, ROWS>), grid, block, 0, st, a, n, t_lo, nt, ent, lut);  \
        break;
    switch (a.C) {
        FR3D_PAIR_CASE(1)
        FR3D_PAIR_CASE(2)
        FR3D_PAIR_CASE(3)
        FR3D_PAIR_CASE(4)
        default: throw Error("SOR kernel is instantiated for 1..4 channels");
    }
#undef FR3D_PAIR_CASE
    FR3D_LAUNCH_CHECK();
}

// Runs all iterations; the increments end in a.d (iterations even or zero: buffer of iteration T-1 ...)
// -- see sor_pair_result().  Returns the number of launches.
template <typename S>
long long launch_sor_pair(hipStream_t st, const SorArgsT<S> &a_in, bool fp64, const SorSched &sc)
{
    SorArgsT<S> a = a_in;
    a.dbg = 0;
    FR3D_CHECK(sc.lag == 3 && (sc.by == 6 || sc.by == 14), "internal: not a pair schedule");
    FR3D_CHECK(a.dB[0] && a.dB[1] && a.dB[2], "internal: pair sweep needs the second increment buffer");
    long long launches = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        if (sc.ntiles[l] <= 0) continue;
        const SorEntry *ent = sc.entries + sc.first[l];
        const int *lut = sc.lut + sc.lut_first[l];
        const bool wide = fp64 || sizeof(S) == 8;
        if (sc.by == 6) {
            if (wide) launch_pair_step<double, S, 6>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, lut);
            else launch_pair_step<float, S, 6>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, lut);
        } else {
            if (wide) launch_pair_step<double, S, 14>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, lut);
            else launch_pair_step<float, S, 14>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, lut);
        }
        launches++;
    }
    return launches;
}
template long long launch_sor_pair<float>(hipStream_t, const SorArgsT<float> &, bool, const SorSched &);
template long long launch_sor_pair<double>(hipStream_t, const SorArgsT<double> &, bool, const SorSched &);

}  // namespace fr3d
