// k_sor_band.hip -- two hyperplanes per launch for the a_smooth == 1 SOR sweep
// (core/level_solver_3d.py:383-540), still lexicographic-exact and bit-identical to k_sor.hip.
//
// k_sor.hip advances every in-flight iteration by ONE hyperplane per launch; each update then pulls
// its own old value plus both neighbour planes from HBM (4 plane-equivalents of increments per
// update with 2-row tiles) and a level costs S + 2(T-1) dependent launches.  Here a launch advances
// every in-flight iteration by the plane PAIR (P, P+1), P = 2*launch - 3t:
//   * a workgroup owns a band of BY rows (k) over their full length and walks it in 64-lane chunks;
//     BY+1 waves relax plane P (own rows + the row below the band, recomputed and not stored) and
//     publish the new values in an LDS ring, BY more waves relax plane P+1 one chunk behind, taking
//     its three "already updated" neighbours (k,j,i-1), (k,j-1,i), (k-1,j,i) -- all on plane P --
//     from LDS.  HBM sees (4BY+6)/(2BY) = 2.4 plane-equivalents per update instead of 4.
//   * iteration t+1 runs 3 planes behind t (its P+1 plane needs t's values up to its plane + 1,
//     which t finished in the previous launch), so a level takes (S + 3(T-1))/2 launches.
//   * old and new values of a plane must coexist while a neighbouring band still needs the old
//     ones (the band below recomputes our first row's predecessor from OLD data), so the increments
//     are ping-ponged: iteration t reads t-1's results from D[(t-1)&1] and writes D[t&1].
#include <algorithm>
#include <cstdlib>

#include "fr3d_internal.h"
#include "k_sor_core.h"

namespace fr3d {

#define BAND_BX 64
#define BAND_RING 256  // LDS ring: 4 chunks of 64 (phase B of chunk c-1 touches chunks c-2..c)

// Wave roles: threadIdx.y in [0,BY] relaxes row k = k0-1+ty of plane P ("A" waves; ty == 0 is the row
// below the band, recomputed and not stored), threadIdx.y in (BY,2BY] relaxes row k0-1+(ty-BY) of
// plane P+1 ("B" waves) one chunk behind.  One barrier per chunk step: A(c) publishes its new
// values in the LDS ring, B(c-1) has meanwhile fetched everything it needs from HBM and finishes
// from the ring after the barrier while the A waves are already loading chunk c+1.
template <typename R, typename S, int C, int BY>
__global__ void __launch_bounds__(BAND_BX *(2 * BY + 1)) __attribute__((amdgpu_waves_per_eu(7, 8)))
k_sor_band(const SorArgsT<S> a, int twoL, int t_lo, int nt, const SorEntry *__restrict__ ent)
{
    __shared__ S ring[3][BY + 1][BAND_RING];
    const int Z = a.sk.Z, Y = a.sk.Y, X = a.sk.X, Yp = a.sk.Yp, NS = a.sk.S;
    const long long plane = a.sk.plane;
    const int vol = blockIdx.y;
    const int b = blockIdx.x;
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ent[mid].pre <= b) lo = mid;
        else hi = mid - 1;
    }
    const SorEntry en = ent[lo];
    const int t = t_lo + lo;
    const int P = twoL - 3 * t;  // planes P and P+1 of iteration t
    const int tx = threadIdx.x;
    const bool isB = (int)threadIdx.y > BY;
    const int ty = isB ? (int)threadIdx.y - BY : (int)threadIdx.y;  // row slot in the ring (B: 1..BY)
    const int k = (en.kb0 + (b - en.pre)) * BY - 1 + ty;
    const int nch = en.njb;
    const bool upd = (t % a.update_lag) == 0;
    const int par = t & 1;
    S *const nU = (par ? a.d2[0] : a.d[0]) + vol * a.vsD;  // written by iteration t
    S *const nV = (par ? a.d2[1] : a.d[1]) + vol * a.vsD;
    S *const nW = (par ? a.d2[2] : a.d[2]) + vol * a.vsD;
    const S *const oU = (par ? a.d[0] : a.d2[0]) + vol * a.vsD;  // results of iteration t-1
    const S *const oV = (par ? a.d[1] : a.d2[1]) + vol * a.vsD;
    const S *const oW = (par ? a.d[2] : a.d2[2]) + vol * a.vsD;

    // this wave's row: (s,k) with s = P (A) or P+1 (B); r = i + j, first valid j, valid voxels
    const int s = P + (isB ? 1 : 0);
    const bool kin = k >= 0 && k < Z;
    const int r = s - k;
    const int jm0 = sk_jm(X, r);
    const int len = (kin && s >= 0 && s < NS && r >= 0 && r <= X + Y - 2) ? min(Y - 1, r) - jm0 + 1 : 0;
    // element offsets inside one volume's array stay below 2^32 (checked by the launcher), so the
    // per-lane address part is 32 bits and every load is scalar base + vector offset
    const unsigned row = (unsigned)((long long)s * plane + (long long)k * Yp);
    const unsigned upl = (unsigned)plane;
    const int d1 = jm0 - sk_jm(X, r - 1), d2 = jm0 - sk_jm(X, r + 1);
    const long long vM = vol * a.vsM, vA = vol * a.vsA, vL = vol * a.vsL;

#pragma unroll 1
    for (int c = 0; c <= nch; c++) {
        // B works one chunk behind A
        const int jj = (isB ? c - 1 : c) * BAND_BX + tx;
        const bool act = jj >= 0 && jj < len && (isB ? c >= 1 && !(a.dbg & 2) : c < nch);
        const int j = jj + jm0, i = r - j;
        const unsigned c0 = row + (unsigned)jj;
        const bool hxm = i > 0, hxp = i < X - 1, hym = j > 0, hyp = j < Y - 1, hzm = k > 0, hzp = k < Z - 1;
        // A missing (ghost) neighbour holds the voxel's own old value (set_boundary_3d): its index
        // falls back to c0, so every load is unconditional and all of them are issued together.
        R du0 = 0, dv0 = 0, dw0 = 0, pxu = 0, pxv = 0, pxw = 0, pyu = 0, pyv = 0, pyw = 0, pzu = 0, pzv = 0, pzw = 0;
        R mxu = 0, mxv = 0, mxw = 0, myu = 0, myv = 0, myw = 0, mzu = 0, mzv = 0, mzw = 0;
        R m[9];
        if (act) {
            // everything that does not depend on this launch's plane-P results
            const unsigned px = hxp ? c0 + upl + (unsigned)d2 : c0, py = hyp ? c0 + upl + (unsigned)(d2 + 1) : c0,
                           pz = hzp ? c0 + upl + (unsigned)Yp : c0;
            du0 = (R)oU[c0]; dv0 = (R)oV[c0]; dw0 = (R)oW[c0];
            pxu = (R)oU[px]; pxv = (R)oV[px]; pxw = (R)oW[px];
            pyu = (R)oU[py]; pyv = (R)oV[py]; pyw = (R)oW[py];
            pzu = (R)oU[pz]; pzv = (R)oV[pz]; pzw = (R)oW[pz];
            if (!isB) {
                // A: the already-updated neighbours are on plane P-1 in HBM
                const unsigned mx = hxm ? c0 - upl + (unsigned)d1 : c0, my = hym ? c0 - upl + (unsigned)(d1 - 1) : c0,
                               mz = hzm ? c0 - upl - (unsigned)Yp : c0;
                mxu = (R)nU[mx]; mxv = (R)nV[mx]; mxw = (R)nW[mx];
                myu = (R)nU[my]; myv = (R)nV[my]; myw = (R)nW[my];
                mzu = (R)nU[mz]; mzv = (R)nV[mz]; mzw = (R)nW[mz];
            }
            sor_system<R, S, C, unsigned>(a, upd, isB || ty > 0, vM, vA, vL, c0, du0, dv0, dw0, m);
        }
        if (!isB && act) {
            // ---- A: plane P, chunk c ------------------------------------------------------------
            const R su_x = (hxm ? mxu : du0) + pxu, sv_x = (hxm ? mxv : dv0) + pxv, sw_x = (hxm ? mxw : dw0) + pxw;
            const R su_y = (hym ? myu : du0) + pyu, sv_y = (hym ? myv : dv0) + pyv, sw_y = (hym ? myw : dw0) + pyw;
            const R su_z = (hzm ? mzu : du0) + pzu, sv_z = (hzm ? mzv : dv0) + pzv, sw_z = (hzm ? mzw : dw0) + pzw;
            R du1, dv1, dw1;
            sor_relax<R>(m, a.ax, a.ay, a.az, su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z, du0, dv0, dw0,
                         du1, dv1, dw1);
            const S ru = (S)du1, rv = (S)dv1, rw = (S)dw1;
            const int q = jj & (BAND_RING - 1);
            ring[0][ty][q] = ru;
            ring[1][ty][q] = rv;
            ring[2][ty][q] = rw;
            if (ty > 0) {
                nU[c0] = ru;
                nV[c0] = rv;
                nW[c0] = rw;
            }
        }
        if (!(a.dbg & 1)) __syncthreads();
        if (isB && act) {
            // ---- B: plane P+1, chunk c-1; the already-updated neighbours are in the ring ---------
            const int qx = (jj + d1) & (BAND_RING - 1), qy = (jj + d1 - 1) & (BAND_RING - 1), qz = jj & (BAND_RING - 1);
            const R su_x = (hxm ? (R)ring[0][ty][qx] : du0) + pxu, sv_x = (hxm ? (R)ring[1][ty][qx] : dv0) + pxv,
                    sw_x = (hxm ? (R)ring[2][ty][qx] : dw0) + pxw;
            const R su_y = (hym ? (R)ring[0][ty][qy] : du0) + pyu, sv_y = (hym ? (R)ring[1][ty][qy] : dv0) + pyv,
                    sw_y = (hym ? (R)ring[2][ty][qy] : dw0) + pyw;
            const R su_z = (hzm ? (R)ring[0][ty - 1][qz] : du0) + pzu, sv_z = (hzm ? (R)ring[1][ty - 1][qz] : dv0) + pzv,
                    sw_z = (hzm ? (R)ring[2][ty - 1][qz] : dw0) + pzw;
            R du1, dv1, dw1;
            sor_relax<R>(m, a.ax, a.ay, a.az, su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z, du0, dv0, dw0,
                         du1, dv1, dw1);
            nU[c0] = (S)du1;
            nV[c0] = (S)dv1;
            nW[c0] = (S)dw1;
        }
    }
}

#ifndef BAND_BY
#define BAND_BY 6
#endif

static void launch_band(hipStream_t st, const SorArgsT<float> &a, int twoL, int t_lo, int nt, int nblocks,
                        const SorEntry *ent)
{
    dim3 grid(nblocks, a.nvol > 0 ? a.nvol : 1), block(BAND_BX, 2 * BAND_BY + 1);
    hipLaunchKernelGGL((k_sor_band<float, float, 1, BAND_BY>), grid, block, 0, st, a, twoL, t_lo, nt, ent);
}

// The band kernel is built for the default solver mode (fp32 storage and arithmetic), one channel,
// and per-volume arrays below 2^32 bytes (32-bit lane offsets); everything else runs on k_sor.hip.
bool sor_band_usable(const Skew &sk, int C, int solver_fp64, double a_smooth)
{
    return sor_kernel_choice() == 1 && C == 1 && solver_fp64 == 0 && a_smooth == 1.0 &&
           (unsigned long long)sk.total * sizeof(float) < (1ull << 32);
}

int sor_kernel_choice()
{
    const char *env = getenv("FR3D_SOR_KERNEL");  // read per call: tests switch kernels in-process
    return env ? atoi(env) : 0;
}

// Launch l handles, for every iteration t with a valid plane in the pair, planes (P, P+1),
// P = 2l - 3t.  SorSched::tau holds 2l; entries give, per iteration, the workgroup prefix, the
// first band and the chunk count of the longest row of the pair.
SorSched build_band_schedule(const Skew &sk, int T)
{
    SorSched sc;
    sc.by = BAND_BY;
    const int S = sk.S, Z = sk.Z, Y = sk.Y, X = sk.X;
    std::vector<SorEntry> ent;
    if (T <= 0) return sc;
    for (int l = 0;; l++) {
        // valid t: P + 1 >= 0 and P <= S - 1
        int t_hi = (2 * l + 1) / 3;
        if (t_hi > T - 1) t_hi = T - 1;
        int t_lo = 2 * l - (S - 1);
        t_lo = t_lo <= 0 ? 0 : (t_lo + 2) / 3;
        if (t_lo > T - 1) break;  // the last iteration has passed the last plane
        if (t_lo > t_hi) continue;
        sc.tau.push_back(2 * l);
        sc.t_lo.push_back(t_lo);
        sc.nt.push_back(t_hi - t_lo + 1);
        sc.first.push_back((int)ent.size());
        int pre = 0;
        for (int t = t_lo; t <= t_hi; t++) {
            const int P = 2 * l - 3 * t;
            int kmin = Z, kmax = -1, maxlen = 0;
            for (int s = P; s <= P + 1; s++) {
                if (s < 0 || s > S - 1) continue;
                const int klo = std::max(0, s - (X - 1) - (Y - 1)), khi = std::min(Z - 1, s);
                for (int k = klo; k <= khi; k++) {
                    const int r = s - k;
                    const int len = std::min(Y - 1, r) - sk_jm(X, r) + 1;
                    if (len <= 0) continue;
                    maxlen = std::max(maxlen, len);
                    kmin = std::min(kmin, k);
                    kmax = std::max(kmax, k);
                }
            }
            SorEntry e;
            e.pre = pre;
            e.pad0 = e.pad1 = 0;
            e.kb0 = 0;
            e.njb = 1;
            if (kmax >= kmin) {
                e.kb0 = (short)(kmin / BAND_BY);
                e.njb = (short)cdiv(maxlen, BAND_BX);
                pre += kmax / BAND_BY - kmin / BAND_BY + 1;
            }
            ent.push_back(e);
        }
        sc.ntiles.push_back(pre);
    }
    FR3D_HIP(hipMalloc((void **)&sc.entries, ent.size() * sizeof(SorEntry)));
    FR3D_HIP(hipMemcpy(sc.entries, ent.data(), ent.size() * sizeof(SorEntry), hipMemcpyHostToDevice));
    return sc;
}

long long launch_sor_band(hipStream_t st, const SorArgsT<float> &a_in, const SorSched &sc)
{
    FR3D_CHECK(a_in.C == 1 && a_in.d2[0] != nullptr, "internal: band kernel preconditions");
    SorArgsT<float> a = a_in;
    const char *dbg_env = getenv("FR3D_SOR_DBG");
    a.dbg = dbg_env ? atoi(dbg_env) : 0;
    long long launches = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        if (sc.ntiles[l] <= 0) continue;
        launch_band(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], sc.entries + sc.first[l]);
        launches++;
    }
    return launches;
}

}  // namespace fr3d
