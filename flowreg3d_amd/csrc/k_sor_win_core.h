// k_sor_win_core.h -- the temporally blocked ("window") form of the a_smooth == 1 SOR sweep: per-thread state and the
// step function, shared by the gfx950 kernel (k_sor_win.hip) and its CPU emulator (tools/emu/sor_win_emu.hip).
// Reference: core/level_solver_3d.py:383-540 (lexicographic sweep, omega = 1.95), :356-377 (psi_data every
// update_lag iterations), :246-259 (Neumann ghosts).
//
// Why.  k_sor_step (k_sor.hip) streams the frozen system (9 values) and the increments (3 read, 3 written, plus the
// neighbour planes) of every voxel through HBM on EVERY iteration: 100-140 B per voxel update at 5.7 TB/s, and nothing
// tried in three rounds moved it.  This kernel keeps a whole psi window -- W = update_lag
// iterations -- on chip: HBM sees the factors, the weights, the Laplacian terms and the increments once per WINDOW
// (about 25-40 B per voxel update with fp64 storage), and the sweep becomes bound by its arithmetic and LDS traffic.
//
// How.  The dependences of the lexicographic sweep: voxel (k,j,i) of iteration t needs iteration t of (k,j,i-1),
// (k,j-1,i), (k-1,j,i) and iteration t-1 of itself and of (k,j,i+1), (k,j+1,i), (k+1,j,i).  A workgroup owns a tile of
// BK x BJ LINES (k,j) and marches along x.  In step s, "slot" q (iteration t0 + q of the window) of line (k,j)
// updates voxel i = s - 2q - k - j: all six neighbour values were produced in step s-1 (slot q: the -1 neighbours,
// slot q-1: the +1 neighbours) and the voxel's own old value in step s-2 (slot q-1).
// One THREAD = one (slot, line) pair (W x BK x BJ threads, slot-major so that a wave is one slot): a thread keeps its
// own last output (the i-1 neighbour) and last step's i+1 neighbour (= this step's own old value) in registers and
// reads everything else from LDS, where every thread publishes its output each step (double-buffered by step parity,
// one barrier per step); the frozen system of a voxel is built by the slot-0 thread and handed from slot to slot
// through LDS (it is needed again 2, 4, .. steps later: read one step ahead, held one step).  Per-thread state is
// ~30 values, so three to four waves share a SIMD and HBM loads are simply used in the step that issues them.
// Time skewing makes the tiles independent: slot q of tile (K,J) covers the lines [K BK - q, (K+1) BK - q) x
// [J BJ - q, (J+1) BJ - q), so every +1 neighbour of iteration t-1 lies in the SAME tile's slot q-1, and the only
// values a tile needs from others come from tiles with smaller K or J (complete before it starts: tiles are
// launched by diagonals K + J, consecutive windows three diagonals apart).  The thread <-> line mapping is cyclic:
// thread (q,a,b) handles line (K BK + a, J BJ + b) while it is inside the slot's range and the line BK (BJ) below once
// the range has moved past it.
// What crosses tiles (through HBM, written by the lower tile in an earlier launch):
//   * E[q]: the increments of slot q < last on the top row / column of the slot's range ("exports"); read as the -1
//     neighbour by the tile above (NB), and by the thread of the next slot that takes the line over (SW: the line's
//     i+1 neighbour, which a step later is its own old value; SX: the +1 neighbour in the other direction, which
//     sits on the same row);
//   * M: the frozen system of lines that will be taken over (built by slot 0 of the tile the line starts in);
//   * d: the final slot writes the increments in place; the next window reads them three diagonals later.
// Results are bit-identical to k_sor_step in every storage format (same per-voxel functions, same rounding to the
// storage format after every update): tests/test_gpu_sor_window.py.
#pragma once

#include <cstring>

#include "k_sor_core.h"

namespace fr3d {

#define WIN_BK 8     // lines of a tile in k
#define WIN_BJ 16    // lines of a tile in j
#define WIN_NL (WIN_BK * WIN_BJ)    // lines of a tile = threads per slot (a multiple of 64: a wave is one slot)
#define WIN_WMAX 5   // slots (iterations) of a window kept on chip
#define WIN_NT (WIN_WMAX * WIN_NL)  // threads of a workgroup
#define WIN_DLAG 3   // tile diagonals between consecutive windows

// one workgroup: tile (K,J) of window [t0, t0 + nslots)
struct alignas(16) WinTile {
    int K, J;
    int t0;
    int info;  // nslots | build << 8 (slot 0 is a psi update: build M,b from the factors) | storeM << 9 (write M for every voxel)
};

template <typename S>
struct WinArgs {
    SorArgsT<S> a;           // M (records of 9), A, weight, L, d (in place), geometry, constants, batch strides
    S *E;                    // exports of slots 0 .. W-2, records of 3: array q starts at E + q * strideE, laid out and
    long long strideE;       // strided (per volume) like d
};

// LDS image of one workgroup (double-buffered by step parity; one barrier per step):
//  * O[p][q][c][line]: outputs of the slots (index q + 1) and of the loader (index 0: d of the slot-0 line, two voxels
//    ahead of slot 0) in the steps of parity p;
//  * H[p][q][w][line]: the frozen system handed to slot q + 1 by slot q, raw storage words.
// Slot 0's HBM operands are NOT staged here: requesting them a step ahead by the gfx950 async global -> LDS copy
// (global_load_lds; lane-linear destinations, 16- and 4-byte chunks -- the 12-byte form faulted on this pool, gpurun r04r)
// or by register loads parked with ds_write both lost against plain same-step loads (numbers at the loads in step_t).
template <typename S, int C, int W, bool BUILD>
struct WinLds {
    using V = typename Sto<S>::val;
    using WT = typename StoWt<S>::type;
    static constexpr int NWH = sizeof(RawRec<S, 9>) / 4;
    V O[2][W + 1][3][WIN_NL];
    unsigned H[2][W > 1 ? W - 1 : 1][NWH][WIN_NL];
};

// Row-start tables of the compact skewed layout (Skew::pb / cp) as the kernel sees them: the device keeps copies in
// LDS (a dependent GLOBAL load in front of every prefetch address made each step a chain of memory round trips), the
// emulator reads the host vectors.
struct WinTabPtr {
    const long long *pb;
    const int *cp;
    FR3D_HD long long pbv(int n) const { return pb[n]; }
    FR3D_HD int cpv(int n) const { return cp[n]; }
};
template <typename Tab>
FR3D_HD long long win_index(const Tab &tb, int X, int k, int j, int i)
{
    const int r = i + j, s = r + k;
    return tb.pbv(s + 1) - (long long)tb.cpv(r + 1) + (j - sk_jm(X, r));
}

// Memory hook: the device build reads and writes plainly; the emulator's hook checks that a value read was written
// by the expected window (and not by a workgroup of the same launch).
struct WinNoHook {
    FR3D_HD void rd(int, long long, int) const {}
    FR3D_HD void wr(int, long long, int) const {}
};
enum { WIN_ARR_D = 0, WIN_ARR_M = 1, WIN_ARR_E0 = 2 };  // array ids of the hook (E[q] = WIN_ARR_E0 + q)

#if defined(__HIP_DEVICE_COMPILE__)
#define WIN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WIN_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)  // wave-uniform: a branch, not a per-lane select
#else
#define WIN_SCHED_FENCE() ((void)0)
#define WIN_ANY(c) (c)
#endif

template <typename R, typename S, int C, int W, bool BUILD, typename Tab = WinTabPtr, typename Hook = WinNoHook>
struct WinThread {
    using V = typename Sto<S>::val;
    using WT = typename StoWt<S>::type;
    static_assert(W >= 1 && W <= WIN_WMAX, "window slots");
    static_assert(C >= 1, "channel count is a template parameter");
    static_assert(WIN_NL % 64 == 0 && W - 1 < WIN_BK && W - 1 < WIN_BJ, "tile shape");
    using Lds = WinLds<S, C, W, BUILD>;

    // ---- constant per thread ----
    int q, a, b, line, tid, lane0_line;
    int K, J, nsl;
    bool storeM;
    bool spa, spb;  // this thread's line is the bottom of its slot's range in a / in b: imports instead of neighbours
    int k, j;       // its line
    int Z, Y, X;
    long long vD, vM, vA, vL;  // volume offsets (storage elements)
    int win, win_build;        // window of this workgroup / of the psi update its system belongs to (hook versions)

    // ---- state ----
    V xp_prev[3];    // last step's i+1 neighbour = this step's own old value
    V out_prev[3];   // last step's output = this step's i-1 neighbour
    RawRec<S, 9> Hcur;  // slots >= 1: the system of this step's voxel (read from LDS at the end of the previous step)

    FR3D_HD static int kline(int K, int a, int q) { return K * WIN_BK + a - (a >= WIN_BK - q ? WIN_BK : 0); }
    FR3D_HD static int jline(int J, int b, int q) { return J * WIN_BJ + b - (b >= WIN_BJ - q ? WIN_BJ : 0); }
    FR3D_HD bool line_ok(int kk, int jj) const { return kk >= 0 && kk < Z && jj >= 0 && jj < Y; }
    FR3D_HD static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
    FR3D_HD static V sel(bool c, V x, V y) { return c ? x : y; }

    template <typename T>
    FR3D_HD void words_put(unsigned (*dst)[WIN_NL], const T &r) const
    {
        struct Words { unsigned w[sizeof(T) / 4]; };
        const Words x = __builtin_bit_cast(Words, r);
#pragma unroll
        for (int n = 0; n < (int)(sizeof(T) / 4); n++) dst[n][line] = x.w[n];
    }
    template <typename T>
    FR3D_HD T words_get(const unsigned (*src)[WIN_NL]) const
    {
        struct Words { unsigned w[sizeof(T) / 4]; };
        Words x;
#pragma unroll
        for (int n = 0; n < (int)(sizeof(T) / 4); n++) x.w[n] = src[n][line];
        return __builtin_bit_cast(T, x);
    }
    FR3D_HD static RawRec<S, 9> encode9(const Rec<S, 9> &r)
    {
        RawRec<S, 9> raw;
        if constexpr (std::is_same<S, pk42>::value) {
            alignas(16) S tmp[12];
            strec<S, 9>(tmp, 0, r);
            raw = *reinterpret_cast<const RawRec<S, 9> *>(tmp);
        } else {
#pragma unroll
            for (int n = 0; n < 9; n++) raw.w[n] = r.v[n];
        }
        return raw;
    }

    FR3D_HD void init(const WinArgs<S> &wa, const WinTile &tl, int vol, int tid_, int win_, int win_build_)
    {
        const SorArgsT<S> &A_ = wa.a;
        tid = tid_;
        q = tid / WIN_NL;  // slot-major: a wave is one slot
        line = tid % WIN_NL;
        lane0_line = line - (tid & 63);
        a = line / WIN_BJ;
        b = line % WIN_BJ;
        K = tl.K; J = tl.J;
        nsl = tl.info & 0xff;
        storeM = (tl.info >> 9) & 1;
        win = win_; win_build = win_build_;
        spa = q == (WIN_BK - a) % WIN_BK;
        spb = q == (WIN_BJ - b) % WIN_BJ;
        k = kline(K, a, q);
        j = jline(J, b, q);
        Z = A_.sk.Z; Y = A_.sk.Y; X = A_.sk.X;
        vD = vol * A_.vsD; vM = vol * A_.vsM; vA = vol * A_.vsA; vL = vol * A_.vsL;
#pragma unroll
        for (int c = 0; c < 3; c++) xp_prev[c] = out_prev[c] = 0;
        Hcur = RawRec<S, 9>{};
    }

    // first step in which any slot of this tile has a voxel, and the last one; the kernel starts WIN_LEAD steps
    // earlier, with every slot still idle, so that the pipelines (requests, loader, own-old values, systems) fill
    // through the ordinary step code
    FR3D_HD static void step_range(const Skew &sk, const WinTile &tl, int &s_first, int &s_last)
    {
        const int nsl = tl.info & 0xff;
        const int klo = tl.K * WIN_BK - (nsl - 1) > 0 ? tl.K * WIN_BK - (nsl - 1) : 0;
        const int jlo = tl.J * WIN_BJ - (nsl - 1) > 0 ? tl.J * WIN_BJ - (nsl - 1) : 0;
        const int khi = (tl.K + 1) * WIN_BK - 1 < sk.Z - 1 ? (tl.K + 1) * WIN_BK - 1 : sk.Z - 1;
        const int jhi = (tl.J + 1) * WIN_BJ - 1 < sk.Y - 1 ? (tl.J + 1) * WIN_BJ - 1 : sk.Y - 1;
        s_first = klo + jlo;
        s_last = khi + jhi + (sk.X - 1) + 2 * (nsl - 1);
    }

    // an increment record of this window from another tile: slot qq's export (or, for the window's last slot, d)
    FR3D_HD Rec<S, 3> import3(const WinArgs<S> &wa, const Tab &tb, int qq, int kk, int jj, int ii, const Hook &hk) const
    {
        const bool last = qq == nsl - 1;
        const long long e = win_index(tb, X, kk, jj, ii);
        hk.rd(last ? WIN_ARR_D : WIN_ARR_E0 + qq, e, win);
        const S *src = last ? wa.a.d + vD : wa.E + (long long)qq * wa.strideE + vD;
        return ldrec<S, 3>(src, e);
    }

    // step s of this thread's slot: read the LDS image of step s-1 (parity pp), compute, publish in the image of step s.
    // SLOT0 = this thread is a slot-0 thread (a separate code path: the two roles need different registers).
    // The caller puts a barrier behind it.
    template <bool SLOT0>
    FR3D_HD void step_t(const WinArgs<S> &wa, const Tab &tb, int s, Lds &lds, const Hook &hk)
    {
        const SorArgsT<S> &A_ = wa.a;
        const int pp = (s - 1) & 1, pc = s & 1;
        const int i = s - 2 * q - k - j;
        const bool lok = q < nsl && line_ok(k, j);
        const bool active = lok && i >= 0 && i < X;
        const int t_am = ((a + WIN_BK - 1) % WIN_BK) * WIN_BJ + b, t_ap = ((a + 1) % WIN_BK) * WIN_BJ + b;
        const int t_bm = a * WIN_BJ + (b + WIN_BJ - 1) % WIN_BJ, t_bp = a * WIN_BJ + (b + 1) % WIN_BJ;

        V own[3], xp[3], xm[3], yp[3], ym[3], zp[3], zm[3];
#pragma unroll
        for (int c = 0; c < 3; c++) own[c] = xp_prev[c];

        // ---- the frozen system of this step's voxel ----
        Rec<S, 9> mr;
        Rec<S, 3> Lrec;
        if constexpr (SLOT0) {
            // Same-step loads (clamped, unconditional).  Two ways of requesting them one step early were measured and
            // lost at 256^3 fp64 storage (107 ms of sweeps per volume with these loads): LDS-DMA into a staging
            // buffer (14 wave-instructions of 60-185 issue cycles each: 135 ms) and register loads parked in LDS at
            // the end of the step (38 more live registers in this role, 132 B of scratch: 208 ms).
            const int kc = clampi(k, 0, Z - 1), jc = clampi(j, 0, Y - 1);
            {   // the loader: d of this line two voxels ahead of slot 0
                const int il = i + 2;
                const long long e = win_index(tb, X, kc, jc, clampi(il, 0, X - 1));
                if (lok && il >= 0 && il < X) hk.rd(WIN_ARR_D, e, win - 1);
                Lrec = ldrec<S, 3>(A_.d + vD, e);
            }
            const long long e = win_index(tb, X, kc, jc, clampi(i, 0, X - 1));
            if constexpr (BUILD) {
                SorAcc<R> acc;
#pragma unroll
                for (int c = 0; c < C; c++)
                    sor_accum_channel<R, S>(ldrec<S, 12>(A_.A[c] + vA, e), (double)A_.weight[c][e], A_.a_data[c], (R)own[0],
                                            (R)own[1], (R)own[2], acc);
                mr = sor_finish_system<R, S>(acc, ldrec<S, 3>(A_.L + vL, e));
                // lines that another tile takes over later need the record in memory
                if (active && (storeM || a >= WIN_BK - W + 1 || b >= WIN_BJ - W + 1)) {
                    hk.wr(WIN_ARR_M, e, win);
                    strec<S, 9>(A_.M + vM, e, mr);
                }
            } else {
                if (active) hk.rd(WIN_ARR_M, e, win_build);
                mr = ldrec<S, 9>(A_.M + vM, e);
            }
        }

        // ---- the LDS image of step s-1 ----
#pragma unroll
        for (int c = 0; c < 3; c++) {
            xm[c] = out_prev[c];
            xp[c] = lds.O[pp][q][c][line];          // this line in slot q-1 (q = 0: the loader), one voxel ahead
            zm[c] = lds.O[pp][q + 1][c][t_am];       // -1 neighbours: this slot
            ym[c] = lds.O[pp][q + 1][c][t_bm];
            zp[c] = lds.O[pp][q][c][t_ap];           // +1 neighbours: slot q-1
            yp[c] = lds.O[pp][q][c][t_bp];
        }

        // ---- what comes from other tiles instead (straight into the neighbour values: no registers of their own) ----
        auto take = [&](V (&dst)[3], const Rec<S, 3> &r) {
#pragma unroll
            for (int c = 0; c < 3; c++) dst[c] = r.v[c];
        };
        if constexpr (SLOT0) {
            if (active) {
                // top row / column of slot 0: the +1 neighbour belongs to the tile above; its value of the previous window is in d
                if (a == WIN_BK - 1 && k + 1 < Z) {
                    const long long en = win_index(tb, X, k + 1, j, i);
                    hk.rd(WIN_ARR_D, en, win - 1);
                    take(zp, ldrec<S, 3>(A_.d + vD, en));
                }
                if (b == WIN_BJ - 1 && j + 1 < Y) {
                    const long long en = win_index(tb, X, k, j + 1, i);
                    hk.rd(WIN_ARR_D, en, win - 1);
                    take(yp, ldrec<S, 3>(A_.d + vD, en));
                }
            }
        }
        if (lok && (spa || spb)) {
            // -1 neighbours across the tile edge: this window's slot-q value of the line below
            if (active && spa && k > 0) take(zm, import3(wa, tb, q, k - 1, j, i, hk));
            if (active && spb && j > 0) take(ym, import3(wa, tb, q, k, j - 1, i, hk));
            if constexpr (!SLOT0) {
                // the line's slot q-1 values come from the exports of the tile it is taken over from: its i+1 neighbour
                // (one step later its own old value), its frozen system, and the +1 neighbour in the other direction
                // (on the same exported row / column)
                if (i + 1 >= 0 && i + 1 < X) take(xp, import3(wa, tb, q - 1, k, j, i + 1, hk));
                if (active) {
                    const long long e = win_index(tb, X, k, j, i);
                    hk.rd(WIN_ARR_M, e, win_build);
                    Hcur = ldraw<S, 9>(A_.M + vM, e);
                    if (spa && j + 1 < Y) take(yp, import3(wa, tb, q - 1, k, j + 1, i, hk));
                    if (spb && k + 1 < Z) take(zp, import3(wa, tb, q - 1, k + 1, j, i, hk));
                }
            }
        }
        if constexpr (!SLOT0) mr = Hcur.dec();  // handed over by slot q-1, or just imported with the line

        // Neumann ghosts (set_boundary_3d :246-259): a missing neighbour is the voxel's own old value
        // (xp keeps the neighbour's value for the next step's own old value; the ghosted copy is xpg)
        V xpg[3];
#pragma unroll
        for (int c = 0; c < 3; c++) xpg[c] = xp[c];
        const bool ghost = i <= 0 || i >= X - 1 || j == 0 || j == Y - 1 || k == 0 || k == Z - 1;
        if (WIN_ANY(ghost)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                xpg[c] = sel(i < X - 1, xp[c], own[c]);
                xm[c] = sel(i > 0, xm[c], own[c]);
                zm[c] = sel(k > 0, zm[c], own[c]);
                ym[c] = sel(j > 0, ym[c], own[c]);
                zp[c] = sel(k < Z - 1, zp[c], own[c]);
                yp[c] = sel(j < Y - 1, yp[c], own[c]);
            }
        }
        R m[9];
#pragma unroll
        for (int n = 0; n < 9; n++) m[n] = (R)mr.v[n];
        R du1, dv1, dw1;
        sor_relax<R>(m, A_.ax, A_.ay, A_.az, (R)xm[0] + (R)xpg[0], (R)xm[1] + (R)xpg[1], (R)xm[2] + (R)xpg[2],
                     (R)ym[0] + (R)yp[0], (R)ym[1] + (R)yp[1], (R)ym[2] + (R)yp[2], (R)zm[0] + (R)zp[0],
                     (R)zm[1] + (R)zp[1], (R)zm[2] + (R)zp[2], (R)own[0], (R)own[1], (R)own[2], du1, dv1, dw1);
        Rec<S, 3> out;
        out.v[0] = Sto<S>::quant(du1);
        out.v[1] = Sto<S>::quant(dv1);
        out.v[2] = Sto<S>::quant(dw1);

        // ---- publish ----
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lds.O[pc][q + 1][c][line] = out.v[c];
            if constexpr (SLOT0) lds.O[pc][0][c][line] = Lrec.v[c];
            out_prev[c] = out.v[c];
            xp_prev[c] = xp[c];
        }
        // the record travels on: slot q+1 reads it at the end of the next step and uses it in the one after
        if (q + 1 < W) words_put(lds.H[pc][q], SLOT0 ? encode9(mr) : Hcur);
        // the system of the NEXT step's voxel, handed over by slot q-1 in step s-1 (the image of step s-1 is stable
        // until the barrier)
        if constexpr (!SLOT0) Hcur = words_get<RawRec<S, 9>>(lds.H[pp][q - 1]);
        if (active) {
            const bool fin = q == nsl - 1;
            const bool exp_ = q < W - 1 && (a == WIN_BK - 1 - q || b == WIN_BJ - 1 - q);
            if (fin || exp_) {
                const long long e = win_index(tb, X, k, j, i);
                hk.wr(fin ? WIN_ARR_D : WIN_ARR_E0 + q, e, win);
                S *dst = fin ? A_.d + vD : wa.E + (long long)q * wa.strideE + vD;
                strec<S, 3>(dst, e, out);
            }
        }
    }

    FR3D_HD void step(const WinArgs<S> &wa, const Tab &tb, int s, Lds &lds, const Hook &hk)
    {
        if (q == 0) step_t<true>(wa, tb, s, lds, hk);
        else step_t<false>(wa, tb, s, lds, hk);
    }
};

#define WIN_LEAD 4  // steps the kernel runs ahead of the tile's first voxel (2 needed: loader, own-old pipeline)

}  // namespace fr3d
