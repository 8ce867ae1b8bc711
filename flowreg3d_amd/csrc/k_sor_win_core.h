// k_sor_win_core.h -- the temporally blocked ("window") form of the a_smooth == 1 SOR sweep: per-thread state and the
// step function, shared by the gfx950 kernel (k_sor_win.hip) and its CPU emulator (tools/emu/sor_win_emu.hip).
// Reference: core/level_solver_3d.py:383-540 (lexicographic sweep, omega = 1.95), :356-377 (psi_data every
// update_lag iterations), :246-259 (Neumann ghosts).
//
// Why.  k_sor_step (k_sor.hip) streams the frozen system (9 values) and the increments (3 read, 3 written, plus the
// neighbour planes) of every voxel through HBM on EVERY iteration: 100-140 B per voxel update at 5.7 TB/s, and nothing
// tried in three rounds moved it.  This kernel keeps a whole psi window -- W = update_lag
// iterations -- on chip: HBM sees the factors, the weights, the Laplacian terms and the increments once per WINDOW
// (about 25-40 B per voxel update with fp64 storage), the sweep becomes bound by its fp64 arithmetic, and fp64-grade
// storage costs nothing extra.
//
// How.  The dependences of the lexicographic sweep: voxel (k,j,i) of iteration t needs iteration t of (k,j,i-1),
// (k,j-1,i), (k-1,j,i) and iteration t-1 of itself and of (k,j,i+1), (k,j+1,i), (k+1,j,i).  A workgroup owns a tile of
// WIN_BK x WIN_BJ LINES (k,j) and marches along x.  In step s, "slot" q (iteration t0 + q of the window) of line
// (k,j) updates voxel i = s - 2q - k - j: all six neighbour values were produced in step s-1 (slot q: the -1
// neighbours, slot q-1: the +1 neighbours) and the voxel's own old value in step s-2 (slot q-1), so increments live
// for two steps: in registers for the line itself, in LDS (double-buffered, one barrier per step) for the four
// neighbouring lines.  The frozen system of a voxel is built by slot 0 and handed from slot to slot in registers
// (H: it is needed again 2, 4, .. steps later).
// Time skewing makes the tiles independent: slot q of tile (K,J) covers the lines [K BK - q, (K+1) BK - q) x
// [J BJ - q, (J+1) BJ - q), so every +1 neighbour of iteration t-1 lies in the SAME tile's slot q-1, and the only
// values a tile needs from others come from tiles with smaller K or J (complete before it starts: tiles are
// launched by diagonals K + J, consecutive windows three diagonals apart).  The thread <-> line mapping is cyclic:
// thread (a,b) handles line (K BK + a, J BJ + b) while it is inside the slot's range and the line BK (BJ) below once
// the range has moved past it, so a thread changes its line at most once per direction, every lane has work in every
// slot, and a line's history stays in one thread's registers except at that change.
// What crosses tiles (through HBM, written by the lower tile in an earlier launch):
//   * E[q]: the increments of slot q < last on the top row / column of the slot's range ("exports"); read as the -1
//     neighbour by the tile above (NB), and by the thread that takes the line over in slot q+1 (SW: the line's own
//     old value, its i+1 neighbour, and the +1 neighbour in the other direction, which sits on the same row);
//   * M: the frozen system of lines that will be taken over (built by slot 0 of the tile the line starts in);
//   * d: the final slot writes the increments in place; the next window reads them three diagonals later.
// Results are bit-identical to k_sor_step in every storage format (same per-voxel functions, same rounding to the
// storage format after every update): tests/test_gpu_sor_window.py.
#pragma once

#include "k_sor_core.h"

namespace fr3d {

#define WIN_BK 16
#define WIN_BJ 16
#define WIN_NT (WIN_BK * WIN_BJ)
#define WIN_WMAX 5   // slots (iterations) of a window kept on chip
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

// LDS image of one workgroup:
//  * O: outputs of step s-1 (and s) of slot -1 (the loader) .. W-1, structure of arrays;
//  * stg: every thread's PRIVATE staging of the operands it requested from HBM for the next step (the loader's record,
//    slot 0's factors / weights / Laplacian terms or stored system): requested at the top of a step, parked here at its
//    bottom, read where slot 0 needs them one step later.  Loop-carried LOADED registers made the compiler wait for the
//    data (and copy it into the loop's registers) right where the load is issued; values that live in LDS between
//    the steps leave only computed scalars as loop-carried state;
//  * imp: the neighbour values a thread gets from another tile instead of from the neighbouring thread (compact: only
//    5 of the 16 rows / columns of threads ever import): the slot reads its import entry INSTEAD of the neighbour's
//    output -- one select on the LDS address, not one per value.
#define WIN_NSPEC (5 * 16)  // threads that are ever the bottom of a slot's range in one direction: a (b) in {0, 12..15}
template <typename V, int W, int NSTG>
struct WinLds {
    V O[2][W + 1][3][WIN_NT];
    unsigned stg[NSTG][WIN_NT];
    V imp_zm[WIN_NSPEC][3], imp_ypsw[WIN_NSPEC][3];  // direction a: -1 neighbour in k; +1 neighbour in j of a taken-over line
    V imp_ym[WIN_NSPEC][3], imp_zpsw[WIN_NSPEC][3];  // direction b
    V imp_zptop[WIN_BJ][3], imp_yptop[WIN_BK][3];    // slot 0, top row / column
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
    // 32-bit words of the staged operands of one thread
    static constexpr int NW3 = sizeof(RawRec<S, 3>) / 4, NW9 = sizeof(RawRec<S, 9>) / 4, NW12 = sizeof(RawRec<S, 12>) / 4;
    static constexpr int NWT = sizeof(WT) / 4;
    static constexpr int NSTG = NW3 + (BUILD ? C * (NW12 + NWT) + NW3 : NW9);
    using Lds = WinLds<V, W, NSTG>;

    // ---- constant per thread ----
    int a, b, tid;
    int K, J, nsl;
    bool storeM;
    int qa, qb;    // the slot in which this thread is the bottom of the range in a / in b (>= W: never)
    int ixa, ixb;  // its entry in the import arrays of that direction
    int Z, Y, X;
    long long vD, vM, vA, vL;  // volume offsets (storage elements)
    int win, win_build;        // window of this workgroup / of the psi update its system belongs to (hook versions)

    // ---- loop-carried state: computed values only ----
    V Om1[W][3], Om2[W][3];  // slot q's line in slot q-1 (the loader for q = 0), steps s-1 and s-2: its i+1 neighbour, its own old value
    V H1[W][9], H2[W][9];    // frozen systems on their way to slot q: H2[q] is read in this step, H1[q] in the next

    FR3D_HD static int kline(int K, int a, int q) { return K * WIN_BK + a - (a >= WIN_BK - q ? WIN_BK : 0); }
    FR3D_HD static int jline(int J, int b, int q) { return J * WIN_BJ + b - (b >= WIN_BJ - q ? WIN_BJ : 0); }
    FR3D_HD int kq(int q) const { return kline(K, a, q); }
    FR3D_HD int jq(int q) const { return jline(J, b, q); }
    FR3D_HD bool line_ok(int k, int j) const { return k >= 0 && k < Z && j >= 0 && j < Y; }
    FR3D_HD static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
    // value select: `c ? x : y` on two members is an LVALUE conditional, which reaches the optimiser as a select of two
    // ADDRESSES inside the thread object and keeps the whole object in scratch memory (no scalar replacement)
    FR3D_HD static V sel(bool c, V x, V y) { return c ? x : y; }

    // a raw record <-> this thread's staging words
    template <typename T>
    FR3D_HD void stage_put(Lds &lds, int w0, const T &r) const
    {
        static_assert(sizeof(T) % 4 == 0, "staged types are whole words");
        struct Words { unsigned w[sizeof(T) / 4]; };
        const Words x = __builtin_bit_cast(Words, r);
#pragma unroll
        for (int n = 0; n < (int)(sizeof(T) / 4); n++) lds.stg[w0 + n][tid] = x.w[n];
    }
    template <typename T>
    FR3D_HD T stage_get(const Lds &lds, int w0) const
    {
        struct Words { unsigned w[sizeof(T) / 4]; };
        Words x;
#pragma unroll
        for (int n = 0; n < (int)(sizeof(T) / 4); n++) x.w[n] = lds.stg[w0 + n][tid];
        return __builtin_bit_cast(T, x);
    }

    FR3D_HD void init(const WinArgs<S> &wa, const WinTile &tl, int vol, int tid_, int win_, int win_build_)
    {
        const SorArgsT<S> &A_ = wa.a;
        tid = tid_;
        a = tid / WIN_BJ;
        b = tid % WIN_BJ;
        K = tl.K; J = tl.J;
        nsl = tl.info & 0xff;
        storeM = (tl.info >> 9) & 1;
        win = win_; win_build = win_build_;
        qa = (WIN_BK - a) % WIN_BK;
        qb = (WIN_BJ - b) % WIN_BJ;
        ixa = (qa < WIN_WMAX ? qa : 0) * WIN_BJ + b;
        ixb = (qb < WIN_WMAX ? qb : 0) * WIN_BK + a;
        Z = A_.sk.Z; Y = A_.sk.Y; X = A_.sk.X;
        vD = vol * A_.vsD; vM = vol * A_.vsM; vA = vol * A_.vsA; vL = vol * A_.vsL;
#pragma unroll
        for (int q = 0; q < W; q++) {
#pragma unroll
            for (int c = 0; c < 3; c++) Om1[q][c] = Om2[q][c] = 0;
#pragma unroll
            for (int n = 0; n < 9; n++) H1[q][n] = H2[q][n] = 0;
        }
    }
    // the staging words of this thread start out as zeros (they are read before the first request has been parked)
    FR3D_HD void init_lds(Lds &lds) const
    {
#pragma unroll
        for (int n = 0; n < NSTG; n++) lds.stg[n][tid] = 0u;
    }

    // first step in which any slot of this tile has a voxel, and the last one; the kernel starts WIN_LEAD steps
    // earlier, with every slot still idle, so that the request pipelines fill through the ordinary step code
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

    // What a thread requests from HBM at the top of step s for step s + 1 (all step-local).
    struct Req {
        RawRec<S, 3> L;                   // loader: d of the slot-0 line at voxel (s + 1) + 2 - k - j
        RawRec<S, 12> fr[C];              // BUILD: slot 0's factors, weights, Laplacian terms
        WT wt[C];
        RawRec<S, 3> lr;
        RawRec<S, 9> mx;                  // !BUILD: the stored system
        RawRec<S, 3> NBa, NBb;            // -1 neighbour across the tile edge (slot qa / qb)
        RawRec<S, 3> SWa, SWb;            // taken-over line: the i+1 neighbour of step s+1 (one step later the voxel's own old value)
        RawRec<S, 3> SXa, SXb;            // taken-over line: +1 neighbour in the other direction (same exported row)
        RawRec<S, 9> SMa, SMb;            // taken-over line: frozen system
        RawRec<S, 3> TOPa, TOPb;          // slot 0, top row / column: +1 neighbour of the previous window (d)
    };

    // Every thread's own-line operands: unconditional loads at clamped coordinates (a voxel outside the volume reads a
    // valid record that nobody uses), so that nothing merges a loaded value with an older one.
    FR3D_HD void request_own(const WinArgs<S> &wa, const Tab &tb, int s_next, Req &rq, const Hook &hk) const
    {
        const int k = kq(0), j = jq(0);
        const int kc = clampi(k, 0, Z - 1), jc = clampi(j, 0, Y - 1);
        {
            const int i = s_next + 2 - k - j;
            const long long e = win_index(tb, X, kc, jc, clampi(i, 0, X - 1));
            if (line_ok(k, j) && i >= 0 && i < X) hk.rd(WIN_ARR_D, e, win - 1);
            rq.L = ldraw<S, 3>(wa.a.d + vD, e);
        }
        const int i = s_next - k - j;
        const bool ok = line_ok(k, j) && i >= 0 && i < X;
        const long long e = win_index(tb, X, kc, jc, clampi(i, 0, X - 1));
        if constexpr (BUILD) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                rq.fr[c] = ldraw<S, 12>(wa.a.A[c] + vA, e);
                rq.wt[c] = wa.a.weight[c][e];
            }
            rq.lr = ldraw<S, 3>(wa.a.L + vL, e);
        } else {
            if (ok) hk.rd(WIN_ARR_M, e, win_build);
            rq.mx = ldraw<S, 9>(wa.a.M + vM, e);
        }
    }
    // slot 0, top row / column: the +1 neighbour belongs to the tile above; its value of the previous window is in d
    FR3D_HD void request_top(const WinArgs<S> &wa, const Tab &tb, int s_next, Req &rq, const Hook &hk) const
    {
        rq.TOPa = RawRec<S, 3>{};
        rq.TOPb = RawRec<S, 3>{};
        const int k = kq(0), j = jq(0), i = s_next - k - j;
        if (!(line_ok(k, j) && i >= 0 && i < X)) return;
        if (a == WIN_BK - 1 && k + 1 < Z) {
            const long long en = win_index(tb, X, k + 1, j, i);
            hk.rd(WIN_ARR_D, en, win - 1);
            rq.TOPa = ldraw<S, 3>(wa.a.d + vD, en);
        }
        if (b == WIN_BJ - 1 && j + 1 < Y) {
            const long long en = win_index(tb, X, k, j + 1, i);
            hk.rd(WIN_ARR_D, en, win - 1);
            rq.TOPb = ldraw<S, 3>(wa.a.d + vD, en);
        }
    }
    // the imports of a thread that is the bottom of the range in a (dir 0) or in b (dir 1), in its slot q = qa / qb
    template <int dir>
    FR3D_HD void request_special(const WinArgs<S> &wa, const Tab &tb, int q, int s_next, RawRec<S, 3> &NB,
                                 RawRec<S, 3> &SW, RawRec<S, 3> &SX, RawRec<S, 9> &SM, const Hook &hk) const
    {
        NB = RawRec<S, 3>{}; SW = RawRec<S, 3>{}; SX = RawRec<S, 3>{}; SM = RawRec<S, 9>{};
        if (q >= nsl) return;
        const int k = kq(q), j = jq(q), i = s_next - 2 * q - k - j;
        if (!line_ok(k, j)) return;
        const bool last = q == nsl - 1;
        const bool in = i >= 0 && i < X;
        // -1 neighbour across the edge: this window's slot-q value of the line below (exported, or final in d)
        const int kn = dir == 0 ? k - 1 : k, jn = dir == 0 ? j : j - 1;
        if (in && kn >= 0 && jn >= 0) {
            const long long e = win_index(tb, X, kn, jn, i);
            hk.rd(last ? WIN_ARR_D : WIN_ARR_E0 + q, e, win);
            const S *src = last ? wa.a.d + vD : wa.E + (long long)q * wa.strideE + vD;
            NB = ldraw<S, 3>(src, e);
        }
        if (q == 0) return;
        const S *Eprev = wa.E + (long long)(q - 1) * wa.strideE + vD;
        // the line is taken over from the tile below: its slot q-1 values come from that tile's exports
        // (both directions change in the same slot: direction a loads the line's own values)
        if (dir == 0 || qa != qb) {
            // voxel i + 1: the i+1 neighbour of step s_next, and one step later the voxel's own old value
            if (i + 1 >= 0 && i + 1 < X) {
                const long long e = win_index(tb, X, k, j, i + 1);
                hk.rd(WIN_ARR_E0 + q - 1, e, win);
                SW = ldraw<S, 3>(Eprev, e);
            }
            if (in) {
                const long long e = win_index(tb, X, k, j, i);
                hk.rd(WIN_ARR_M, e, win_build);
                SM = ldraw<S, 9>(wa.a.M + vM, e);
            }
        }
        // the +1 neighbour in the OTHER direction lies on the same exported row / column
        const int kx = dir == 0 ? k : k + 1, jx = dir == 0 ? j + 1 : j;
        if (in && kx < Z && jx < Y) {
            const long long e = win_index(tb, X, kx, jx, i);
            hk.rd(WIN_ARR_E0 + q - 1, e, win);
            SX = ldraw<S, 3>(Eprev, e);
        }
    }

    // ---- one slot of one step ----
    // The line's own history (Om1/Om2) and its system record (H2) already hold the imported values where the line was
    // taken over (end of step()), so the slot body has no special cases except WHERE the four cross-line neighbours
    // are read from, and the ghosts at the volume's faces (a wave-uniform branch: most waves have none).
    template <int q>
    FR3D_HD void slot(const WinArgs<S> &wa, const Tab &tb, int s, Lds &lds, int pp, V (&Onew)[W + 1][3], V (&Hnew)[W][9],
                      const Hook &hk)
    {
        const SorArgsT<S> &A_ = wa.a;
        const int k = kq(q), j = jq(q), i = s - 2 * q - k - j;
        const bool spa = q == qa, spb = q == qb;
        const bool active = q < nsl && line_ok(k, j) && i >= 0 && i < X;
        V own[3], xp[3], xm[3], yp[3], ym[3], zp[3], zm[3], sys[9];
        const int t_am = ((a + WIN_BK - 1) % WIN_BK) * WIN_BJ + b, t_ap = ((a + 1) % WIN_BK) * WIN_BJ + b;
        const int t_bm = a * WIN_BJ + (b + WIN_BJ - 1) % WIN_BJ, t_bp = a * WIN_BJ + (b + 1) % WIN_BJ;
        // -1 neighbours: this slot, step s-1;  +1 neighbours: slot q-1 (q = 0: the loader), step s-1; or the import
        // entry (component stride 1 there, WIN_NT in the output arrays)
        const V *pzm = spa ? &lds.imp_zm[ixa][0] : &lds.O[pp][q + 1][0][t_am];
        const V *pym = spb ? &lds.imp_ym[ixb][0] : &lds.O[pp][q + 1][0][t_bm];
        const int szm = spa ? 1 : WIN_NT, sym = spb ? 1 : WIN_NT;
        const V *pzp, *pyp;
        int szp, syp;
        if (q == 0) {
            const bool ta = a == WIN_BK - 1, tb_ = b == WIN_BJ - 1;
            pzp = ta ? &lds.imp_zptop[b][0] : &lds.O[pp][q][0][t_ap];
            pyp = tb_ ? &lds.imp_yptop[a][0] : &lds.O[pp][q][0][t_bp];
            szp = ta ? 1 : WIN_NT; syp = tb_ ? 1 : WIN_NT;
        } else {
            pzp = spb ? &lds.imp_zpsw[ixb][0] : &lds.O[pp][q][0][t_ap];
            pyp = spa ? &lds.imp_ypsw[ixa][0] : &lds.O[pp][q][0][t_bp];
            szp = spb ? 1 : WIN_NT; syp = spa ? 1 : WIN_NT;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            own[c] = Om2[q][c];
            xp[c] = Om1[q][c];
            // i-1 neighbour: this thread's own output of step s-1 (Om1[q + 1] holds another line's import where
            // slot q+1 takes a line over)
            xm[c] = lds.O[pp][q + 1][c][tid];
            zm[c] = pzm[c * szm]; ym[c] = pym[c * sym];
            zp[c] = pzp[c * szp]; yp[c] = pyp[c * syp];
        }
        // Neumann ghosts (set_boundary_3d :246-259): a missing neighbour is the voxel's own old value
        const bool ghost = i <= 0 || i >= X - 1 || j == 0 || j == Y - 1 || k == 0 || k == Z - 1;
        if (WIN_ANY(ghost)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                xp[c] = sel(i < X - 1, xp[c], own[c]);
                xm[c] = sel(i > 0, xm[c], own[c]);
                zm[c] = sel(k > 0, zm[c], own[c]);
                ym[c] = sel(j > 0, ym[c], own[c]);
                zp[c] = sel(k < Z - 1, zp[c], own[c]);
                yp[c] = sel(j < Y - 1, yp[c], own[c]);
            }
        }
        // the frozen system
        if (q == 0) {
            Rec<S, 9> mr;
            if constexpr (BUILD) {
                SorAcc<R> acc;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const RawRec<S, 12> f = stage_get<RawRec<S, 12>>(lds, NW3 + c * (NW12 + NWT));
                    const WT w = stage_get<WT>(lds, NW3 + c * (NW12 + NWT) + NW12);
                    sor_accum_channel<R, S>(f.dec(), (double)w, A_.a_data[c], (R)own[0], (R)own[1], (R)own[2], acc);
                }
                mr = sor_finish_system<R, S>(acc, stage_get<RawRec<S, 3>>(lds, NW3 + C * (NW12 + NWT)).dec());
                // lines that another tile takes over later need the record in memory
                if (active && (storeM || a >= WIN_BK - W + 1 || b >= WIN_BJ - W + 1)) {
                    const long long e = win_index(tb, X, k, j, i);
                    hk.wr(WIN_ARR_M, e, win);
                    strec<S, 9>(A_.M + vM, e, mr);
                }
            } else {
                mr = stage_get<RawRec<S, 9>>(lds, NW3).dec();
            }
#pragma unroll
            for (int n = 0; n < 9; n++) sys[n] = mr.v[n];
        } else {
#pragma unroll
            for (int n = 0; n < 9; n++) sys[n] = H2[q][n];
        }
        R m[9];
#pragma unroll
        for (int n = 0; n < 9; n++) m[n] = (R)sys[n];
        R du1, dv1, dw1;
        sor_relax<R>(m, A_.ax, A_.ay, A_.az, (R)xm[0] + (R)xp[0], (R)xm[1] + (R)xp[1], (R)xm[2] + (R)xp[2],
                     (R)ym[0] + (R)yp[0], (R)ym[1] + (R)yp[1], (R)ym[2] + (R)yp[2], (R)zm[0] + (R)zp[0],
                     (R)zm[1] + (R)zp[1], (R)zm[2] + (R)zp[2], (R)own[0], (R)own[1], (R)own[2], du1, dv1, dw1);
        Rec<S, 3> out;
        out.v[0] = Sto<S>::quant(du1);
        out.v[1] = Sto<S>::quant(dv1);
        out.v[2] = Sto<S>::quant(dw1);
#pragma unroll
        for (int c = 0; c < 3; c++) Onew[q + 1][c] = out.v[c];
        if (active) {
            const bool fin = q == nsl - 1;
            bool exp_ = false;
            if constexpr (q < W - 1) exp_ = a == WIN_BK - 1 - q || b == WIN_BJ - 1 - q;
            if (fin || exp_) {
                const long long e = win_index(tb, X, k, j, i);
                hk.wr(fin ? WIN_ARR_D : WIN_ARR_E0 + q, e, win);
                S *dst = fin ? A_.d + vD : wa.E + (long long)q * wa.strideE + vD;
                strec<S, 3>(dst, e, out);
            }
        }
        // the record travels on: slot q+1 reads it two steps from now
        if constexpr (q + 1 < W) {
#pragma unroll
            for (int n = 0; n < 9; n++) Hnew[q + 1][n] = sys[n];
        }
        WIN_SCHED_FENCE();
    }

    template <int q>
    FR3D_HD void slots_up(const WinArgs<S> &wa, const Tab &tb, int s, Lds &lds, int pp, V (&Onew)[W + 1][3],
                          V (&Hnew)[W][9], const Hook &hk)
    {
        slot<q>(wa, tb, s, lds, pp, Onew, Hnew, hk);
        if constexpr (q + 1 < W) slots_up<q + 1>(wa, tb, s, lds, pp, Onew, Hnew, hk);
    }

    // step s: everything step s+1 needs from HBM is requested first (step-local values: they have the whole step to
    // arrive), then every slot reads the LDS image of step s-1 and computes; at the bottom the outputs are published,
    // the requests are parked in this thread's staging / import entries, and the state moves on one step -- where a
    // line is taken over from another tile (slot qa / qb of this thread) its history and its system are REPLACED by the
    // imports there, so the slots never look at them.  The caller puts a barrier behind it.
    FR3D_HD void step(const WinArgs<S> &wa, const Tab &tb, int s, Lds &lds, const Hook &hk)
    {
        V Onew[W + 1][3], Hnew[W][9];
        {
            const Rec<S, 3> l = stage_get<RawRec<S, 3>>(lds, 0).dec();  // the loader's output of this step
#pragma unroll
            for (int c = 0; c < 3; c++) Onew[0][c] = l.v[c];
        }
        // slot 0 (the psi update: the step's register peak) first, the requests behind it: they have the other slots'
        // arithmetic to arrive and do not sit in registers during the peak
        slot<0>(wa, tb, s, lds, (s - 1) & 1, Onew, Hnew, hk);
        Req rq;
        request_own(wa, tb, s + 1, rq, hk);
        if (qa < W) request_special<0>(wa, tb, qa, s + 1, rq.NBa, rq.SWa, rq.SXa, rq.SMa, hk);
        if (qb < W) request_special<1>(wa, tb, qb, s + 1, rq.NBb, rq.SWb, rq.SXb, rq.SMb, hk);
        if (a == WIN_BK - 1 || b == WIN_BJ - 1) request_top(wa, tb, s + 1, rq, hk);
        WIN_SCHED_FENCE();
        if constexpr (W > 1) slots_up<1>(wa, tb, s, lds, (s - 1) & 1, Onew, Hnew, hk);
        const int pc = s & 1;
#pragma unroll
        for (int q = 0; q <= W; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) lds.O[pc][q][c][tid] = Onew[q][c];
        // park the requests
        stage_put(lds, 0, rq.L);
        if constexpr (BUILD) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                stage_put(lds, NW3 + c * (NW12 + NWT), rq.fr[c]);
                stage_put(lds, NW3 + c * (NW12 + NWT) + NW12, rq.wt[c]);
            }
            stage_put(lds, NW3 + C * (NW12 + NWT), rq.lr);
        } else {
            stage_put(lds, NW3, rq.mx);
        }
        Rec<S, 3> swa_, swb_;
        Rec<S, 9> sma_, smb_;
        if (qa < W) {
            const Rec<S, 3> nb = rq.NBa.dec(), sx = rq.SXa.dec();
            swa_ = rq.SWa.dec(); sma_ = rq.SMa.dec();
#pragma unroll
            for (int c = 0; c < 3; c++) { lds.imp_zm[ixa][c] = nb.v[c]; lds.imp_ypsw[ixa][c] = sx.v[c]; }
        }
        if (qb < W) {
            const Rec<S, 3> nb = rq.NBb.dec(), sx = rq.SXb.dec();
            swb_ = rq.SWb.dec(); smb_ = rq.SMb.dec();
#pragma unroll
            for (int c = 0; c < 3; c++) { lds.imp_ym[ixb][c] = nb.v[c]; lds.imp_zpsw[ixb][c] = sx.v[c]; }
        }
        if (a == WIN_BK - 1) {
            const Rec<S, 3> t = rq.TOPa.dec();
#pragma unroll
            for (int c = 0; c < 3; c++) lds.imp_zptop[b][c] = t.v[c];
        }
        if (b == WIN_BJ - 1) {
            const Rec<S, 3> t = rq.TOPb.dec();
#pragma unroll
            for (int c = 0; c < 3; c++) lds.imp_yptop[a][c] = t.v[c];
        }
        // the state moves on; taken-over lines: slot q's history of the next step is the import (voxel i+1 now, its own
        // old value one step later), its system the imported record
#pragma unroll
        for (int q = 0; q < W; q++) {
            const bool ia = q >= 1 && q == qa, ib = q >= 1 && q == qb && qa != qb;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                Om2[q][c] = Om1[q][c];
                Om1[q][c] = sel(ia, swa_.v[c], sel(ib, swb_.v[c], Onew[q][c]));
            }
            if (q >= 1) {
#pragma unroll
                for (int n = 0; n < 9; n++) {
                    H2[q][n] = sel(ia, sma_.v[n], sel(ib, smb_.v[n], H1[q][n]));
                    H1[q][n] = Hnew[q][n];
                }
            }
        }
    }
};

#define WIN_LEAD 4  // steps the kernel runs ahead of the tile's first voxel: the prefetch pipelines fill (3 needed)

}  // namespace fr3d
