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
    S *E[WIN_WMAX - 1];      // exports of slots 0 .. W-2, records of 3, laid out and strided like d
};

// LDS image of one workgroup: outputs of step s-1 (and s) of slot -1 (the loader) .. W-1, structure of arrays
template <typename V, int W>
struct WinLds {
    V O[2][W + 1][3][WIN_NT];
};

// Memory hook: the device build reads and writes plainly; the emulator's hook checks that a value read was written
// by the expected window (and not by a workgroup of the same launch).
struct WinNoHook {
    FR3D_HD void rd(int, long long, int) const {}
    FR3D_HD void wr(int, long long, int) const {}
};
enum { WIN_ARR_D = 0, WIN_ARR_M = 1, WIN_ARR_E0 = 2 };  // array ids of the hook (E[q] = WIN_ARR_E0 + q)

template <typename R, typename S, int C, int W, typename Hook = WinNoHook>
struct WinThread {
    using V = typename Sto<S>::val;
    static_assert(W >= 1 && W <= WIN_WMAX, "window slots");
    static_assert(C >= 1, "channel count is a template parameter");

    // ---- constant per thread ----
    int a, b, tid;
    int K, J, t0, nsl;
    bool build, storeM;
    int qa, qb;  // the slot in which this thread is the bottom of the range in a / in b (>= W: never)
    int Z, Y, X;
    long long vD, vM, vA, vL;  // volume offsets (storage elements)

    // ---- state ----
    V Om1[W + 1][3], Om2[W + 1][3];  // outputs of slot q-1 .. (index q+1) in steps s-1 and s-2
    V H[W][9];                       // H[q]: system record slot q needs in THIS step parity ... see hand-off below
    V H2[W][9];                      // the other parity
    // prefetched for the next use (same program point, one step later)
    Rec<S, 3> Lnext;                 // loader: d_in of the slot-0 line at voxel i0 + 3 of the step it is issued in
    Rec<S, 12> fr[C];
    double wt[C];
    Rec<S, 3> lr;
    Rec<S, 9> mx;
    Rec<S, 3> NBa, NBb;              // -1 neighbour across the tile edge (slot qa / qb)
    Rec<S, 3> SWa0, SWa1, SWb0, SWb1;  // taken-over line: own old value (voxel i), its i+1 neighbour
    Rec<S, 3> SXa, SXb;              // taken-over line: +1 neighbour in the other direction (same exported row)
    Rec<S, 9> SMa, SMb;              // taken-over line: frozen system
    Rec<S, 3> TOPa, TOPb;            // slot 0, top row / column: +1 neighbour of the previous window (d)

    FR3D_HD static int kline(int K, int a, int q) { return K * WIN_BK + a - (a >= WIN_BK - q ? WIN_BK : 0); }
    FR3D_HD static int jline(int J, int b, int q) { return J * WIN_BJ + b - (b >= WIN_BJ - q ? WIN_BJ : 0); }
    FR3D_HD int kq(int q) const { return kline(K, a, q); }
    FR3D_HD int jq(int q) const { return jline(J, b, q); }
    FR3D_HD bool line_ok(int k, int j) const { return k >= 0 && k < Z && j >= 0 && j < Y; }

    template <int N>
    FR3D_HD static void zero(Rec<S, N> &r)
    {
#pragma unroll
        for (int n = 0; n < N; n++) r.v[n] = 0;
    }

    // record loads / stores at voxel (k,j,i) of one volume's slab; `arr`: hook id
    template <int N>
    FR3D_HD Rec<S, N> ld(const S *base, long long vol_off, const Skew &sk, int k, int j, int i, int arr, int want,
                         const Hook &hk) const
    {
        const long long e = sk_index(sk, k, j, i);
        hk.rd(arr, e, want);
        return ldrec<S, N>(base + vol_off, e);
    }
    template <int N>
    FR3D_HD void st(S *base, long long vol_off, const Skew &sk, int k, int j, int i, const Rec<S, N> &r, int arr,
                    int ver, const Hook &hk) const
    {
        const long long e = sk_index(sk, k, j, i);
        hk.wr(arr, e, ver);
        strec<S, N>(base + vol_off, e, r);
    }

    // window index of this workgroup's window / of the psi update its system belongs to (hook versions)
    int win, win_build;

    FR3D_HD void init(const WinArgs<S> &wa, const WinTile &tl, int vol, int tid_, int win_, int win_build_)
    {
        const SorArgsT<S> &A_ = wa.a;
        tid = tid_;
        a = tid / WIN_BJ;
        b = tid % WIN_BJ;
        K = tl.K; J = tl.J; t0 = tl.t0;
        nsl = tl.info & 0xff;
        build = (tl.info >> 8) & 1;
        storeM = (tl.info >> 9) & 1;
        win = win_; win_build = win_build_;
        qa = (WIN_BK - a) % WIN_BK;
        qb = (WIN_BJ - b) % WIN_BJ;
        Z = A_.sk.Z; Y = A_.sk.Y; X = A_.sk.X;
        vD = vol * A_.vsD; vM = vol * A_.vsM; vA = vol * A_.vsA; vL = vol * A_.vsL;
#pragma unroll
        for (int q = 0; q <= W; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) Om1[q][c] = Om2[q][c] = 0;
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int n = 0; n < 9; n++) H[q][n] = H2[q][n] = 0;
        zero(Lnext); zero(lr); zero(mx); zero(NBa); zero(NBb); zero(SWa0); zero(SWa1); zero(SWb0); zero(SWb1);
        zero(SXa); zero(SXb); zero(SMa); zero(SMb); zero(TOPa); zero(TOPb);
#pragma unroll
        for (int c = 0; c < C; c++) { zero(fr[c]); wt[c] = 0; }
    }

    // first step in which any slot of this tile has a voxel, and the last one
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

    // ---- the prefetches: each is (re)issued right after its value for the current step has been consumed, for the
    // voxel of the NEXT step; prime() issues them all for the first step ----
    FR3D_HD void load_loader(const WinArgs<S> &wa, int s_next, const Hook &hk)
    {
        // O_{-1}(s) = d_in of the slot-0 line at voxel s + 2 - k - j
        const int k = kq(0), j = jq(0), i = s_next + 2 - k - j;
        if (line_ok(k, j) && i >= 0 && i < X) Lnext = ld<3>(wa.a.d, vD, wa.a.sk, k, j, i, WIN_ARR_D, win - 1, hk);
    }
    FR3D_HD void load_slot0(const WinArgs<S> &wa, int s_next, const Hook &hk)
    {
        const int k = kq(0), j = jq(0), i = s_next - k - j;
        if (!(line_ok(k, j) && i >= 0 && i < X)) return;
        const long long e = sk_index(wa.a.sk, k, j, i);
        if (build) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                fr[c] = ldrec<S, 12>(wa.a.A[c] + vA, e);
                wt[c] = (double)wa.a.weight[c][e];
            }
            lr = ldrec<S, 3>(wa.a.L + vL, e);
        } else {
            hk.rd(WIN_ARR_M, e, win_build);
            mx = ldrec<S, 9>(wa.a.M + vM, e);
        }
        // slot 0, top row / column: the +1 neighbour belongs to the tile above; its value of the previous window is in d
        if (a == WIN_BK - 1 && k + 1 < Z) TOPa = ld<3>(wa.a.d, vD, wa.a.sk, k + 1, j, i, WIN_ARR_D, win - 1, hk);
        if (b == WIN_BJ - 1 && j + 1 < Y) TOPb = ld<3>(wa.a.d, vD, wa.a.sk, k, j + 1, i, WIN_ARR_D, win - 1, hk);
    }
    // the slot-q imports of a thread that is the bottom of the range in a (dir 0) or in b (dir 1)
    FR3D_HD void load_special(const WinArgs<S> &wa, int q, int dir, int s_next, const Hook &hk)
    {
        if (q >= nsl) return;
        const int k = kq(q), j = jq(q), i = s_next - 2 * q - k - j;
        if (!line_ok(k, j)) return;
        const Skew &sk = wa.a.sk;
        const bool last = q == nsl - 1;
        // -1 neighbour across the edge: this window's slot-q value of the line below (exported, or final in d)
        const int kn = dir == 0 ? k - 1 : k, jn = dir == 0 ? j : j - 1;
        if (i >= 0 && i < X && kn >= 0 && jn >= 0) {
            const S *src = last ? wa.a.d : wa.E[q];
            const Rec<S, 3> r = ld<3>(src, vD, sk, kn, jn, i, last ? WIN_ARR_D : WIN_ARR_E0 + q, win, hk);
            if (dir == 0) NBa = r; else NBb = r;
        }
        if (q == 0) return;
        // the line is taken over from the tile below: its slot q-1 values come from that tile's exports.
        // (both directions change in the same slot: direction a loads the line's own values)
        const bool own = dir == 0 || qa != qb;
        if (own) {
            // own old value of the NEXT step's voxel is this step's i+1 neighbour: load voxel i + 1 only
            if (i + 1 >= 0 && i + 1 < X) {
                const Rec<S, 3> r = ld<3>(wa.E[q - 1], vD, sk, k, j, i + 1, WIN_ARR_E0 + q - 1, win, hk);
                if (dir == 0) SWa1 = r; else SWb1 = r;
            }
            if (i >= 0 && i < X) {
                const long long e = sk_index(sk, k, j, i);
                hk.rd(WIN_ARR_M, e, win_build);
                const Rec<S, 9> r = ldrec<S, 9>(wa.a.M + vM, e);
                if (dir == 0) SMa = r; else SMb = r;
            }
        }
        // the +1 neighbour in the OTHER direction lies on the same exported row / column
        const int kx = dir == 0 ? k : k + 1, jx = dir == 0 ? j + 1 : j;
        if (i >= 0 && i < X && kx < Z && jx < Y) {
            const Rec<S, 3> r = ld<3>(wa.E[q - 1], vD, sk, kx, jx, i, WIN_ARR_E0 + q - 1, win, hk);
            if (dir == 0) SXa = r; else SXb = r;
        }
    }

    FR3D_HD void prime(const WinArgs<S> &wa, int s_first, const Hook &hk)
    {
        // loader pipeline: O_{-1}(s_first - 2), O_{-1}(s_first - 1) and the value published in step s_first
        load_loader(wa, s_first - 2, hk);
#pragma unroll
        for (int c = 0; c < 3; c++) Om2[0][c] = Lnext.v[c];
        load_loader(wa, s_first - 1, hk);
#pragma unroll
        for (int c = 0; c < 3; c++) Om1[0][c] = Lnext.v[c];
        load_loader(wa, s_first, hk);
        load_slot0(wa, s_first, hk);
        // taken-over lines: SW?1 of step s - 1 becomes SW?0 of step s
        if (qa >= 1 && qa < W) { load_special(wa, qa, 0, s_first - 1, hk); SWa0 = SWa1; }
        if (qb >= 1 && qb < W) { load_special(wa, qb, 1, s_first - 1, hk); SWb0 = SWb1; }
        if (qa < W) load_special(wa, qa, 0, s_first, hk);
        if (qb < W) load_special(wa, qb, 1, s_first, hk);
    }

    // ---- one step ----
    // `prev` / `cur`: LDS images of steps s-1 and s.  Slots are processed from the last to the first so that the
    // system record of slot q can be handed to slot q+1 (for step s+2) once slot q+1 has used its own.
    template <int q>
    FR3D_HD void slot(const WinArgs<S> &wa, int s, const V (*prev)[3][WIN_NT], V (&Onew)[W + 1][3], V (&Hp)[W][9],
                      const Hook &hk)
    {
        const SorArgsT<S> &A_ = wa.a;
        const int k = kq(q), j = jq(q), i = s - 2 * q - k - j;
        const bool spa = q == qa, spb = q == qb;
        const bool active = q < nsl && line_ok(k, j) && i >= 0 && i < X;
        V sys[9];
        if (active) {
            V own[3], xp[3], xm[3], yp[3], ym[3], zp[3], zm[3];
            const bool swa = q >= 1 && spa, swb = q >= 1 && spb && !spa;
            // the line's own history: registers, or the import of the slot in which the line was taken over
#pragma unroll
            for (int c = 0; c < 3; c++) {
                own[c] = swa ? SWa0.v[c] : (swb ? SWb0.v[c] : Om2[q][c]);
                const V xpv = swa ? SWa1.v[c] : (swb ? SWb1.v[c] : Om1[q][c]);
                xp[c] = i < X - 1 ? xpv : own[c];
                xm[c] = i > 0 ? Om1[q + 1][c] : own[c];
            }
            const int t_am = ((a + WIN_BK - 1) % WIN_BK) * WIN_BJ + b, t_ap = ((a + 1) % WIN_BK) * WIN_BJ + b;
            const int t_bm = a * WIN_BJ + (b + WIN_BJ - 1) % WIN_BJ, t_bp = a * WIN_BJ + (b + 1) % WIN_BJ;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // -1 neighbours: this slot, step s-1
                zm[c] = k > 0 ? (spa ? NBa.v[c] : prev[q + 1][c][t_am]) : own[c];
                ym[c] = j > 0 ? (spb ? NBb.v[c] : prev[q + 1][c][t_bm]) : own[c];
                // +1 neighbours: slot q-1 (q = 0: the loader), step s-1
                V zpv, ypv;
                if (q == 0) {
                    zpv = a == WIN_BK - 1 ? TOPa.v[c] : prev[q][c][t_ap];
                    ypv = b == WIN_BJ - 1 ? TOPb.v[c] : prev[q][c][t_bp];
                } else {
                    zpv = spb ? SXb.v[c] : prev[q][c][t_ap];
                    ypv = spa ? SXa.v[c] : prev[q][c][t_bp];
                }
                zp[c] = k < Z - 1 ? zpv : own[c];
                yp[c] = j < Y - 1 ? ypv : own[c];
            }
            // the frozen system
            if (q == 0) {
                Rec<S, 9> mr;
                if (build) {
                    SorAcc<R> acc;
#pragma unroll
                    for (int c = 0; c < C; c++)
                        sor_accum_channel<R, S>(fr[c], wt[c], A_.a_data[c], (R)own[0], (R)own[1], (R)own[2], acc);
                    mr = sor_finish_system<R, S>(acc, lr);
                    // lines that another tile takes over later need the record in memory
                    if (storeM || a >= WIN_BK - W + 1 || b >= WIN_BJ - W + 1) st<9>(A_.M, vM, A_.sk, k, j, i, mr, WIN_ARR_M, win, hk);
                } else {
                    mr = mx;
                }
#pragma unroll
                for (int n = 0; n < 9; n++) sys[n] = mr.v[n];
            } else {
#pragma unroll
                for (int n = 0; n < 9; n++) sys[n] = swa ? SMa.v[n] : (swb ? SMb.v[n] : Hp[q][n]);
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
            if (q == nsl - 1) st<3>(A_.d, vD, A_.sk, k, j, i, out, WIN_ARR_D, win, hk);
            else if constexpr (q < W - 1) {
                if (a == WIN_BK - 1 - q || b == WIN_BJ - 1 - q) st<3>(wa.E[q], vD, A_.sk, k, j, i, out, WIN_ARR_E0 + q, win, hk);
            }
        }
        // hand the record on: slot q+1 needs it in step s+2 (same parity); slot q+1 has already run in this step
        if (q + 1 < W) {
#pragma unroll
            for (int n = 0; n < 9; n++) Hp[q + 1][n] = active ? sys[n] : Hp[q + 1][n];
        }
        // refill this slot's prefetch registers for step s+1
        if (q == 0) load_slot0(wa, s + 1, hk);
        if (spa || spb) {
            if (q >= 1) {
                if (spa) SWa0 = SWa1;
                if (spb) SWb0 = SWb1;
            }
            if (spa) load_special(wa, q, 0, s + 1, hk);
            if (spb) load_special(wa, q, 1, s + 1, hk);
        }
    }

    template <int q>
    FR3D_HD void slots_down(const WinArgs<S> &wa, int s, const V (*prev)[3][WIN_NT], V (&Onew)[W + 1][3], V (&Hp)[W][9],
                            const Hook &hk)
    {
        slot<q>(wa, s, prev, Onew, Hp, hk);
        if constexpr (q > 0) slots_down<q - 1>(wa, s, prev, Onew, Hp, hk);
    }

    // step s: read `prev`, compute every slot, publish the outputs in `cur`; the caller puts a barrier behind it
    FR3D_HD void step(const WinArgs<S> &wa, int s, const V (*prev)[3][WIN_NT], V (*cur)[3][WIN_NT], const Hook &hk)
    {
        V Onew[W + 1][3];
#pragma unroll
        for (int q = 0; q <= W; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) Onew[q][c] = 0;
        // the loader's output of this step was requested one step ago
#pragma unroll
        for (int c = 0; c < 3; c++) Onew[0][c] = Lnext.v[c];
        load_loader(wa, s + 1, hk);
        if (s & 1) slots_down<W - 1>(wa, s, prev, Onew, H2, hk);
        else slots_down<W - 1>(wa, s, prev, Onew, H, hk);
#pragma unroll
        for (int q = 0; q <= W; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                cur[q][c][tid] = Onew[q][c];
                Om2[q][c] = Om1[q][c];
                Om1[q][c] = Onew[q][c];
            }
    }
};

}  // namespace fr3d
