// sor_win_emu.hip -- CPU emulation of the window sweep kernel (flowreg3d_amd/csrc/k_sor_win_core.h), thread by
// thread and workgroup by workgroup in launch order, against the plain lexicographic sweep built from the same
// per-voxel functions.  Development tool: the kernel's indexing, hand-offs, imports / exports and launch schedule
// are checked here, where an out-of-range access is a host error and not a GPU fault.  The memory hook verifies
// that every value read across workgroups was written by the expected window in an EARLIER launch.
//   build: hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iflowreg3d_amd/csrc -o /tmp/sor_win_emu tools/emu/sor_win_emu.hip
//   run:   /tmp/sor_win_emu Z Y X iterations lag [mode 0..3] [C]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

#include "k_sor_win_sched.h"

using namespace fr3d;

struct Shadow {
    std::vector<int> ver, launch, block;
    void init(size_t n) { ver.assign(n, -1); launch.assign(n, -1); block.assign(n, -1); }
};
static long long g_errors = 0;
struct EmuHook {
    Shadow *sh;  // WIN_ARR_D, WIN_ARR_M, WIN_ARR_E0 + q
    int launch, block;
    void rd(int arr, long long e, int want) const
    {
        Shadow &s = sh[arr];
        if (e < 0 || e >= (long long)s.ver.size()) { if (g_errors++ < 20) printf("  OOB read arr %d e %lld\n", arr, e); return; }
        if (s.ver[e] != want) { if (g_errors++ < 20) printf("  version: arr %d e %lld has %d want %d (launch %d block %d)\n", arr, e, s.ver[e], want, launch, block); }
        else if (s.launch[e] == launch && s.block[e] != block) { if (g_errors++ < 20) printf("  race: arr %d e %lld written by block %d in this launch %d, read by %d\n", arr, e, s.block[e], launch, block); }
    }
    void wr(int arr, long long e, int ver) const
    {
        Shadow &s = sh[arr];
        if (e < 0 || e >= (long long)s.ver.size()) { if (g_errors++ < 20) printf("  OOB write arr %d e %lld\n", arr, e); return; }
        if (s.launch[e] == launch && s.block[e] != block) { if (g_errors++ < 20) printf("  write race arr %d e %lld\n", arr, e); }
        s.ver[e] = ver; s.launch[e] = launch; s.block[e] = block;
    }
};

template <typename R, typename S, int C>
static int run(int Z, int Y, int X, int T, int lag, unsigned seed)
{
    using V = typename Sto<S>::val;
    using WT = typename StoWt<S>::type;
    std::vector<long long> pb;
    std::vector<int> cp;
    const long long total = make_compact_tables(Z, Y, X, pb, cp);
    Skew sk = make_skew(Z, Y, X);
    sk.total = total;
    sk.pb = pb.data();
    sk.cp = cp.data();
    const size_t e3 = (size_t)sto_elems<S>(total * 3), e9 = 3 * e3, e12 = 4 * e3;
    std::mt19937 rng(seed);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::vector<S> A((size_t)e12 * C), L(e3), M0(e9), d0(e3);
    std::vector<WT> w((size_t)total * C);
    memset(M0.data(), 0, M0.size() * sizeof(S));
    memset(d0.data(), 0, d0.size() * sizeof(S));
    memset(A.data(), 0, A.size() * sizeof(S));
    memset(L.data(), 0, L.size() * sizeof(S));
    for (int k = 0; k < Z; k++) for (int j = 0; j < Y; j++) for (int i = 0; i < X; i++) {
        const long long e = sk_index(sk, k, j, i);
        for (int c = 0; c < C; c++) {
            Rec<S, 12> fr;
            for (int q = 0; q < 12; q++) fr.v[q] = Sto<S>::quant((q % 4 == 3 ? 0.2 : 2.0) * N01(rng));
            strec<S, 12>(A.data() + (size_t)c * e12, e, fr);
            w[(size_t)c * total + e] = (WT)(1.0 / C);
        }
        Rec<S, 3> lr;
        for (int q = 0; q < 3; q++) lr.v[q] = Sto<S>::quant(0.05 * N01(rng));
        strec<S, 3>(L.data(), e, lr);
    }
    const double ax = 0.25, ay = 0.31, az = 0.2;
    const double adc[FR3D_MAX_CHANNELS] = {0.45, 0.6, 0.45, 1.0, 0.45, 0.45, 0.45, 0.45};

    // ---- reference: the lexicographic sweep (what k_sor_step reproduces) ----
    std::vector<S> Mr = M0, dr = d0;
    for (int t = 0; t < T; t++) {
        const bool upd = t % lag == 0;
        for (int k = 0; k < Z; k++) for (int j = 0; j < Y; j++) for (int i = 0; i < X; i++) {
            const long long e = sk_index(sk, k, j, i);
            const Rec<S, 3> q0 = ldrec<S, 3>(dr.data(), e);
            auto nb = [&](bool ok, int kk, int jj, int ii) { return ok ? ldrec<S, 3>(dr.data(), sk_index(sk, kk, jj, ii)) : q0; };
            const Rec<S, 3> xm = nb(i > 0, k, j, i - 1), xp = nb(i < X - 1, k, j, i + 1), ym = nb(j > 0, k, j - 1, i),
                            yp = nb(j < Y - 1, k, j + 1, i), zm = nb(k > 0, k - 1, j, i), zp = nb(k < Z - 1, k + 1, j, i);
            Rec<S, 9> mr;
            if (upd) {
                SorAcc<R> acc;
                for (int c = 0; c < C; c++)
                    sor_accum_channel<R, S>(ldrec<S, 12>(A.data() + (size_t)c * e12, e), (double)w[(size_t)c * total + e], adc[c],
                                            (R)q0.v[0], (R)q0.v[1], (R)q0.v[2], acc);
                mr = sor_finish_system<R, S>(acc, ldrec<S, 3>(L.data(), e));
                strec<S, 9>(Mr.data(), e, mr);
            } else mr = ldrec<S, 9>(Mr.data(), e);
            R m[9];
            for (int n = 0; n < 9; n++) m[n] = (R)mr.v[n];
            R du1, dv1, dw1;
            sor_relax<R>(m, ax, ay, az, (R)xm.v[0] + (R)xp.v[0], (R)xm.v[1] + (R)xp.v[1], (R)xm.v[2] + (R)xp.v[2],
                         (R)ym.v[0] + (R)yp.v[0], (R)ym.v[1] + (R)yp.v[1], (R)ym.v[2] + (R)yp.v[2], (R)zm.v[0] + (R)zp.v[0],
                         (R)zm.v[1] + (R)zp.v[1], (R)zm.v[2] + (R)zp.v[2], (R)q0.v[0], (R)q0.v[1], (R)q0.v[2], du1, dv1, dw1);
            Rec<S, 3> out;
            out.v[0] = Sto<S>::quant(du1); out.v[1] = Sto<S>::quant(dv1); out.v[2] = Sto<S>::quant(dw1);
            strec<S, 3>(dr.data(), e, out);
        }
    }

    // ---- emulation of the window kernel ----
    constexpr int W = WIN_WMAX;
    std::vector<S> Mw = M0, dw = d0;
    std::vector<S> E((size_t)(W - 1) * e3);
    memset(E.data(), 0, E.size() * sizeof(S));
    WinArgs<S> wa;
    memset(&wa, 0, sizeof(wa));
    wa.a.M = Mw.data();
    for (int c = 0; c < C; c++) { wa.a.A[c] = A.data() + (size_t)c * e12; wa.a.weight[c] = w.data() + (size_t)c * total; wa.a.a_data[c] = adc[c]; }
    wa.a.L = L.data();
    wa.a.d = dw.data();
    wa.a.sk = sk;
    wa.a.ax = ax; wa.a.ay = ay; wa.a.az = az;
    wa.a.C = C; wa.a.iterations = T; wa.a.update_lag = lag; wa.a.nvol = 1;
    wa.E = E.data();
    wa.strideE = (long long)e3;
    const WinTabPtr tb{pb.data(), cp.data()};
    const WinSchedHost sc = make_win_schedule(Z, Y, T, lag, W);
    Shadow sh[2 + W - 1];
    for (auto &s : sh) s.init((size_t)total);
    long long wg = 0, steps = 0;
    auto run_tile = [&](auto build_tag, const WinTile &tl, int win, int l, int n) {
        constexpr bool BUILD = decltype(build_tag)::value;
        using Th = WinThread<R, S, C, W, BUILD, WinTabPtr, EmuHook>;
        constexpr int NT = W * WIN_NL;
        static std::vector<Th> th(NT);
        static typename Th::Lds lds;
        EmuHook hk{sh, l, n};
        int s0, s1;
        Th::step_range(sk, tl, s0, s1);
        memset(&lds, 0xff, sizeof(lds));  // stale LDS content must not matter
        for (int tid = 0; tid < NT; tid++) th[tid].init(wa, tl, 0, tid, win, sc.windows[win].win_build);
        for (int s = s0 - WIN_LEAD; s <= s1; s++) {
            for (int tid = 0; tid < NT; tid++) th[tid].step(wa, tb, s, lds, hk);
            steps++;
        }
        wg++;
    };
    for (size_t l = 0; l < sc.first.size(); l++) {
        for (int n = 0; n < sc.count[l]; n++) {
            const WinTile &tl = sc.tiles[sc.first[l] + n];
            const int win = sc.win[sc.first[l] + n];
            const bool build = (tl.info >> 8) & 1;
            if (build != (n < sc.nbuild[l])) { printf("  schedule: build tiles must come first\n"); g_errors++; }
            if (build) run_tile(std::true_type{}, tl, win, (int)l, n);
            else run_tile(std::false_type{}, tl, win, (int)l, n);
        }
    }
    long long bad = 0;
    for (int k = 0; k < Z; k++) for (int j = 0; j < Y; j++) for (int i = 0; i < X; i++) {
        const long long e = sk_index(sk, k, j, i);
        const Rec<S, 3> r0 = ldrec<S, 3>(dr.data(), e), r1 = ldrec<S, 3>(dw.data(), e);
        for (int c = 0; c < 3; c++)
            if (memcmp(&r0.v[c], &r1.v[c], sizeof(V)) != 0) {
                if (bad++ < 8) printf("  mismatch (%d,%d,%d) c%d ref %.17g win %.17g\n", k, j, i, c, (double)r0.v[c], (double)r1.v[c]);
            }
    }
    printf("%dx%dx%d T=%d lag=%d C=%d: %zu launches %lld workgroups %lld wg-steps; mismatches %lld, hook errors %lld\n", Z, Y, X, T,
           lag, C, sc.first.size(), wg, steps, bad, g_errors);
    return bad || g_errors ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc < 6) { printf("usage: %s Z Y X iterations lag [mode] [C]\n", argv[0]); return 2; }
    const int Z = atoi(argv[1]), Y = atoi(argv[2]), X = atoi(argv[3]), T = atoi(argv[4]), lag = atoi(argv[5]);
    const int mode = argc > 6 ? atoi(argv[6]) : 2, C = argc > 7 ? atoi(argv[7]) : 1;
    if (C == 1) {
        if (mode == 0) return run<float, float, 1>(Z, Y, X, T, lag, 7);
        if (mode == 1) return run<double, float, 1>(Z, Y, X, T, lag, 7);
        if (mode == 2) return run<double, double, 1>(Z, Y, X, T, lag, 7);
        return run<double, pk42, 1>(Z, Y, X, T, lag, 7);
    }
    if (mode == 2) return run<double, double, 2>(Z, Y, X, T, lag, 7);
    return run<double, pk42, 2>(Z, Y, X, T, lag, 7);
}
