// k_sor_win_sched.h -- host side of the window sweep's schedule (no device calls): which (window, tile) pairs run
// in which launch.  Shared by k_sor_win.hip and the CPU emulator.
#pragma once

#include <algorithm>
#include <vector>

#include "k_sor_win_core.h"

namespace fr3d {

// One window = up to `wmax` consecutive iterations [t0, t0 + n).  A psi period (update_lag iterations, the first of
// which rebuilds the frozen system, level_solver_3d.py:356) is cut into windows of at most wmax iterations; only the
// first window of a period builds the system, the others read it from memory (so the builder stores it for every voxel).
struct WinWindow {
    int t0, n;
    bool build, storeM;
    int win_build;  // index of the window that built this window's system
};

static inline std::vector<WinWindow> make_windows(int iterations, int update_lag, int wmax)
{
    std::vector<WinWindow> w;
    for (int p0 = 0; p0 < iterations; p0 += update_lag) {
        const int pend = std::min(iterations, p0 + update_lag);
        const int first = (int)w.size();
        for (int t0 = p0; t0 < pend; t0 += wmax) {
            WinWindow x;
            x.t0 = t0;
            x.n = std::min(wmax, pend - t0);
            x.build = t0 == p0;
            x.storeM = x.build && pend - p0 > wmax;
            x.win_build = first;
            w.push_back(x);
        }
    }
    return w;
}

struct WinSchedHost {
    std::vector<WinTile> tiles;     // all launches back to back
    std::vector<int> first, count;  // per launch
    std::vector<int> nbuild;        // per launch: tiles of windows that build their system (they come first)
    std::vector<int> win;           // per tile: window index (emulator's version checks)
    std::vector<WinWindow> windows;
    int nwin = 0;
};

// Launch l runs the tiles (K,J) with K + J = l - WIN_DLAG * b of every window b for which that diagonal exists:
// a tile needs its lower neighbours of the same window (earlier diagonals) and, for the previous window's final
// values around it, the diagonals up to K + J + 2 of window b - 1.
static inline WinSchedHost make_win_schedule(int Z, int Y, int iterations, int update_lag, int wmax)
{
    WinSchedHost sc;
    const std::vector<WinWindow> ww = make_windows(iterations, update_lag, wmax);
    sc.nwin = (int)ww.size();
    sc.windows = ww;
    if (ww.empty()) return sc;
    auto ntile = [](int len, int n, int blk) { return (len + n - 1 + blk - 1) / blk; };  // lines up to len-1 + (n-1) skew
    int dmax = 0;
    for (const WinWindow &x : ww) dmax = std::max(dmax, ntile(Z, x.n, WIN_BK) + ntile(Y, x.n, WIN_BJ) - 2);
    const int nl = dmax + WIN_DLAG * (sc.nwin - 1) + 1;
    for (int l = 0; l < nl; l++) {
        sc.first.push_back((int)sc.tiles.size());
        for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) sc.nbuild.push_back((int)sc.tiles.size() - sc.first.back());
        for (int b = 0; b < sc.nwin; b++) {
            const int D = l - WIN_DLAG * b;
            if (D < 0) break;
            const WinWindow &x = ww[b];
            if (x.build != (pass == 0)) continue;
            const int Kn = ntile(Z, x.n, WIN_BK), Jn = ntile(Y, x.n, WIN_BJ);
            for (int K = std::max(0, D - (Jn - 1)); K <= std::min(D, Kn - 1); K++) {
                WinTile t;
                t.K = K;
                t.J = D - K;
                t.t0 = x.t0;
                t.info = x.n | ((int)x.build << 8) | ((int)x.storeM << 9);
                sc.tiles.push_back(t);
                sc.win.push_back(b);
            }
        }
        }
        sc.count.push_back((int)sc.tiles.size() - sc.first.back());
    }
    return sc;
}

}  // namespace fr3d
