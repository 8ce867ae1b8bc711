// k_sor_win.hip -- the window form of the a_smooth == 1 SOR sweep on gfx950: kernel, device schedule and launcher.
// The algorithm, the per-thread state and the step function are in k_sor_win_core.h (shared with the CPU emulator
// tools/emu/sor_win_emu.hip); reference: core/level_solver_3d.py:314-546.
//
// One workgroup = WIN_WMAX x WIN_NL threads = one tile of WIN_BK x WIN_BJ lines for one psi window: thread (q, line)
// runs slot q (iteration t0 + q) of its line and hands the increments and the frozen system on to slot q + 1 through
// LDS (double-buffered by step parity, one barrier per step); only slot 0 reads the operands from HBM, only the last
// slot writes the increments back.  HBM traffic per update falls to about a W-th of the plane sweep's; the kernel is
// bound by its instruction stream instead.
#include <cstdio>
#include <cstdlib>

#include "fr3d_internal.h"
#include "k_sor_win_sched.h"

namespace fr3d {

// row-start tables of the level in LDS (dynamic shared memory): pb (S + 2 entries of 8 bytes), then cp (X + Y + 2 ints)
extern __shared__ long long win_tables[];
struct WinTabLds {
    int npb;
    __device__ __forceinline__ long long pbv(int n) const { return win_tables[n]; }
    __device__ __forceinline__ int cpv(int n) const { return reinterpret_cast<const int *>(win_tables + npb)[n]; }
};
static size_t win_table_bytes(const Skew &sk) { return (size_t)(sk.S + 2) * 8 + (size_t)(sk.X + sk.Y + 2) * 4; }

template <typename R, typename S, int C, int W, bool BUILD>
__global__ void __launch_bounds__(WIN_WMAX * WIN_NL)
k_sor_win(const WinArgs<S> wa, const WinTile *__restrict__ tiles)
{
    using Th = WinThread<R, S, C, W, BUILD, WinTabLds>;
    __shared__ typename Th::Lds lds;
    const WinTile tl = tiles[blockIdx.x];
    const int tid = (int)threadIdx.x;
    const Skew &sk = wa.a.sk;
    WinTabLds tb;
    tb.npb = sk.S + 2;
    for (int n = tid; n < sk.S + 2; n += WIN_NT) win_tables[n] = sk.pb[n];
    int *cpl = reinterpret_cast<int *>(win_tables + tb.npb);
    for (int n = tid; n < sk.X + sk.Y + 2; n += WIN_NT) cpl[n] = sk.cp[n];
    Th th;
    th.init(wa, tl, (int)blockIdx.y, tid, 0, 0);
    int s0, s1;
    Th::step_range(sk, tl, s0, s1);
    const WinNoHook hk;
    __syncthreads();
    for (int s = s0 - WIN_LEAD; s <= s1; s++) {
        th.step(wa, tb, s, lds, hk);
        __syncthreads();
    }
}

WinSched build_win_schedule(const Skew &sk, int iterations, int update_lag)
{
    WinSched ws;
    const WinSchedHost h = make_win_schedule(sk.Z, sk.Y, iterations, update_lag, WIN_WMAX);
    ws.first = h.first;
    ws.count = h.count;
    ws.nbuild = h.nbuild;
    if (!h.tiles.empty()) {
        FR3D_HIP(hipMalloc((void **)&ws.tiles, h.tiles.size() * sizeof(WinTile)));
        FR3D_HIP(hipMemcpy(ws.tiles, h.tiles.data(), h.tiles.size() * sizeof(WinTile), hipMemcpyHostToDevice));
    }
    return ws;
}

void free_win_schedule(WinSched &ws)
{
    if (ws.tiles) (void)hipFree(ws.tiles);
    ws.tiles = nullptr;
}

bool sor_win_supports(int C) { return C >= 1 && C <= 2; }
template <typename S> bool sor_win_storage(int C) { return !std::is_same<S, double>::value || C == 1; }
template bool sor_win_storage<float>(int);
template bool sor_win_storage<double>(int);
template bool sor_win_storage<pk42>(int);
// the level's row-start tables must fit beside the exchange buffers in LDS
bool sor_win_fits(const Skew &sk) { return sk.pb != nullptr && win_table_bytes(sk) <= 24 * 1024; }

template <typename R, typename S, int C>
static void launch_win_step(hipStream_t st, const WinArgs<S> &wa, const WinTile *tiles, int count, bool build)
{
    if (count <= 0) return;
    {
        const dim3 grid(count, wa.a.nvol > 0 ? wa.a.nvol : 1), block(WIN_NT);
        if (build) hipLaunchKernelGGL((k_sor_win<R, S, C, WIN_WMAX, true>), grid, block, win_table_bytes(wa.a.sk), st, wa, tiles);
        else hipLaunchKernelGGL((k_sor_win<R, S, C, WIN_WMAX, false>), grid, block, win_table_bytes(wa.a.sk), st, wa, tiles);
        FR3D_LAUNCH_CHECK();
    }
}

template <typename S>
long long launch_sor_win(hipStream_t st, const WinArgs<S> &wa, bool fp64, const WinSched &ws)
{
    FR3D_CHECK(sor_win_supports(wa.a.C), "window sweep: 1 or 2 channels");
    FR3D_CHECK(sor_win_fits(wa.a.sk), "window sweep: level too large for the LDS tables (or pitched layout)");
    long long launches = 0;
    for (size_t l = 0; l < ws.first.size(); l++) {
        if (ws.count[l] <= 0) continue;
        // the tiles of a launch: first those of windows that build their system, then those that read it (two kernels)
        for (int part = 0; part < 2; part++) {
            const int n = part == 0 ? ws.nbuild[l] : ws.count[l] - ws.nbuild[l];
            if (n <= 0) continue;
            const WinTile *tiles = ws.tiles + ws.first[l] + (part == 0 ? 0 : ws.nbuild[l]);
            const bool r64 = Sto<S>::wide || fp64;
            if (wa.a.C == 1) {
                if (r64) launch_win_step<double, S, 1>(st, wa, tiles, n, part == 0);
                else if constexpr (!Sto<S>::wide) launch_win_step<float, S, 1>(st, wa, tiles, n, part == 0);
            } else if constexpr (!std::is_same<S, double>::value) {
                if (r64) launch_win_step<double, S, 2>(st, wa, tiles, n, part == 0);
                else if constexpr (!Sto<S>::wide) launch_win_step<float, S, 2>(st, wa, tiles, n, part == 0);
            } else {
                // not built: 165+ VGPRs for one channel already (the build role would spill), and several channels
                // default to the plane sweep with fp64 storage anyway
                throw Error("window sweep: two channels with fp64 solver storage are not built (LDS)");
            }
            launches++;
        }
    }
    return launches;
}
template long long launch_sor_win<float>(hipStream_t, const WinArgs<float> &, bool, const WinSched &);
template long long launch_sor_win<double>(hipStream_t, const WinArgs<double> &, bool, const WinSched &);
template long long launch_sor_win<pk42>(hipStream_t, const WinArgs<pk42> &, bool, const WinSched &);

}  // namespace fr3d
