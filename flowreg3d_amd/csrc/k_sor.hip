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
// Data sits in the skewed layout of fr3d_internal.h, so a wave reads/writes contiguous j-runs
// and the six neighbours are row-uniform offsets.  The algorithmic traffic of the reference's
// update is 9C tensor entries + C (w*psi) + 3 Laplacian terms + 3 increments read, 3 written:
// 4*(10C+9) bytes (76 B for C = 1) -- the figure bench.py prices the kernel against.  The kernel
// itself streams less: between psi updates the per-voxel 3x3 system (6+3 floats, channels summed)
// is frozen, so ordinary iterations read 9 + 3 floats and write 3; psi-update iterations (every
// update_lag-th) read the 12C square-root factors + C weights + 3 L + 3 d and write 9 + 3.
//
// Fusions: the Neumann ghost copy set_boundary_3d (:246-259) becomes "a missing neighbour is the
// voxel's own old value"; the psi_data update (:356-377) is pointwise in the old increment, so it
// is evaluated inside the sweep on iterations with t % update_lag == 0 and stored as w*psi;
// the u,v,w part of the stencil is iteration-invariant and precomputed once (k_laplace).
#include <algorithm>
#include <cstdlib>

#include "fr3d_internal.h"

namespace fr3d {

#define SOR_OMEGA 1.95
#define SOR_BX 64
#define SOR_BY_MAX 4  // rows of a tile = blockDim.y (1, 2 or 4; chosen per level, see sor_tile_rows)

template <typename R> __device__ __forceinline__ R fma_(R a, R b, R c);
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return fma(a, b, c); }

template <typename R, typename S, int C>
__global__ void __launch_bounds__(SOR_BX * SOR_BY_MAX)
k_sor_step(const SorArgsT<S> a, int tau, int t_lo, int nt, const SorEntry *__restrict__ ent)
{
    const int Z = a.sk.Z, Y = a.sk.Y, X = a.sk.X, Yp = a.sk.Yp;
    const long long plane = a.sk.plane;
    // blockIdx.x enumerates the tiles of all in-flight iterations (schedule built on the host):
    // find the iteration by bisection on the tile prefix, then the tile inside its bounding box
    const int vol = blockIdx.y;
    const int b = blockIdx.x;
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ent[mid].pre <= b) lo = mid;
        else hi = mid - 1;
    }
    const SorEntry en = ent[lo];
    const int local = b - en.pre;
    const int t = t_lo + lo;
    const int s = tau - 2 * t;
    const int k = (en.kb0 + local / en.njb) * (int)blockDim.y + threadIdx.y;
    if (k >= Z) return;
    const int r = s - k;                       // i + j of this row
    const int jm0 = sk_jm(X, r);               // first valid j of the row (left-aligned storage)
    const int jj = (local % en.njb) * SOR_BX + threadIdx.x;
    const int j = jj + jm0;
    const int i = r - j;
    if (r < 0 || j >= Y || i < 0) return;      // i < X holds by construction of jm0

    const size_t c0 = (size_t)((long long)s * plane + (long long)k * Yp + jj);
    const long long d1 = jm0 - sk_jm(X, r - 1), d2 = jm0 - sk_jm(X, r + 1);
    const size_t oM = c0 + (size_t)(vol * a.vsM), oA = c0 + (size_t)(vol * a.vsA), oL = c0 + (size_t)(vol * a.vsL);
    S *const dU = a.d[0] + vol * a.vsD, *const dV = a.d[1] + vol * a.vsD, *const dW = a.d[2] + vol * a.vsD;
    const R du0 = (R)dU[c0], dv0 = (R)dV[c0], dw0 = (R)dW[c0];

    // neighbour sums; a ghost neighbour holds the voxel's own old value (set_boundary_3d)
    R su_x, sv_x, sw_x, su_y, sv_y, sw_y, su_z, sv_z, sw_z;
    const bool nonb = a.dbg & 1;
    {
        const bool hm = i > 0 && !nonb, hp = i < X - 1 && !nonb;
        const size_t m = (size_t)((long long)c0 - plane + d1), p = (size_t)((long long)c0 + plane + d2);
        su_x = (hm ? (R)dU[m] : du0) + (hp ? (R)dU[p] : du0);
        sv_x = (hm ? (R)dV[m] : dv0) + (hp ? (R)dV[p] : dv0);
        sw_x = (hm ? (R)dW[m] : dw0) + (hp ? (R)dW[p] : dw0);
    }
    {
        const bool hm = j > 0 && !nonb, hp = j < Y - 1 && !nonb;
        const size_t m = (size_t)((long long)c0 - plane + d1 - 1), p = (size_t)((long long)c0 + plane + d2 + 1);
        su_y = (hm ? (R)dU[m] : du0) + (hp ? (R)dU[p] : du0);
        sv_y = (hm ? (R)dV[m] : dv0) + (hp ? (R)dV[p] : dv0);
        sw_y = (hm ? (R)dW[m] : dw0) + (hp ? (R)dW[p] : dw0);
    }
    {
        const bool hm = k > 0 && !nonb, hp = k < Z - 1 && !nonb;
        const size_t m = c0 - (size_t)plane - Yp, p = c0 + (size_t)plane + Yp;
        su_z = (hm ? (R)dU[m] : du0) + (hp ? (R)dU[p] : du0);
        sv_z = (hm ? (R)dV[m] : dv0) + (hp ? (R)dV[p] : dv0);
        sw_z = (hm ? (R)dW[m] : dw0) + (hp ? (R)dW[p] : dw0);
    }
    // System of this voxel for the current psi window: M = sum_c w_c psi_c J_c (6 entries of the
    // symmetric 3x3 block) and b = L - sum_c w_c psi_c (J14,J24,J34)_c.  psi is frozen between
    // psi-update iterations (level_solver_3d.py:356), so M and b are too: an update iteration
    // builds them from the factors and stores them, the other iterations just stream the 9 values
    // -- independent of the channel count.
    R M11, M22, M33, M12, M13, M23, b_u, b_v, b_w;
    const bool upd = (t % a.update_lag) == 0 && !(a.dbg & 4);
    if (upd) {
        M11 = M22 = M33 = M12 = M13 = M23 = (R)0;
        R bu = 0, bv = 0, bw = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            // psi_data update (level_solver_3d.py:356-377) from the increments of iteration t-1.
            // The quadratic form is evaluated as the sum of three squared residuals of the tensor's
            // square-root factors (see k_tensor.hip) -- algebraically the reference's expression,
            // but stable with fp32 storage.
            S f[12];
#pragma unroll
            for (int q = 0; q < 12; q++) f[q] = a.A[q * FR3D_MAX_CHANNELS + c][oA];
            double wt = (double)a.weight[c][c0];
            const double adc = a.a_data[c];
            if (adc != 1.0) {
                const double u_ = (double)du0, v_ = (double)dv0, w_ = (double)dw0;
                double val = 0.0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double r = fma((double)f[4 * k], u_, fma((double)f[4 * k + 1], v_,
                                   fma((double)f[4 * k + 2], w_, (double)f[4 * k + 3])));
                    val = fma(r, r, val);
                }
                // fp32 powf (~1 ulp): the products below are stored in fp32 anyway, and the fp64
                // pow's ~600-instruction dependent chain set a ~6 us latency floor on every launch
                if (sizeof(S) == 8) wt *= adc * pow(val + 1e-6, adc - 1.0);  // reference-grade mode
                else wt *= adc * (double)powf((float)(val + 1e-6), (float)(adc - 1.0));
            }
            const R w = (R)(S)wt;
            const R x0 = (R)f[0], x1 = (R)f[1], x2 = (R)f[2], x3 = (R)f[3];
            const R y0 = (R)f[4], y1 = (R)f[5], y2 = (R)f[6], y3 = (R)f[7];
            const R z0 = (R)f[8], z1 = (R)f[9], z2 = (R)f[10], z3 = (R)f[11];
            M11 = fma_<R>(w, fma_<R>(z0, z0, fma_<R>(y0, y0, x0 * x0)), M11);
            M22 = fma_<R>(w, fma_<R>(z1, z1, fma_<R>(y1, y1, x1 * x1)), M22);
            M33 = fma_<R>(w, fma_<R>(z2, z2, fma_<R>(y2, y2, x2 * x2)), M33);
            M12 = fma_<R>(w, fma_<R>(z0, z1, fma_<R>(y0, y1, x0 * x1)), M12);
            M13 = fma_<R>(w, fma_<R>(z0, z2, fma_<R>(y0, y2, x0 * x2)), M13);
            M23 = fma_<R>(w, fma_<R>(z1, z2, fma_<R>(y1, y2, x1 * x2)), M23);
            bu = fma_<R>(w, fma_<R>(z0, z3, fma_<R>(y0, y3, x0 * x3)), bu);
            bv = fma_<R>(w, fma_<R>(z1, z3, fma_<R>(y1, y3, x1 * x3)), bv);
            bw = fma_<R>(w, fma_<R>(z2, z3, fma_<R>(y2, y3, x2 * x3)), bw);
        }
        b_u = (R)a.L[0][oL] - bu;
        b_v = (R)a.L[1][oL] - bv;
        b_w = (R)a.L[2][oL] - bw;
        a.M[0][oM] = (S)M11; a.M[1][oM] = (S)M22; a.M[2][oM] = (S)M33;
        a.M[3][oM] = (S)M12; a.M[4][oM] = (S)M13; a.M[5][oM] = (S)M23;
        a.M[6][oM] = (S)b_u; a.M[7][oM] = (S)b_v; a.M[8][oM] = (S)b_w;
        // use the stored (rounded) values so update and non-update iterations see one system
        M11 = (R)(S)M11; M22 = (R)(S)M22; M33 = (R)(S)M33;
        M12 = (R)(S)M12; M13 = (R)(S)M13; M23 = (R)(S)M23;
        b_u = (R)(S)b_u; b_v = (R)(S)b_v; b_w = (R)(S)b_w;
    } else {
        M11 = (R)a.M[0][oM]; M22 = (R)a.M[1][oM]; M33 = (R)a.M[2][oM];
        M12 = (R)a.M[3][oM]; M13 = (R)a.M[4][oM]; M23 = (R)a.M[5][oM];
        b_u = (R)a.M[6][oM]; b_v = (R)a.M[7][oM]; b_w = (R)a.M[8][oM];
    }
    const R ax = (R)a.ax, ay = (R)a.ay, az = (R)a.az;
    const R num_u = fma_<R>(az, su_z, fma_<R>(ay, su_y, fma_<R>(ax, su_x, b_u)));
    const R num_v = fma_<R>(az, sv_z, fma_<R>(ay, sv_y, fma_<R>(ax, sv_x, b_v)));
    const R num_w = fma_<R>(az, sw_z, fma_<R>(ay, sw_y, fma_<R>(ax, sw_x, b_w)));
    const R diag = (R)(2.0 * a.ax + 2.0 * a.ay + 2.0 * a.az);
    const R den_u = diag + M11, den_v = diag + M22, den_w = diag + M33;

    const R om = (R)SOR_OMEGA, om1 = (R)(1.0 - SOR_OMEGA);
    // du (uses old dv, dw), dv (new du, old dw), dw (new du, dv): level_solver_3d.py:503-540
    R n2 = num_u - fma_<R>(M13, dw0, M12 * dv0);
    const R du1 = fma_<R>(om, (den_u != (R)0 ? n2 / den_u : (R)0), om1 * du0);
    n2 = num_v - fma_<R>(M23, dw0, M12 * du1);
    const R dv1 = fma_<R>(om, (den_v != (R)0 ? n2 / den_v : (R)0), om1 * dv0);
    n2 = num_w - fma_<R>(M23, dv1, M13 * du1);
    const R dw1 = fma_<R>(om, (den_w != (R)0 ? n2 / den_w : (R)0), om1 * dw0);

    dU[c0] = (S)du1;
    dV[c0] = (S)dv1;
    dW[c0] = (S)dw1;
}

template <typename R, typename S>
static void launch_step(hipStream_t st, const SorArgsT<S> &a, int tau, int t_lo, int nt, int ntiles,
                        const SorEntry *ent, int by)
{
    dim3 grid(ntiles, a.nvol > 0 ? a.nvol : 1), block(SOR_BX, by);
    switch (a.C) {
        case 1: hipLaunchKernelGGL((k_sor_step<R, S, 1>), grid, block, 0, st, a, tau, t_lo, nt, ent); break;
        case 2: hipLaunchKernelGGL((k_sor_step<R, S, 2>), grid, block, 0, st, a, tau, t_lo, nt, ent); break;
        case 3: hipLaunchKernelGGL((k_sor_step<R, S, 3>), grid, block, 0, st, a, tau, t_lo, nt, ent); break;
        case 4: hipLaunchKernelGGL((k_sor_step<R, S, 4>), grid, block, 0, st, a, tau, t_lo, nt, ent); break;
        default: throw Error("SOR kernel is instantiated for 1..4 channels");
    }
}

int sor_tile_rows(const Skew &sk)
{
    static const char *env = getenv("FR3D_SOR_BY");
    if (env) {
        const int v = atoi(env);
        if (v == 1 || v == 2 || v == 4) return v;
    }
    return (sk.X >= 320 && sk.Y >= 320) ? 2 : 4;
}

SorSched build_sor_schedule(const Skew &sk, int T, int by)
{
    SorSched sc;
    sc.by = by;
    const int S = sk.S, Z = sk.Z, Y = sk.Y, X = sk.X;
    std::vector<SorEntry> ent;
    if (T <= 0) return sc;
    const int last = (S - 1) + 2 * (T - 1);
    for (int tau = 0; tau <= last; tau++) {
        int t_lo = tau - (S - 1);
        t_lo = t_lo <= 0 ? 0 : (t_lo + 1) / 2;
        int t_hi = tau / 2;
        if (t_hi > T - 1) t_hi = T - 1;
        if (t_lo > t_hi) continue;
        sc.tau.push_back(tau);
        sc.t_lo.push_back(t_lo);
        sc.nt.push_back(t_hi - t_lo + 1);
        sc.first.push_back((int)ent.size());
        int pre = 0;
        for (int t = t_lo; t <= t_hi; t++) {
            const int s = tau - 2 * t;
            // valid rows of hyperplane s: k in [klo,khi]; row (s,k) holds r = s-k, j in
            // [jm(r), min(Y-1,r)], stored left-aligned, so tiles start at 0 and the row count of
            // j-tiles is set by the longest row
            const int klo = std::max(0, s - (X - 1) - (Y - 1)), khi = std::min(Z - 1, s);
            SorEntry e;
            e.pre = pre;
            e.pad0 = e.pad1 = 0;
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
                    e.kb0 = (short)kb0;
                    e.njb = (short)cdiv(maxlen, SOR_BX);
                    pre += (kb1 - kb0 + 1) * e.njb;
                }
            }
            ent.push_back(e);
        }
        sc.ntiles.push_back(pre);
    }
    FR3D_HIP(hipMalloc((void **)&sc.entries, ent.size() * sizeof(SorEntry)));
    FR3D_HIP(hipMemcpy(sc.entries, ent.data(), ent.size() * sizeof(SorEntry), hipMemcpyHostToDevice));
    return sc;
}

void free_sor_schedule(SorSched &s)
{
    if (s.entries) (void)hipFree(s.entries);
    s.entries = nullptr;
}

template <typename S>
long long launch_sor(hipStream_t st, const SorArgsT<S> &a_in, bool fp64, const SorSched &sc)
{
    SorArgsT<S> a = a_in;
    static const char *dbg_env = getenv("FR3D_SOR_DBG");
    a.dbg = dbg_env ? atoi(dbg_env) : 0;
    long long launches = 0;
    for (size_t l = 0; l < sc.tau.size(); l++) {
        if (sc.ntiles[l] <= 0) continue;
        const SorEntry *ent = sc.entries + sc.first[l];
        if (fp64 || sizeof(S) == 8) launch_step<double, S>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, sc.by);
        else launch_step<float, S>(st, a, sc.tau[l], sc.t_lo[l], sc.nt[l], sc.ntiles[l], ent, sc.by);
        launches++;
    }
    return launches;
}
template long long launch_sor<float>(hipStream_t, const SorArgsT<float> &, bool, const SorSched &);
template long long launch_sor<double>(hipStream_t, const SorArgsT<double> &, bool, const SorSched &);

// ---- layout conversion and the iteration-invariant stencil part -------------------------------

// Natural (Z,Y,X) <-> skewed layout for `narr` arrays at once, through a 64x64 LDS tile of one
// z-slice: the natural side moves as 256-B row segments (x contiguous), the skewed side as runs
// along the tile's anti-diagonals (x+y constant => same hyperplane and row, j contiguous), so both
// sides are coalesced.  A direct scatter costs ~8x write amplification (4-B writes, 128-B lines).
#define SKT 64
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
k_skew_tiled(const TS *__restrict__ src, long long src_stride, TD *__restrict__ dst,
             long long dst_stride, int Z, int Y, int X, int Yp, long long plane, int to_skew)
{
    __shared__ TD tile[SKT][SKT + 2];
    const int txn = (X + SKT - 1) / SKT;
    const int x0 = (blockIdx.x % txn) * SKT, y0 = (blockIdx.x / txn) * SKT;
    const int z = blockIdx.y;
    src += (size_t)blockIdx.z * src_stride;
    dst += (size_t)blockIdx.z * dst_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (to_skew) {
        for (int row = wave; row < SKT; row += 4) {
            const int y = y0 + row, x = x0 + lane;
            if (y < Y && x < X) tile[row][lane] = (TD)src[((size_t)z * Y + y) * X + x];
        }
        __syncthreads();
        for (int d = wave; d < 2 * SKT - 1; d += 4) {
            const int ly = lane, lx = d - lane;
            const int y = y0 + ly, x = x0 + lx;
            if (lx >= 0 && lx < SKT && y < Y && x < X) dst[(size_t)sk_index(X, Yp, plane, z, y, x)] = tile[ly][lx];
        }
    } else {
        for (int d = wave; d < 2 * SKT - 1; d += 4) {
            const int ly = lane, lx = d - lane;
            const int y = y0 + ly, x = x0 + lx;
            if (lx >= 0 && lx < SKT && y < Y && x < X) tile[ly][lx] = (TD)src[(size_t)sk_index(X, Yp, plane, z, y, x)];
        }
        __syncthreads();
        for (int row = wave; row < SKT; row += 4) {
            const int y = y0 + row, x = x0 + lane;
            if (y < Y && x < X) dst[((size_t)z * Y + y) * X + x] = tile[row][lane];
        }
    }
}

template <typename TS, typename TD>
static void launch_skew_tiled(hipStream_t st, const TS *src, long long src_stride, TD *dst,
                              long long dst_stride, int narr, const Skew &sk, int to_skew)
{
    if (narr <= 0) return;
    dim3 grid(cdiv(sk.X, SKT) * cdiv(sk.Y, SKT), sk.Z, narr);
    hipLaunchKernelGGL((k_skew_tiled<TS, TD>), grid, dim3(256), 0, st, src, src_stride, dst, dst_stride, sk.Z,
                       sk.Y, sk.X, sk.Yp, sk.plane, to_skew);
}

template <typename TS, typename TD>
void launch_skew_copy_n(hipStream_t st, const TS *src, long long src_stride, TD *dst, long long dst_stride,
                        int narr, const Skew &sk)
{
    launch_skew_tiled<TS, TD>(st, src, src_stride, dst, dst_stride, narr, sk, 1);
}

template <typename TS, typename TD>
void launch_unskew_copy_n(hipStream_t st, const TS *src, long long src_stride, TD *dst, long long dst_stride,
                          int narr, const Skew &sk)
{
    launch_skew_tiled<TS, TD>(st, src, src_stride, dst, dst_stride, narr, sk, 0);
}
template void launch_skew_copy_n<float, float>(hipStream_t, const float *, long long, float *, long long, int, const Skew &);
template void launch_skew_copy_n<float, double>(hipStream_t, const float *, long long, double *, long long, int, const Skew &);
template void launch_skew_copy_n<double, double>(hipStream_t, const double *, long long, double *, long long, int, const Skew &);
template void launch_unskew_copy_n<float, float>(hipStream_t, const float *, long long, float *, long long, int, const Skew &);
template void launch_unskew_copy_n<double, float>(hipStream_t, const double *, long long, float *, long long, int, const Skew &);

// L = ax*(u_ip + u_im - 2u) + ay*(...) + az*(...) with edge-padded u (add_boundary,
// core/optical_flow_3d.py:88), evaluated in fp64 from the fp32-exact level flow.
template <typename TL>
__global__ void __launch_bounds__(256)
k_laplace(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ w,
          int Z, int Y, int X, int Yp, long long plane, double ax, double ay, double az,
          TL *__restrict__ Lu, TL *__restrict__ Lv, TL *__restrict__ Lw)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)Z * Y * X;
    if (t >= total) return;
    int x = (int)(t % X);
    long long r = t / X;
    int y = (int)(r % Y);
    int z = (int)(r / Y);
    const long long sx = 1, sy = X, sz = (long long)Y * X;
    const long long xm = x > 0 ? -sx : 0, xp = x < X - 1 ? sx : 0;
    const long long ym = y > 0 ? -sy : 0, yp = y < Y - 1 ? sy : 0;
    const long long zm = z > 0 ? -sz : 0, zp = z < Z - 1 ? sz : 0;
    size_t o = plane ? (size_t)sk_index(X, Yp, plane, z, y, x) : (size_t)t;
    const float *f[3] = {u, v, w};
    TL *L[3] = {Lu, Lv, Lw};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float *q = f[d] + t;
        double c = (double)q[0];
        double acc = ax * ((double)q[xp] + (double)q[xm] - 2.0 * c);
        acc += ay * ((double)q[yp] + (double)q[ym] - 2.0 * c);
        acc += az * ((double)q[zp] + (double)q[zm] - 2.0 * c);
        L[d][o] = (TL)acc;
    }
}

template <typename TL>
void launch_laplace(hipStream_t st, const float *u, const float *v, const float *w, const Skew &sk,
                    double ax, double ay, double az, TL *Lu, TL *Lv, TL *Lw, bool natural)
{
    long long total = (long long)sk.Z * sk.Y * sk.X;
    hipLaunchKernelGGL(k_laplace<TL>, dim3(cdiv(total, 256)), dim3(256), 0, st, u, v, w, sk.Z, sk.Y,
                       sk.X, sk.Yp, natural ? 0LL : sk.plane, ax, ay, az, Lu, Lv, Lw);
}

template void launch_laplace<float>(hipStream_t, const float *, const float *, const float *, const Skew &, double,
                                    double, double, float *, float *, float *, bool);
template void launch_laplace<double>(hipStream_t, const float *, const float *, const float *, const Skew &, double,
                                     double, double, double *, double *, double *, bool);

}  // namespace fr3d
