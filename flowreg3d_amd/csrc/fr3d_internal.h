// fr3d_internal.h -- shared declarations of the gfx950 engine (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/flowreg3d_hip.h"

namespace fr3d {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define FR3D_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            throw ::fr3d::Error(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" +      \
                                __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
    } while (0)

#define FR3D_CHECK(cond, msg)                                                                   \
    do {                                                                                        \
        if (!(cond)) throw ::fr3d::Error(std::string(msg));                                     \
    } while (0)

// after a kernel launch: a rejected launch (bad grid, too much LDS, ...) must not return rc = 0
#define FR3D_LAUNCH_CHECK() FR3D_HIP(hipGetLastError())

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Skewed ("hyperplane-major") layout used by the SOR sweep.
//
// Interior voxel (k,j,i) = (z,y,x) of a level lives at  s*plane + k*Yp + (j - jm(s-k))  with
// s = i+j+k and jm(r) = max(0, r-(X-1)) the first valid j of row (s,k): rows are LEFT-ALIGNED, so
// the valid voxels of a row start at a 256-B boundary (one partly used cache line per row instead
// of two, and only the last 64-lane tile of a row is partly filled).  All voxels of one
// lexicographic-Gauss-Seidel wavefront (constant s) are contiguous in j for fixed k, and the six
// stencil neighbours are at row-uniform offsets (d1 = jm(r)-jm(r-1) in {0,1}, d2 = jm(r)-jm(r+1)
// in {0,-1}, r = s-k):
//   (k,j,i-1) -> -plane+d1   (k,j-1,i) -> -plane+d1-1  (k-1,j,i) -> -plane-Yp
//   (k,j,i+1) -> +plane+d2   (k,j+1,i) -> +plane+d2+1  (k+1,j,i) -> +plane+Yp
// Storage is S*Z*Yp elements (S = X+Y+Z-2 hyperplanes): ~3x the voxel count for a cube; only the
// valid third is ever touched, so it costs HBM capacity (288 GB), not bandwidth.
// ---------------------------------------------------------------------------------------------
struct Skew {
    int Z, Y, X;
    int Yp;           // row pitch (Y rounded up to 64 elements)
    int S;            // number of hyperplanes
    long long plane;  // Z*Yp
    long long total;  // elements of one array: S*plane, or the packed size when pb/cp are set
    // COMPACT variant (the a_smooth == 1 sweep and its operands): the rows of a hyperplane are packed back
    // to back, each padded to a multiple of 64 elements (256-B aligned starts, no unused tail of up to
    // Yp - len elements per row): 1.1-1.3x the voxel count instead of 3x.  Row (s,k) starts at
    //     pb[s + 1] - cp[s - k + 1]
    // with cp[n] = padded lengths of the rows r' = i+j < n summed (a row's length depends on r' only) and
    // pb[s + 1] = start of plane s + cp[s - klo(s) + 1]; both tables are tiny (S + 2 and X + Y + 2 entries) and
    // stay in the scalar / vector caches.  nullptr: the pitched layout above.
    const long long *pb;
    const int *cp;
};

__host__ __device__ static inline int sk_jm(int X, int r) { return r > X - 1 ? r - (X - 1) : 0; }
// storage index of lane 0 of row (s,k); valid rows only (0 <= k < Z, 0 <= s - k <= X + Y - 2)
__host__ __device__ static inline long long sk_row(const Skew &sk, int s, int k)
{
    return sk.pb ? sk.pb[s + 1] - (long long)sk.cp[s - k + 1] : (long long)s * sk.plane + (long long)k * sk.Yp;
}
// storage index of interior voxel (z,y,x)
__host__ __device__ static inline long long sk_index(const Skew &sk, int z, int y, int x)
{
    return sk_row(sk, x + y + z, z) + (y - sk_jm(sk.X, x + y));
}
// pitched layout only (kernels that receive the geometry as scalars)
__host__ __device__ static inline long long sk_index(int X, int Yp, long long plane, int z, int y, int x)
{
    return (long long)(x + y + z) * plane + (long long)z * Yp + (y - sk_jm(X, x + y));
}

// a / b, correctly rounded, from y = RN(1/b) computed by the host's IEEE division: q0 = RN(a*y) is within 2 ulp
// of a/b, one residual step makes it faithful, and a faithful quotient corrected once more with the exact FMA
// residual is the correctly rounded one (Markstein's theorem; no overflow/underflow for image and flow values
// over grid spacings of order 1).  5 instructions instead of the ~12 of a full fp64 division.  Differs from a
// true division only for infinite a (NaN instead of inf) and in the sign of a zero quotient.
__device__ __forceinline__ double div_by_const(double a, double b, double y)
{
    const double q0 = a * y;
    const double q1 = fma(fma(-b, q0, a), y, q0);
    return fma(fma(-b, q1, a), y, q1);
}

// Anti-diagonal walk of a TY x 32 tile of one z-slice (TY = 32 or 16) by a group of 32 lanes: in step m = 0 .. TY-1
// lane l handles element (ly, lx).  Lanes with consecutive ly on one diagonal x + y = const are consecutive voxels
// of a skewed row.  A short diagonal d and its partner d + 32 fill the 32 lanes together (d + 1 and 31 - d
// elements): every step uses all lanes, where one diagonal per step leaves half of them idle on average.
template <int TY>
__device__ __forceinline__ void tile_diag(int lane, int m, int &ly, int &lx)
{
    static_assert(TY == 32 || TY == 16, "tile_diag: 32 or 16 rows");
    if constexpr (TY == 32) {
        ly = lane;
        lx = (m - lane) & 31;
    } else {
        ly = lane & 15;
        lx = (m + (lane & 16) - ly) & 31;
    }
}

static inline Skew make_skew(int Z, int Y, int X)
{
    Skew k;
    k.Z = Z; k.Y = Y; k.X = X;
    k.Yp = ((Y + 63) / 64) * 64;
    k.S = X + Y + Z - 2;
    k.plane = (long long)Z * k.Yp;
    k.total = (long long)k.S * k.plane;
    k.pb = nullptr;
    k.cp = nullptr;
    return k;
}

// Host tables of the compact variant: pb (S + 2 entries), cp (X + Y + 2 entries); returns the packed size.
static inline long long make_compact_tables(int Z, int Y, int X, std::vector<long long> &pb, std::vector<int> &cp)
{
    const int S = X + Y + Z - 2, R = X + Y - 1;  // rows have r = i + j in [0, R)
    cp.assign((size_t)X + Y + 2, 0);
    for (int n = 1; n < X + Y + 2; n++) {
        const int r = n - 1;
        int len = 0;
        if (r < R) len = std::min(Y - 1, r) - sk_jm(X, r) + 1;
        cp[n] = cp[n - 1] + ((len + 63) / 64) * 64;
    }
    pb.assign((size_t)S + 2, 0);
    long long start = 0;
    for (int s = 0; s < S; s++) {
        const int klo = std::max(0, s - (R - 1)), khi = std::min(Z - 1, s);
        pb[s + 1] = start + cp[s - klo + 1];
        start += (long long)cp[s - klo + 1] - (long long)cp[s - khi];
    }
    pb[S + 1] = start;
    return start;
}

// Solver operands of the a_smooth == 1 sweep (k_sor.hip).  S is the storage type of everything the sweep
// streams: float (default) or double (fr3d_params.solver_fp64 == 2, for configurations where the reference
// iteration itself is ill-conditioned -- see DESIGN.md section 2).
//
// RECORD layout: the values of one voxel that are always used together sit next to each other (9 system
// entries, 12 factors per channel, 3 Laplacian terms, 3 increments), records follow the skewed voxel order
// above.  A wave then fetches a voxel's operands with 10 wide loads instead of 30 dword loads from 30
// separate arrays, and a partly filled wave (rows of a hyperplane have every length from 1 to min(X,Y))
// still moves long contiguous pieces: tools/microbench/row_layout.hip measures +18...+38 % useful bandwidth
// for rows of 40...200 voxels against one array per operand.
// packed 42-bit storage element (k_sor_core.h: Sto<pk42>): one dword; three values share four dwords
struct pk42 {
    uint32_t w;
};
// storage type of the per-voxel channel weights (resampled fp32 values: float is exact)
template <typename S> struct StoWt { using type = S; };
template <> struct StoWt<pk42> { using type = float; };
// storage elements that hold `nvals` values (records of the sweep are multiples of 3 values)
template <typename S> constexpr long long sto_elems(long long nvals) { return nvals; }
template <> constexpr long long sto_elems<pk42>(long long nvals) { return nvals / 3 * 4; }
template <typename S> constexpr double sto_bytes_per_value() { return (double)sizeof(S); }
template <> constexpr double sto_bytes_per_value<pk42>() { return 16.0 / 3.0; }

template <typename S>
struct SorArgsT {
    // frozen per-voxel system of the current psi window, 9 per voxel: M11,M22,M33,M12,M13,M23,b_u,b_v,b_w
    // (written on psi-update iterations, read on the others; channels already summed)
    S *M;
    // square-root factors of the motion tensor per channel, 12 per voxel, index 4*k + a (k = x,y,z equation;
    // a = u,v,w,t column): J = sum_k a_k a_k^T.  Read on psi-update iterations.
    const S *A[FR3D_MAX_CHANNELS];
    const typename StoWt<S>::type *weight[FR3D_MAX_CHANNELS];  // one value per voxel (plain skewed arrays, shared by a batch)
    const S *L;  // alpha-weighted Laplacian of u,v,w (constant over the iterations), 3 per voxel
    S *d;        // du,dv,dw, 3 per voxel, updated in place
    Skew sk;
    double ax, ay, az;  // alpha/h^2
    double a_data[FR3D_MAX_CHANNELS];
    int C;
    int iterations, update_lag;
    // batch of volumes solved in lock step by the same launches: pointers above are volume 0,
    // volume v adds v*stride storage elements (sto_elems; weights are shared by all volumes of a batch)
    int nvol;
    long long vsM, vsA, vsL, vsD;
    // numerics experiments (builds with -DFR3D_EXPERIMENTS only; FR3D_SOR_DBG, meaningful with S = double): round
    // to fp32 when stored / read -- 1 increments, 2 frozen system, 4 Laplacian terms, 8 factors
    // (profiles/r02/numerics_512_rounding_groups.md)
    int dbg;
};
using SorArgs = SorArgsT<float>;

// ---- verification mode (k_verify.hip): the reference's arithmetic on the compact skewed layout, fp64 ----
struct VerifyArgs {
    Skew sk;
    const double *J[FR3D_MAX_CHANNELS];  // records of 10 per voxel: J11,J22,J33,J44,J12,J13,J23,J14,J24,J34
    const float *w[FR3D_MAX_CHANNELS];   // channel weight per voxel (resampled fp32 values)
    double *psi[FR3D_MAX_CHANNELS];      // psi_data per voxel, written on update iterations
    const double *U;                     // u,v,w of the level, records of 3
    double *D;                           // du,dv,dw, records of 3, updated in place
    double ax, ay, az;                   // alpha / h^2
    double a_data[FR3D_MAX_CHANNELS];
    int C, update_lag;
    // a_smooth != 1 (level_solver_3d.py:262-311, 400-471): psi_smooth of the running iteration on the PADDED natural grid
    // (Z+2, Y+2, X+2); nullptr: the constant-diffusion stencil.  The iterations then run one at a time (psi_smooth of
    // iteration t is a function of the whole field of t-1), `t_base` = the iteration the launches belong to.
    const double *Ps;
    int t_base;
    // a caller-chosen ghost ring of u,v,w (level_solver called directly with arrays that are not edge-padded): the
    // PADDED natural arrays (3, Z+2, Y+2, X+2); nullptr: the ghost ring is the edge pad of the interior (add_boundary)
    const double *Ug;
};
struct SorChainSched;
long long launch_sor_verify(hipStream_t st, const VerifyArgs &a, const SorChainSched &sc);
// psi_smooth of one iteration in the reference's arithmetic: D = increments as they stand (iteration t-1), Dm2 = those
// of iteration t-2 (the padded arrays' ghost ring, set_boundary_3d of the previous iteration), all records of 3 on `sk`
void launch_psi_smooth_verify(hipStream_t st, const Skew &sk, const double *U, const double *D, const double *Dm2, double a_smooth,
                              double hx, double hy, double hz, double *Ps, const double *Ug = nullptr);
void launch_median5_f64(hipStream_t st, const double *in, int Z, int Y, int X, double *out);
template <typename TS, typename TD>
void launch_cast(hipStream_t st, const TS *src, long long n, TD *dst);
void launch_axpy_f64(hipStream_t st, double *y, const double *x, long long n);
void launch_ppow(hipStream_t st, const double *x, const double *y, long long n, double *out);
void launch_pack3_f64(hipStream_t st, const double *a, const double *b, const double *c, long long n, double *out);

// ---- a_smooth != 1 solver path (k_sor_smooth.hip) ------------------------------------------------
#define SM_LAG 4  // hyperplanes between consecutive in-flight iterations on this path
template <typename S>
struct SmoothView {
    const S *U;    // u,v,w skewed (interior), records of 3
    const S *Dm1;  // increments of iteration t-1, records of 3
    const S *Dm2;  // increments of iteration t-2 (ghost source)
    int Z, Y, X;
    Skew sk;       // storage of the skewed rows (compact or pitched)
    double tx, ty, tz, rtx, rty, rtz;  // 2h per axis and RN(1 / 2h) (div_by_const)
    double a_smooth;
};

// Every per-volume operand of a lock-step batch is `vs*` elements behind the previous
// volume's (the kernel's blockIdx.y is the volume of the batch); `weight` is shared.
template <typename S>
struct SmoothArgs {
    SmoothView<S> view;      // U + geometry; Dm1/Dm2 are set per launch from `D`
    S *D[3];                 // increments, records of 3; iteration t writes D[t % 3]
    S *Ps;                   // psi_s of the iteration that will sweep next (in place per plane)
    S *M;                    // frozen data-term system, records of 9 (M11,M22,M33,M12,M13,M23, b_u,b_v,b_w; b without L)
    const S *A[FR3D_MAX_CHANNELS];  // square-root factors, records of 12 per channel
    const S *weight[FR3D_MAX_CHANNELS];
    long long vsU, vsD, vsP, vsM, vsA;
    double ax, ay, az;
    double a_data[FR3D_MAX_CHANNELS];
    int C, iterations, update_lag, S_planes, nvol;
    int dbg;                 // experiment build only (FR3D_SM_DBG): kinds of workgroups left out of a step, for timing
};
template <typename S>
inline void smooth_set_spacing(SmoothView<S> &v, double hx, double hy, double hz)
{
    v.tx = 2.0 * hx; v.ty = 2.0 * hy; v.tz = 2.0 * hz;
    v.rtx = 1.0 / v.tx; v.rty = 1.0 / v.ty; v.rtz = 1.0 / v.tz;
}

struct SorSched;
struct SorChainSched;
template <typename S>
long long launch_sor_smooth(hipStream_t st, const SmoothArgs<S> &a, const SorSched &sched);
// experiment build: P-stage and sweep tiles sharing the plane between them (chain schedule of 2 x iterations, shape 4 x 2)
template <typename S>
long long launch_sor_smooth_fused(hipStream_t st, const SmoothArgs<S> &a, const SorSched &sched, const SorChainSched &chain,
                                  bool paired);

// ---- launchers (each enqueues on `st`, no synchronisation) ------------------------------------

// K1 resample: one separable pass.  src element (a,b,c) at ((a*n1+b)*n2+c)*cs+co ; dst planar.
// axis 2 = x (innermost), 1 = y, 0 = z.
void launch_resize_pass(hipStream_t st, const float *src, int cs, int co, int n0, int n1, int n2,
                        int axis, int out_len, const int *idx, const float *wt, int P, float *dst);

// K2 warp
template <typename T>
void launch_pad_edge(hipStream_t st, const T *src, int cs, int co, int Z, int Y, int X, int npad,
                     double *dst);
void launch_prefilter3(hipStream_t st, double *c, int PZ, int PY, int PX);
// pad-free form (k_warp.hip, 2b): coefficients -2 .. N+1 per axis, coef (Z+4,Y+4,X+4), tmp (Z+4)(Y+4)X doubles
bool prefilter_compact_ok(int Z, int Y, int X);
template <typename T>
void launch_prefilter3_compact(hipStream_t st, const T *vol, int cs, int co, int Z, int Y, int X, double *coef, double *tmp);
// flow components pu/pv/pw with element stride fs; displacement = value / h (per axis).  TO = float
// inside the pyramid; the executor tail writes the raw volume's own element type (OutCast in k_warp.hip)
template <typename TF, typename TR, typename TO>
void launch_warp_cubic(hipStream_t st, const double *coef, int npad, const TF *pu, const TF *pv,
                       const TF *pw, int fs, double hx, double hy, double hz, const TR *ref,
                       int rcs, int rco, int Z, int Y, int X, TO *out, int ocs, int oco);
template <typename TV, typename TF, typename TR, typename TO>
void launch_warp_linear(hipStream_t st, const TV *vol, int vcs, int vco, const TF *pu,
                        const TF *pv, const TF *pw, int fs, const TR *ref, int Z, int Y, int X,
                        TO *out, int ocs, int oco);

// K3 motion tensor: f1,f2 planar (Z,Y,X) fp32.  Jout[a] for a = J11,J22,J33,J44,J12,J13,J23,
// J14,J24,J34; A (nullable): 12 factor arrays a_stride apart; written skewed (sk != nullptr) or
// natural.
template <typename TA, typename TJ = float>
void launch_motion_tensor(hipStream_t st, const float *f1, const float *f2, int Z, int Y, int X,
                          double hz, double hy, double hx, TJ *const Jout[10], TA *A,
                          long long a_stride, const Skew *sk);

// K3 straight into the solver's record layout: dst = 12 factor values per voxel in the (compact or pitched)
// skewed voxel order of `sk`
template <typename TA>
void launch_motion_tensor_rec(hipStream_t st, const float *f1, const float *f2, double hz, double hy, double hx,
                              TA *dst, const Skew &sk);

// K4-K7 SOR
// narr arrays, src_stride / dst_stride elements apart; element types may differ (converted)
// nrec (1, 3 or 12) natural planar arrays, src_stride elements apart -> one skewed array of nrec-value records
// (pitched or compact layout, whatever `sk` describes)
template <typename TS, typename TD>
void launch_skew_pack(hipStream_t st, const TS *src, long long src_stride, TD *dst, int nrec, const Skew &sk);
// skewed records of nrec (3) values -> nrec natural planar arrays, dst_stride elements apart
template <typename TS, typename TD>
void launch_unskew_unpack(hipStream_t st, const TS *src, TD *dst, long long dst_stride, int nrec, const Skew &sk);
// the same three terms as one record per voxel in the skewed order of `sk`
template <typename TL>
void launch_laplace_rec(hipStream_t st, const float *u, const float *v, const float *w, const Skew &sk, double ax,
                        double ay, double az, TL *dst);
// Launch schedule of one level geometry: for every launch tau and every in-flight iteration t the
// bounding box (in tile units) of the valid part of hyperplane s = tau - 2t, so that only tiles that
// can hold voxels are dispatched (an all-covering grid spends ~8 us per launch on empty workgroups).
// 16 bytes of ints: the kernels fetch an entry with ONE scalar load (s_load_dwordx4).  16-bit fields behind the
// first 8 bytes made the compiler fetch them with vector loads -- two extra memory round trips at the head of every
// wave, 14 % of the sweep's rate.
struct alignas(16) SorEntry {
    int pre;  // tiles of this launch before this group
    int kb0;  // first k-tile
    int njb;  // j-tiles per k-tile row (rows are left-aligned)
    int tn;   // first iteration of the group relative to the launch's t_lo (low 16 bits) | iterations in the group << 16
};
__host__ __device__ static inline int sor_entry_toff(const SorEntry &e) { return e.tn & 0xffff; }
__host__ __device__ static inline int sor_entry_nit(const SorEntry &e) { return e.tn >> 16; }
#define SOR_LUT_SHIFT 6
struct SorSched {
    std::vector<int> tau, t_lo, nt, first, ntiles;  // per launch
    SorEntry *entries = nullptr;                     // device, sum(nt) entries
    // per launch, for every group of SOR_LUT_GROUP consecutive tiles the iteration (entry index) that
    // holds the group's first tile: the kernel starts its search there instead of bisecting
    std::vector<int> lut_first;                      // per launch: offset into `lut`
    int *lut = nullptr;                              // device
    int by = 4;                                      // tile rows the schedule was built for
    int lag = 2;                                     // hyperplanes between consecutive in-flight iterations
    std::vector<int> launch_of_tau;                  // tau -> index into the per-launch vectors, -1 = nothing to do
    // lag-4 schedules only: the surface voxels of every hyperplane (k << 16 | j, sorted), so that the
    // a_smooth != 1 kernels can run them in dense workgroups of their own (their ghost handling is
    // expensive and would otherwise diverge in every row's first and last wave)
    std::vector<int> bnd_first, bnd_count;           // per hyperplane s (host copies)
    int *bnd_kj = nullptr;                           // device, all planes back to back
    int *bnd_meta = nullptr;                         // device, (first, count) per hyperplane
    int bnd_max = 0;                                 // largest bnd_count
};
// Schedule of the a_smooth == 1 sweep (k_sor.hip).  An entry is a CHAIN of up to `nch` consecutive iterations
// t0 .. t0+n-1 (ordinary and psi-update iterations alike; a wave decides which it is): one workgroup (64 lanes x `by` rows x `nch`
// chain positions) takes the tile (rows k.., lanes jj..) of iteration t0 on plane s0 = tau - 2 t0 and, at chain
// position n, the rows k-n.. of iteration t0+n on plane s0 - 2n.  Plane s0-2n-1 is the "minus" neighbour plane of
// position n and the "plus" neighbour plane of position n+1, the same rows and (to within one lane) the same
// lanes of it: a chain of n iterations fetches n+1 neighbour planes instead of 2n, all of them written by the
// PREVIOUS launch, so there is nothing to synchronise -- the sharing happens in the CU's L1 and the XCD's L2.
struct SorChainSched {
    std::vector<int> tau, t_lo, first, nent, ntiles, lut_first;  // per launch
    SorEntry *entries = nullptr;  // device
    int *lut = nullptr;           // device
    int by = 2, nch = 1;
};
SorChainSched build_sor_chain_schedule(const Skew &sk, int iterations, int by, int nch);
void free_sor_chain_schedule(SorChainSched &s);
// host replay of the kernel's index arithmetic: 0 = every update issued exactly once in the right launch
long long check_chain_schedule(int Z, int Y, int X, int iterations, int by, int nch, long long *n_updates);
// workgroup shape of the sweep: rows per tile and chain positions
void sor_tile_shape(const Skew &sk, int &by, int &nch);
// iteration t works on hyperplane tau - lag*t in launch tau (lag 2: a_smooth == 1 kernel; lag 4 = SM_LAG:
// the a_smooth != 1 kernels, whose P-stage and sweep share one schedule two launches apart)
SorSched build_sor_schedule(const Skew &sk, int iterations, int by, int lag = 2);
void free_sor_schedule(SorSched &s);
// Runs all `iterations` pipelined hyperplane steps.  Returns the number of kernel launches.
template <typename S>
long long launch_sor(hipStream_t st, const SorArgsT<S> &a, bool fp64, const SorChainSched &sched);

// Window form of the same sweep (k_sor_win.hip): one workgroup per (tile of 16 x 16 lines, psi window); launches go
// by tile diagonals.  `tiles`: device array of all launches back to back.
struct WinTile;
template <typename S> struct WinArgs;
struct WinSched {
    std::vector<int> first, count, nbuild;  // per launch: its tiles, how many of them (the first ones) build their system
    WinTile *tiles = nullptr;               // device
};
WinSched build_win_schedule(const Skew &sk, int iterations, int update_lag);
void free_win_schedule(WinSched &ws);
bool sor_win_supports(int C);
template <typename S> bool sor_win_storage(int C);  // storage formats / channel counts the window kernel is built for
bool sor_win_fits(const Skew &sk);
template <typename S>
long long launch_sor_win(hipStream_t st, const WinArgs<S> &wa, bool fp64, const WinSched &ws);

// K8 median (natural layout)
void launch_median5(hipStream_t st, const float *in, int Z, int Y, int X, float *out);
// nf <= 3 fields of one volume in one launch (field f at in + f*fstride): out[f] = median, or out[f] += median
void launch_median5_fields(hipStream_t st, const float *in, long long fstride, int nf, int Z, int Y, int X,
                           float *const *out, bool accumulate);
bool median_can_accumulate(int Z, int Y, int X);
// exact level tail of the fp64 / packed solver modes: u[f] = RN32(u[f] + median64(in[f])), three fp64 fields
void launch_median5_fields_f64(hipStream_t st, const double *in, long long fstride, int Z, int Y, int X, float *v32,
                               float *const *u);
void launch_accum_round_once(hipStream_t st, float *u, const double *d, long long n);

// f-1 preprocessing (k_preproc.hip)
template <typename TIN>
void launch_gauss_pass(hipStream_t st, const TIN *in, int cs, int co, double nmin, double nden, int T, int Z,
                       int Y, int X, int axis, const double *w, int radius, int mode, double *out);
template <typename TOUT>
void launch_store_channel(hipStream_t st, const double *in, long long n, int C, int c, TOUT *out);
// radius-4 pass (sigma 1): normalise once per loaded element (NORM), register window / LDS segment, output element e at
// out[e * ocs + oco] in TOUT (the last pass of a channel writes the caller's channels-last array); false = not covered
template <typename TIN, typename TOUT, bool NORM>
bool launch_gauss_pass4(hipStream_t st, const TIN *in, int cs, int co, double nmin, double nden, int T, int Z, int Y, int X,
                        int axis, const double *w, int radius, TOUT *out, int ocs, int oco);

// f-2 statistics (k_misc.hip): partial = nblocks x 6 doubles (sum|w|, max|w|, sum div, sum u, sum v, sum w)
void launch_flow_stats(hipStream_t st, const float *flow, int Z, int Y, int X, int nblocks, double *partial);

// f-4 update_reference: acc = first ? x : acc + x (fp64), out[t*C+c] = acc[t] / count
void launch_accum_f64(hipStream_t st, double *acc, const float *x, long long n, bool first);
void launch_mean_store(hipStream_t st, const double *acc, long long n, int C, int c, double count, double *out);

// np.mean(axis=0) of a float32 stack of `count` arrays of n elements (float32 accumulation in stack order)
void launch_mean_stack_f32(hipStream_t st, const float *stack, int count, long long n, float *out);

// K9 pointwise helpers
void launch_axpy(hipStream_t st, float *y, const float *x, long long n);  // y += x
void launch_fill(hipStream_t st, float *y, float v, long long n);
// eight read-only streams of n floats each starting at x (measurement aid)
void launch_read8(hipStream_t st, const float *x, long long n, float *sink);
// interleave/deinterleave between (n,C) channels-last and planar (C,n)
void launch_pack(hipStream_t st, const float *planar, int C, long long n, float *interleaved);
void launch_pack3(hipStream_t st, const float *a, const float *b, const float *c, long long n, float *interleaved);
void launch_unpack(hipStream_t st, const float *interleaved, int C, long long n, float *planar);

}  // namespace fr3d
