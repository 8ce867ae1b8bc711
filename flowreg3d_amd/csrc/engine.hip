// engine.hip -- host side of the gfx950 engine: device workspace, resampling tables, the
// coarse-to-fine level loop of get_displacement (core/optical_flow_3d.py:319-542), the executor
// per-volume body (parallelization/sequential_3d.py:148-175) and the C ABI of
// include/flowreg3d_hip.h.  Everything between the entry copy-in and the exit copy-out stays in
// HBM; the only host work per level is the (cached) table build and kernel launches.
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "fr3d_internal.h"
#include "k_sor_win_core.h"

namespace fr3d {

// ---------------------------------------------------------------------------------------------
// schedule (core/optical_flow_3d.py:77-85, 389-408) -- Python round() is round-half-even
// ---------------------------------------------------------------------------------------------
static long py_round(double x) { return (long)std::nearbyint(x); }

static int warping_depth(double eta, int levels, int p, int m, int n)
{
    double min_dim = (double)std::min(p, std::min(m, n));
    int depth = 0;
    for (int q = 0; q < levels; q++) {
        depth += 1;
        min_dim *= eta;
        if (py_round(min_dim) < 10) break;
    }
    return depth;
}

struct Level {
    int idx;  // pyramid index i (0 = full resolution)
    int z, y, x;
};

static std::vector<Level> make_schedule(int p, int m, int n, double eta, int levels, int &min_level)
{
    int mlz = warping_depth(eta, levels, p, m, n);
    int mly = warping_depth(eta, levels, m, n, p);
    int mlx = warping_depth(eta, levels, n, p, m);
    int ml = std::min(mlx, std::min(mly, mlz)) * 4;
    mlz = std::min(mlz, ml); mly = std::min(mly, ml); mlx = std::min(mlx, ml);
    int top = std::max(mlx, std::max(mly, mlz));
    if (top <= min_level) min_level = top - 1;
    if (min_level < 0) min_level = 0;
    std::vector<Level> out;
    for (int i = top; i >= min_level; i--) {
        Level L;
        L.idx = i;
        L.z = (int)py_round((double)p * std::pow(eta, (double)std::min(i, mlz)));
        L.y = (int)py_round((double)m * std::pow(eta, (double)std::min(i, mly)));
        L.x = (int)py_round((double)n * std::pow(eta, (double)std::min(i, mlx)));
        // e.g. an axis of length 1 with eta = 0.5: round(0.5) = 0.  The reference fails here as well
        // (ZeroDivisionError in imresize_fused_gauss_cubic3D, util/resize_util_3D.py:116-128)
        FR3D_CHECK(L.z >= 1 && L.y >= 1 && L.x >= 1, "pyramid level with a zero-sized axis (eta too small for this volume)");
        out.push_back(L);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// resampling tables (util/resize_util_3D.py:53-111), host side, fp32 exactly like the reference
// run under NumPy promotion rules (fp64 cubic kernel rounded to fp32, fp32 products and sums).
// ---------------------------------------------------------------------------------------------
static double keys_cubic(double x)
{
    const double A = -0.75;
    double ax = std::fabs(x);
    if (ax < 1.0) return (A + 2.0) * std::pow(ax, 3.0) - (A + 3.0) * std::pow(ax, 2.0) + 1.0;
    if (ax < 2.0) return A * std::pow(ax, 3.0) - 5.0 * A * std::pow(ax, 2.0) + 8.0 * A * ax - 4.0 * A;
    return 0.0;
}

static int reflect_index(int j, int n)
{
    if (n <= 1) return 0;
    while (j < 0 || j >= n) j = (j < 0) ? (-j - 1) : (2 * n - 1 - j);
    return j;
}

// np.exp on float32 as NumPy computes it on AVX2/AVX512 hosts (Cody-Waite + P5/Q2 rational, fp32
// FMAs; numpy/_core/src/umath/loops_exponent_log.dispatch.c.src) -- the Gaussian taps of the
// reference go through it (util/resize_util_3D.py:106) and libm's expf is 1 ulp off often enough to
// matter at the 1e-5 flow level.
static float numpy_expf(float x)
{
    const float log2e = 0x1.715476p+0f, magic = 0x1.800000p+23f;
    const float c1 = -0x1.62e400p-1f, c2 = -0x1.7f7d1cp-20f;
    const float P0 = 9.999999999980870924916e-01f, P1 = 7.257664613233124478488e-01f,
                P2 = 2.473615434895520810817e-01f, P3 = 5.114512081637298353406e-02f,
                P4 = 6.757896990527504603057e-03f, P5 = 5.082762527590693718096e-04f;
    const float Q0 = 1.0f, Q1 = -2.742335390411667452936e-01f, Q2 = 2.159509375685829852307e-02f;
    if (x < -87.0f) return 0.0f;
    float q = x * log2e;
    q = (q + magic) - magic;
    float r = std::fmaf(q, c1, x);
    r = std::fmaf(q, c2, r);
    float num = std::fmaf(P5, r, P4);
    num = std::fmaf(num, r, P3);
    num = std::fmaf(num, r, P2);
    num = std::fmaf(num, r, P1);
    num = std::fmaf(num, r, P0);
    float den = std::fmaf(Q2, r, Q1);
    den = std::fmaf(den, r, Q0);
    return std::ldexp(num / den, (int)q);
}

static float pairwise_sum(const std::vector<float> &a)
{
    const int n = (int)a.size();
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    float r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

struct HostTable {
    int P;
    std::vector<int> idx;
    std::vector<float> wt;
};

static HostTable build_table(int in_len, int out_len, double sigma)
{
    const double scale = (double)out_len / (double)in_len;
    int R = 0;
    std::vector<float> g(1, 1.0f);
    if (sigma > 0.0) {
        R = (int)std::ceil(2.0 * sigma);
        g.assign(2 * R + 1, 0.0f);
        const float sig = (float)sigma;
        for (int k = 0; k < 2 * R + 1; k++) {
            float q = (float)(k - R) / sig;
            g[k] = numpy_expf(-0.5f * (q * q));
        }
        float s = pairwise_sum(g);
        for (auto &v : g) v = v / s;
    }
    HostTable t;
    t.P = 2 * R + 4;
    t.idx.resize((size_t)out_len * t.P);
    t.wt.resize((size_t)out_len * t.P);
    for (int i = 0; i < out_len; i++) {
        double x = ((double)i + 0.5) / scale - 0.5;
        int left = (int)std::floor(x - 2.0) - R;
        float ssum = 0.0f;
        for (int p = 0; p < t.P; p++) {
            int j = left + p;
            t.idx[(size_t)i * t.P + p] = reflect_index(j, in_len);
            double d = x - (double)j;
            float acc = 0.0f;
            for (int u = -R; u <= R; u++) acc += g[u + R] * (float)keys_cubic(d - (double)u);
            t.wt[(size_t)i * t.P + p] = acc;
            ssum += acc;
        }
        float inv = 1.0f / ssum;
        for (int p = 0; p < t.P; p++) t.wt[(size_t)i * t.P + p] *= inv;
    }
    return t;
}

// ---------------------------------------------------------------------------------------------
// engine state
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *ensure(size_t bytes)
    {
        if (bytes > cap) {
            if (p) FR3D_HIP(hipFree(p));
            p = nullptr; cap = 0;
            // a little headroom so that slightly larger requests reuse the buffer; capped, or the big slabs of a
            // 1024^3 volume would waste tens of GB
            size_t want = bytes + std::min<size_t>(bytes >> 4, (size_t)256 << 20) + 256;
            FR3D_HIP(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

struct DevTable {
    int P = 0;
    int *idx = nullptr;
    float *wt = nullptr;
};

struct ProfSpan {
    int kid;
    hipEvent_t a, b;
};

struct Engine {
    bool inited = false;
    int device = -1;
    hipStream_t st = nullptr;
    std::map<std::string, DevBuf> bufs;
    std::map<std::tuple<int, int, long long>, DevTable> tables;  // (in,out,sigma bits)
    std::map<std::tuple<int, int, int, int, int>, SorSched> scheds;  // (Z,Y,X,iterations,lag) of a level (a_smooth != 1)
    std::map<std::tuple<int, int, int, int, int, int>, SorChainSched> chain_scheds;  // (Z,Y,X,iterations,rows,chain): a_smooth == 1
    int slab_slots = 0;  // volumes per lock-step batch the solver slabs were last sized for
    // the per-volume solver slabs and level flows (everything get_displacement_core_t sizes by the batch)
    void release_slabs()
    {
        for (auto &kv : bufs) {
            const std::string &k = kv.first;
            if (k.compare(0, 4, "M_sk") == 0 || k.compare(0, 4, "A_sk") == 0 || k.compare(0, 4, "L_sk") == 0 ||
                k.compare(0, 4, "d_sk") == 0 || k.compare(0, 4, "E_sk") == 0 || k.compare(0, 3, "sm_") == 0 ||
                k.compare(0, 3, "uvw") == 0)
                kv.second.release();
        }
        slab_slots = 0;
    }
    std::map<std::tuple<int, int, int, int>, WinSched> win_scheds;  // (Z,Y,iterations,update_lag): window sweep
    const WinSched &win_sched(const Skew &sk, int iterations, int update_lag)
    {
        auto key = std::make_tuple(sk.Z, sk.Y, iterations, update_lag);
        auto it = win_scheds.find(key);
        if (it == win_scheds.end()) it = win_scheds.emplace(key, build_win_schedule(sk, iterations, update_lag)).first;
        return it->second;
    }

    // device tables of the compact skewed layout per level geometry (fr3d_internal.h: Skew::pb / cp)
    struct Compact {
        long long *pb = nullptr;
        int *cp = nullptr;
        long long total = 0;
    };
    std::map<std::tuple<int, int, int>, Compact> compacts;
    Skew compact_skew(int Z, int Y, int X)
    {
        auto key = std::make_tuple(Z, Y, X);
        auto it = compacts.find(key);
        if (it == compacts.end()) {
            std::vector<long long> pb;
            std::vector<int> cp;
            Compact c;
            c.total = make_compact_tables(Z, Y, X, pb, cp);
            FR3D_HIP(hipMalloc((void **)&c.pb, pb.size() * sizeof(long long)));
            FR3D_HIP(hipMalloc((void **)&c.cp, cp.size() * sizeof(int)));
            FR3D_HIP(hipMemcpy(c.pb, pb.data(), pb.size() * sizeof(long long), hipMemcpyHostToDevice));
            FR3D_HIP(hipMemcpy(c.cp, cp.data(), cp.size() * sizeof(int), hipMemcpyHostToDevice));
            it = compacts.emplace(key, c).first;
        }
        Skew sk = make_skew(Z, Y, X);
        sk.pb = it->second.pb;
        sk.cp = it->second.cp;
        sk.total = it->second.total;
        return sk;
    }

    const SorSched &sched(const Skew &sk, int iterations, int lag)
    {
        FR3D_CHECK(lag == SM_LAG, "internal: per-iteration schedules serve the a_smooth != 1 kernels");
        auto key = std::make_tuple(sk.Z, sk.Y, sk.X, iterations, lag);
        auto it = scheds.find(key);
        if (it == scheds.end())  // the lag-4 (a_smooth != 1) kernels are written for 64 x 4 workgroups
            it = scheds.emplace(key, build_sor_schedule(sk, iterations, 4, lag)).first;
        return it->second;
    }
    // by, nch <= 0: the plane sweep's own tile shape
    const SorChainSched &chain_sched(const Skew &sk, int iterations, int by = 0, int nch = 0)
    {
        if (by <= 0 || nch <= 0) sor_tile_shape(sk, by, nch);
        auto key = std::make_tuple(sk.Z, sk.Y, sk.X, iterations, by, nch);
        auto it = chain_scheds.find(key);
        if (it == chain_scheds.end()) {
            it = chain_scheds.emplace(key, build_sor_chain_schedule(sk, iterations, by, nch)).first;
        }
        return it->second;
    }
    // profiling
    bool prof = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    fr3d_kernel_stat acc[FR3D_K_COUNT];

    float *f32(const std::string &name, size_t n) { return (float *)bufs[name].ensure(n * sizeof(float)); }
    double *f64(const std::string &name, size_t n) { return (double *)bufs[name].ensure(n * sizeof(double)); }

    hipEvent_t get_event()
    {
        if (!ev_pool.empty()) {
            hipEvent_t e = ev_pool.back();
            ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        FR3D_HIP(hipEventCreate(&e));
        return e;
    }

    // device copies of SciPy's 1-D Gaussian kernels (f-1 preprocessing), uploaded once per (sigma, truncate)
    std::map<std::pair<long long, long long>, std::pair<double *, int>> gkernels;
    const double *gauss_kernel(double sigma, double truncate, int &radius);

    const DevTable &table(int in_len, int out_len, double sigma)
    {
        long long bits;
        std::memcpy(&bits, &sigma, sizeof(bits));
        auto key = std::make_tuple(in_len, out_len, bits);
        auto it = tables.find(key);
        if (it != tables.end()) return it->second;
        HostTable h = build_table(in_len, out_len, sigma);
        DevTable d;
        d.P = h.P;
        FR3D_HIP(hipMalloc((void **)&d.idx, h.idx.size() * sizeof(int)));
        FR3D_HIP(hipMalloc((void **)&d.wt, h.wt.size() * sizeof(float)));
        FR3D_HIP(hipMemcpy(d.idx, h.idx.data(), h.idx.size() * sizeof(int), hipMemcpyHostToDevice));
        FR3D_HIP(hipMemcpy(d.wt, h.wt.data(), h.wt.size() * sizeof(float), hipMemcpyHostToDevice));
        return tables.emplace(key, d).first->second;
    }
};

static Engine g_eng;
// Second lane (fr3d_set_lanes(2)): an engine of its own -- stream, workspace, tables, schedules -- on the same device.
// fr3d_process_batch* then deals the lock-step batches of a series alternately to the two lanes, so that the
// compute-bound stages of one lane (median, warp, motion tensor, the coarse levels' short launches) run under the
// other lane's sweep.  Everything else (single-volume entries, preprocessing, statistics) runs on lane 0.
static Engine g_eng2;
static int g_lanes = 2;
static thread_local Engine *g_cur = &g_eng;  // the lane the stage wrappers below enqueue on: lane 1 is driven by a host thread of its own
static std::recursive_mutex g_mu;
static thread_local std::string g_err;

struct Span {
    Engine &e;
    int kid;
    hipEvent_t a = nullptr;
    Span(Engine &eng, int k, double bytes, long long launches, long long units) : e(eng), kid(k)
    {
        if (!e.prof) return;
        a = e.get_event();
        FR3D_HIP(hipEventRecord(a, e.st));
        e.acc[k].algo_bytes += bytes;
        e.acc[k].launches += launches;
        e.acc[k].units += units;
    }
    void add(double bytes, long long launches, long long units)
    {
        if (!e.prof) return;
        e.acc[kid].algo_bytes += bytes;
        e.acc[kid].launches += launches;
        e.acc[kid].units += units;
    }
    ~Span()
    {
        if (!a) return;
        hipEvent_t b = e.get_event();
        (void)hipEventRecord(b, e.st);
        e.spans.push_back({kid, a, b});
    }
};

static void ensure_init()
{
    if (!g_eng.inited) throw Error("fr3d_init() has not been called (or failed)");
    // the current device is a per-thread setting: a caller on another host thread than fr3d_init's gets the engine's device
    FR3D_HIP(hipSetDevice(g_eng.device));
}

// ---------------------------------------------------------------------------------------------
// stage wrappers
// ---------------------------------------------------------------------------------------------

// imresize_fused_gauss_cubic3D for one channel: src (D,H,W) with channel stride cs/offset co
// -> planar dst (od,oh,ow)
static void resize3d(Engine &e, const float *src, int cs, int co, int D, int H, int W, int od, int oh,
                     int ow, float *dst, double sigma_coeff = 0.6, bool per_axis = false)
{
    double sz = (double)od / D, sy = (double)oh / H, sx = (double)ow / W;
    double s = sx;
    if (sy < s) s = sy;
    if (sz < s) s = sz;
    double sig = (s < 1.0) ? (sigma_coeff / s) : 0.0;
    double sigx = sig, sigy = sig, sigz = sig;
    if (per_axis) {  // util/resize_util_3D.py:120-123: each axis' sigma from its own scale
        sigx = sx < 1.0 ? sigma_coeff / sx : 0.0;
        sigy = sy < 1.0 ? sigma_coeff / sy : 0.0;
        sigz = sz < 1.0 ? sigma_coeff / sz : 0.0;
    }
    // A pass whose axis keeps its length without smoothing has the taps (0, 0, 1, 0) exactly (Keys' kernel at
    // -2, -1, 0, 1): its output is its input, so it is skipped (finite inputs; the x pass also gathers the
    // channel out of an interleaved source, so it only goes when the source is planar).
    const bool idx_ = W == ow && sigx == 0.0 && cs == 1, idy_ = H == oh && sigy == 0.0, idz_ = D == od && sigz == 0.0;
    const float *cur = src;
    int ccs = cs, cco = co;
    double bytes = 0.0;
    int passes = 0;
    Span sp(e, FR3D_K_RESIZE, 0, 0, (long long)od * oh * ow);
    if (!idx_) {
        const DevTable &tx = e.table(W, ow, sigx);
        float *o = (idy_ && idz_) ? dst : e.f32("rs_t1", (size_t)D * H * ow);
        launch_resize_pass(e.st, cur, ccs, cco, D, H, W, 2, ow, tx.idx, tx.wt, tx.P, o);
        bytes += 4.0 * ((double)D * H * W + (double)D * H * ow);
        passes++;
        cur = o; ccs = 1; cco = 0;
    }
    if (!idy_) {
        const DevTable &ty = e.table(H, oh, sigy);
        float *o = idz_ ? dst : e.f32("rs_t2", (size_t)D * oh * ow);
        launch_resize_pass(e.st, cur, 1, 0, D, H, ow, 1, oh, ty.idx, ty.wt, ty.P, o);
        bytes += 4.0 * ((double)D * H * ow + (double)D * oh * ow);
        passes++;
        cur = o;
    }
    if (!idz_) {
        const DevTable &tz = e.table(D, od, sigz);
        launch_resize_pass(e.st, cur, 1, 0, D, oh, ow, 0, od, tz.idx, tz.wt, tz.P, dst);
        bytes += 4.0 * ((double)D * oh * ow + (double)od * oh * ow);
        passes++;
        cur = dst;
    }
    if (cur != dst) {  // all three axes unchanged: a copy
        FR3D_HIP(hipMemcpyAsync(dst, src, (size_t)od * oh * ow * sizeof(float), hipMemcpyDeviceToDevice, e.st));
        bytes += 8.0 * (double)od * oh * ow;
    }
    sp.add(bytes, passes, 0);
}

// cubic warp of one channel
template <typename TV, typename TF, typename TR = TV, typename TO = float>
static void warp_cubic_chan(Engine &e, const TV *vol, int vcs, int vco, const TF *pu, const TF *pv,
                            const TF *pw, int fs, double hx, double hy, double hz, const TR *ref,
                            int Z, int Y, int X, TO *out, int ocs, int oco)
{
    const bool compact = prefilter_compact_ok(Z, Y, X);
    const int npad = compact ? 2 : 12;  // stored pad (the filter always runs over SciPy's 12)
    const size_t np = (size_t)(Z + 2 * npad) * (Y + 2 * npad) * (X + 2 * npad);
    double *coef = e.f64("warp_coef", np);
    if (compact) {
        const double N = (double)Z * Y * X, s1 = (double)(Z + 4) * Y * X, s2 = (double)(Z + 4) * (Y + 4) * X;
        Span sp(e, FR3D_K_PREFILTER, sizeof(TV) * N + 24.0 * s1 + 8.0 * s1 + 24.0 * s2 + 8.0 * s2 + 24.0 * np, 3, (long long)np);
        double *tmp = e.f64("warp_tmp", (size_t)(Z + 4) * (Y + 4) * X);
        launch_prefilter3_compact<TV>(e.st, vol, vcs, vco, Z, Y, X, coef, tmp);
    } else {
        Span sp(e, FR3D_K_PREFILTER, 8.0 * np * 7.0, 4, (long long)np);
        launch_pad_edge<TV>(e.st, vol, vcs, vco, Z, Y, X, npad, coef);
        launch_prefilter3(e.st, coef, Z + 2 * npad, Y + 2 * npad, X + 2 * npad);
    }
    {
        Span sp(e, FR3D_K_WARP, 24.0 * (double)Z * Y * X, 1, (long long)Z * Y * X);
        launch_warp_cubic<TF, TR, TO>(e.st, coef, npad, pu, pv, pw, fs, hx, hy, hz, ref, vcs, vco, Z, Y, X,
                                      out, ocs, oco);
    }
}

struct RefPyramid {
    // per level, per channel planar arrays (owned by engine buffers)
    std::vector<float *> f1;  // [level] -> C*nl floats
    std::vector<float *> wl;  // [level] -> C*nl floats
};

// ---------------------------------------------------------------------------------------------
// get_displacement on device pointers
// ---------------------------------------------------------------------------------------------
static void build_ref_pyramid(Engine &e, const std::vector<Level> &lv, const float *fixed,
                              const float *weight, int Z, int Y, int X, int C, RefPyramid &rp,
                              const std::string &tag)
{
    const size_t nfull = (size_t)Z * Y * X;
    const float *wsrc = weight;
    if (!weight) {
        // weight=None -> ones/C (core/optical_flow_3d.py:351-352); resized like any other weight
        float *wc = e.f32("w_const", nfull * C);
        launch_fill(e.st, wc, (float)(1.0 / (double)C), (long long)nfull * C);
        wsrc = wc;
    }
    rp.f1.clear();
    rp.wl.clear();
    for (size_t li = 0; li < lv.size(); li++) {
        const Level &L = lv[li];
        const size_t nl = (size_t)L.z * L.y * L.x;
        float *f1 = e.f32(tag + "f1_" + std::to_string(li), nl * C);
        float *wl = e.f32(tag + "wl_" + std::to_string(li), nl * C);
        for (int c = 0; c < C; c++) {
            resize3d(e, fixed, C, c, Z, Y, X, L.z, L.y, L.x, f1 + (size_t)c * nl);
            resize3d(e, wsrc, C, c, Z, Y, X, L.z, L.y, L.x, wl + (size_t)c * nl);
        }
        rp.f1.push_back(f1);
        rp.wl.push_back(wl);
    }
}

static bool use_window_sweep(const fr3d_params &p, int C, bool fp64_storage);

// Solve `nb` volumes against the same reference pyramid in lock step: every stage before and after
// the solver runs per volume, the SOR launches advance all nb volumes at once (the launch count per
// level is fixed by the wavefront schedule, so batching multiplies the work per launch and hides
// the pipeline fill/drain launches that are too small to occupy the chip).
template <typename S>
static void get_displacement_core_t(Engine &e, const fr3d_params &p, const std::vector<Level> &lv,
                                    int min_level, const RefPyramid &rp, int nb, const float *const *moving,
                                    int Z, int Y, int X, int C, const float *uvw_init, float *const *flow_out,
                                    int reserve_nb)
{
    // separate workspaces per storage type
    const std::string sn = std::is_same<S, pk42>::value ? "p42" : (sizeof(S) == 8 ? "64" : "");
    using WT = typename StoWt<S>::type;
    FR3D_CHECK(p.a_smooth >= 0.0, "a_smooth must be >= 0");
    FR3D_CHECK(nb >= 1 && nb <= 64, "internal: bad batch size");
    const size_t nfull = (size_t)Z * Y * X;
    std::vector<float *> uvw(3 * nb, nullptr), uvw_prev(3 * nb, nullptr);
    int pz = 0, py = 0, px = 0;
    int flip = 0;

    for (size_t li = 0; li < lv.size(); li++) {
        const Level &L = lv[li];
        const int lz = L.z, ly = L.y, lx = L.x;
        const size_t nl = (size_t)lz * ly * lx;
        const double hz = (double)Z / lz, hy = (double)Y / ly, hx = (double)X / lx;
        const float *f1l = rp.f1[li];

        // solver operands in the skewed layout, one slab per volume of the batch
        // compact skewed layout, records; a_smooth == 1: k_sor.hip, otherwise the psi_smooth kernels (k_sor_smooth.hip)
        const bool fast = p.a_smooth == 1.0;
        const Skew sk = e.compact_skew(lz, ly, lx);
        const size_t ns = (size_t)sk.total;
        const size_t nres = (size_t)std::max(nb, reserve_nb);  // slabs reserved (>= nb)
        e.slab_slots = std::max(e.slab_slots, (int)nres);
        // sizes and strides in storage elements (sto_elems: values for float / double, 4 dwords per 3 values for pk42)
        const size_t e3 = (size_t)sto_elems<S>((long long)ns * 3), e9 = 3 * e3, e12 = 4 * e3;
        S *Mbuf = (S *)e.bufs["M_sk" + sn].ensure(e9 * nres * sizeof(S));
        S *Abuf = (S *)e.bufs["A_sk" + sn].ensure(e12 * C * nres * sizeof(S));
        WT *wsk = (WT *)e.bufs["w_sk" + sn].ensure(ns * C * sizeof(WT));
        S *Lbuf = (S *)e.bufs["L_sk" + sn].ensure(e3 * nres * sizeof(S));
        S *dbuf = (S *)e.bufs["d_sk" + sn].ensure(e3 * nres * sizeof(S));
        SorArgsT<S> a;
        std::memset(&a, 0, sizeof(a));
        a.sk = sk;
        a.C = C;
        a.nvol = nb;
        a.vsM = (long long)e9;
        a.vsA = (long long)e12 * C;
        a.vsL = (long long)e3;
        a.vsD = (long long)e3;
        // alpha schedule (:485-490) and alpha/h^2 (level_solver_3d.py:473-475)
        const double sc = (L.idx == min_level) ? 1.0 : std::pow(p.eta, -0.5 * (double)L.idx);
        a.ax = (sc * p.alpha[0]) / (hx * hx);
        a.ay = (sc * p.alpha[1]) / (hy * hy);
        a.az = (sc * p.alpha[2]) / (hz * hz);
        for (int c = 0; c < C; c++) {
            a.A[c] = Abuf + (size_t)c * e12;
            a.weight[c] = wsk + (size_t)c * ns;
            a.a_data[c] = p.a_data[c];
        }
        a.M = Mbuf;
        a.L = Lbuf;
        a.d = dbuf;
        {
            Span sp(e, FR3D_K_OTHER, 0, 0, 0);
            for (int c = 0; c < C; c++) launch_skew_pack<float, WT>(e.st, rp.wl[li] + (size_t)c * nl, 0, wsk + (size_t)c * ns, 1, sk);
            if (fast) FR3D_HIP(hipMemsetAsync(dbuf, 0, e3 * nb * sizeof(S), e.st));
        }

        const std::string sfx = flip ? "_a" : "_b";
        flip ^= 1;
        for (int b = nb; b < (int)nres; b++)  // reserved slots: allocate now, not inside a later timed batch
            for (int d = 0; d < 3; d++) (void)e.f32(std::string("uvw") + char('0' + d) + sfx + std::to_string(b), nl);
        for (int b = 0; b < nb; b++) {
            // the moving image on this level; at full resolution the resampler is the identity (see resize3d) and
            // a single-channel volume is read where it lies
            const float *f2l = moving[b];
            if (!(C == 1 && lz == Z && ly == Y && lx == X)) {
                float *buf = e.f32("f2l", nl * C);
                for (int c = 0; c < C; c++) resize3d(e, moving[b], C, c, Z, Y, X, lz, ly, lx, buf + (size_t)c * nl);
                f2l = buf;
            }

            // level flow (interior; ghosts are the edge pad of :88-89, implied)
            float **u = &uvw[3 * b], **up = &uvw_prev[3 * b];
            for (int d = 0; d < 3; d++) up[d] = u[d];
            for (int d = 0; d < 3; d++)
                u[d] = e.f32(std::string("uvw") + char('0' + d) + sfx + std::to_string(b), nl);
            const float *warped = f2l;
            if (li == 0) {
                for (int d = 0; d < 3; d++) {
                    if (uvw_init) resize3d(e, uvw_init, 3, d, Z, Y, X, lz, ly, lx, u[d]);
                    else launch_fill(e.st, u[d], 0.0f, (long long)nl);
                }
            } else {
                for (int d = 0; d < 3; d++) resize3d(e, up[d], 1, 0, pz, py, px, lz, ly, lx, u[d]);
                float *wbuf = e.f32("warped", nl * C);
                for (int c = 0; c < C; c++)
                    warp_cubic_chan<float, float>(e, f2l + (size_t)c * nl, 1, 0, u[0], u[1], u[2], 1, hx, hy, hz,
                                                  f1l + (size_t)c * nl, lz, ly, lx, wbuf + (size_t)c * nl, 1, 0);
                warped = wbuf;
            }
            {
                // factors (and for a_smooth == 1 the Laplacian terms; the psi_smooth kernels form their diffusion
                // stencil from u,v,w themselves) go straight into the solver's records (LDS-tiled kernels)
                Span sp(e, FR3D_K_TENSOR, 4.0 * (2 + 12) * nl * C, C, (long long)nl * C);
                for (int c = 0; c < C; c++) {
                    // only the square-root factors are needed: the solver rebuilds the tensor from
                    // them on psi-update iterations and keeps its own frozen 3x3 system in between
                    S *Adst = Abuf + (size_t)b * a.vsA + (size_t)c * e12;
                    launch_motion_tensor_rec<S>(e.st, f1l + (size_t)c * nl, warped + (size_t)c * nl, hz, hy, hx, Adst, sk);
                }
                if (fast) launch_laplace_rec<S>(e.st, u[0], u[1], u[2], sk, a.ax, a.ay, a.az, Lbuf + (size_t)b * a.vsL);
            }
        }
        a.iterations = p.iterations;
        a.update_lag = p.update_lag;
        if (p.a_smooth == 1.0) {
            Span sp(e, FR3D_K_SOR, 0, 0, 0);
            long long n;
            if (use_window_sweep(p, C, std::is_same<S, double>::value) && sor_win_fits(sk) && sor_win_storage<S>(C)) {
                // window sweep (k_sor_win.hip): the exports of the slots before the last one, laid out like d
                WinArgs<S> wa;
                std::memset(&wa, 0, sizeof(wa));
                wa.a = a;
                S *Ebuf = (S *)e.bufs["E_sk" + sn].ensure(e3 * (WIN_WMAX - 1) * nres * sizeof(S));
                // array q of all volumes together, so that a volume's stride is d's
                wa.E = Ebuf;
                wa.strideE = (long long)(e3 * nres);
                n = launch_sor_win<S>(e.st, wa, p.solver_fp64 != 0, e.win_sched(sk, p.iterations, p.update_lag));
            } else
                n = launch_sor<S>(e.st, a, p.solver_fp64 != 0, e.chain_sched(sk, p.iterations));
            // algorithmic traffic of the reference's update: 9C tensor entries + C (w psi) + 3 L + 3 d read, 3 d written,
            // in the solver's storage type: 4 (10C + 9) B with fp32 storage, twice that with fp64 storage, 4/3 of it
            // with packed 42-bit storage
            sp.add(sto_bytes_per_value<S>() * (10.0 * C + 9.0) * (double)nl * p.iterations * nb, n, (long long)nl * p.iterations * nb);
        } else if constexpr (!std::is_same<S, pk42>::value) {
            // a_smooth != 1 (k_sor_smooth.hip): psi_smooth every iteration, triple-buffered increments; the volumes
            // of the batch share the launches; the result is copied into the batch slab the common tail reads
            Span sp(e, FR3D_K_SOR, 0, 0, 0);
            S *smU = (S *)e.bufs["sm_U" + sn].ensure(ns * 3 * nb * sizeof(S));
            S *smD = (S *)e.bufs["sm_D" + sn].ensure(ns * 9 * nb * sizeof(S));
            S *smP = (S *)e.bufs["sm_P" + sn].ensure(ns * nb * sizeof(S));
            SmoothArgs<S> sa;
            std::memset(&sa, 0, sizeof(sa));
            sa.view.Z = lz; sa.view.Y = ly; sa.view.X = lx; sa.view.sk = sk;
            smooth_set_spacing(sa.view, hx, hy, hz);
            sa.view.a_smooth = p.a_smooth;
            sa.view.U = smU;
            sa.nvol = nb;
            sa.vsU = (long long)ns * 3; sa.vsD = (long long)ns * 3; sa.vsP = (long long)ns;
            sa.vsM = a.vsM; sa.vsA = a.vsA;
            for (int m = 0; m < 3; m++) sa.D[m] = smD + (size_t)m * 3 * ns * nb;
            sa.Ps = smP;
            sa.M = Mbuf;
            for (int c = 0; c < C; c++) {
                sa.A[c] = a.A[c];
                sa.weight[c] = a.weight[c];
                sa.a_data[c] = p.a_data[c];
            }
            sa.ax = a.ax; sa.ay = a.ay; sa.az = a.az;
            sa.C = C;
            sa.iterations = p.iterations;
            sa.update_lag = p.update_lag;
            sa.S_planes = sk.S;
#ifdef FR3D_EXPERIMENTS
            if (const char *v = std::getenv("FR3D_SM_DBG")) sa.dbg = std::atoi(v);
#endif
            {
                float *stage = e.f32("d_nat", nl * 3);  // the increments' scratch is free until the solver has run
                for (int b = 0; b < nb; b++) {
                    for (int d = 0; d < 3; d++)
                        FR3D_HIP(hipMemcpyAsync(stage + (size_t)d * nl, uvw[3 * b + d], nl * sizeof(float), hipMemcpyDeviceToDevice, e.st));
                    launch_skew_pack<float, S>(e.st, stage, (long long)nl, smU + (size_t)b * sa.vsU, 3, sk);
                }
            }
            FR3D_HIP(hipMemsetAsync(smD, 0, ns * 9 * nb * sizeof(S), e.st));
            long long n;
#ifdef FR3D_EXPERIMENTS
            // FR3D_SMOOTH=fused|paired: P-stage and sweep tiles sharing the plane between them (k_sor_smooth.hip; -12 % HBM
            // traffic, 7 % slower: profiles/r04/smooth_fusion.md)
            static const char *sm_env = getenv("FR3D_SMOOTH");
            const bool fused = sm_env && (!strcmp(sm_env, "fused") || !strcmp(sm_env, "paired")) && 2 * p.iterations <= 32767;
            if (fused)
                n = launch_sor_smooth_fused<S>(e.st, sa, e.sched(sk, p.iterations, SM_LAG), e.chain_sched(sk, 2 * p.iterations, 4, 2),
                                               !strcmp(sm_env, "paired"));
            else
#endif
            n = launch_sor_smooth<S>(e.st, sa, e.sched(sk, p.iterations, SM_LAG));
            if (p.iterations > 0) {
                for (int b = 0; b < nb; b++)
                    FR3D_HIP(hipMemcpyAsync(dbuf + (size_t)b * a.vsD, sa.D[(p.iterations - 1) % 3] + (size_t)b * sa.vsD,
                                            ns * 3 * sizeof(S), hipMemcpyDeviceToDevice, e.st));
            } else {
                FR3D_HIP(hipMemsetAsync(dbuf, 0, ns * 3 * nb * sizeof(S), e.st));
            }
            // algorithmic traffic of the reference's a_smooth != 1 iteration (level_solver_3d.py:400-471 + :262-311): the
            // sweep reads 9C tensor entries + C (w psi) + 3 u,v,w + 3 d + psi_smooth and writes 3 d; psi_smooth, re-evaluated
            // every iteration, reads 3 u,v,w + 3 d and writes 1: (10C + 17) values per voxel update
            sp.add((double)sizeof(S) * (10.0 * C + 17.0) * (double)nl * p.iterations * nb, n, (long long)nl * p.iterations * nb);
        } else {
            throw Error("internal: packed solver storage serves the a_smooth == 1 sweep only");
        }
        // increments back to the natural layout, 5^3 median (:517-526), accumulate (:527-529)
        const bool med = std::min(lz, std::min(ly, lx)) > 5;
        if constexpr (std::is_same<S, double>::value) {
            // fp64 storage: the level tail as the reference evaluates it -- fp64 increments, fp64 median, u + du in
            // fp64, ONE rounding to the fp32 the next level's resampler and warp read (k_median.hip: k_median5_refine).
            // With it this mode is within 4e-9 voxels of the CPU path at 256^3 / 512^3 and 2e-11 on the two-channel
            // config 5 (2.8e-4 with the fp32 tail below, whose double rounding of the level flow that iteration
            // amplifies).  Costs 5 ms per 256^3 volume; packed and fp32 storage keep the fp32 tail: their storage
            // rounding dominates (packed: 1.9e-5 with either tail).
            for (int b = 0; b < nb; b++) {
                double *dn = e.f64("d_nat64", nl * 3);
                float **u = &uvw[3 * b];
                {
                    Span sp(e, FR3D_K_OTHER, 0, 0, 0);
                    launch_unskew_unpack<S, double>(e.st, dbuf + (size_t)b * a.vsD, dn, (long long)nl, 3, sk);
                }
                if (med && median_can_accumulate(lz, ly, lx)) {
                    Span sp(e, FR3D_K_MEDIAN, 28.0 * nl * 3, 2, (long long)nl * 3);
                    launch_median5_fields_f64(e.st, dn, (long long)nl, lz, ly, lx, e.f32("d_nat", nl * 3), u);
                    continue;
                }
                double *dm = med ? e.f64("d_med64", nl) : nullptr;
                for (int d = 0; d < 3; d++) {
                    const double *inc = dn + (size_t)d * nl;
                    if (med) {  // levels too small for the tiled kernels: one thread per voxel, fp64 selection
                        Span sp(e, FR3D_K_MEDIAN, 16.0 * nl, 1, (long long)nl);
                        launch_median5_f64(e.st, inc, lz, ly, lx, dm);
                        inc = dm;
                    }
                    Span sp(e, FR3D_K_OTHER, 0, 0, 0);
                    launch_accum_round_once(e.st, u[d], inc, (long long)nl);
                }
            }
        } else
        for (int b = 0; b < nb; b++) {
            float *dn = e.f32("d_nat", nl * 3);
            float **u = &uvw[3 * b];
            {
                Span sp(e, FR3D_K_OTHER, 0, 0, 0);
                // increments leave the solver rounded to fp32: the next level (and the executor) cast to
                // fp32 anyway (util/resize_util_3D.py:116, sequential_3d.py:150) and the median commutes with it
                launch_unskew_unpack<S, float>(e.st, dbuf + (size_t)b * a.vsD, dn, (long long)nl, 3, sk);
            }
            if (med && median_can_accumulate(lz, ly, lx)) {
                // one launch for du, dv, dw, the flow update u += median(du) fused (:517-529)
                Span sp(e, FR3D_K_MEDIAN, 8.0 * nl * 3, 1, (long long)nl * 3);
                launch_median5_fields(e.st, dn, (long long)nl, 3, lz, ly, lx, u, true);
                continue;
            }
            float *dm = med ? e.f32("d_med", nl * 3) : nullptr;
            if (med) {
                Span sp(e, FR3D_K_MEDIAN, 8.0 * nl * 3, 3, (long long)nl * 3);
                for (int d = 0; d < 3; d++) launch_median5(e.st, dn + (size_t)d * nl, lz, ly, lx, dm + (size_t)d * nl);
            }
            {
                Span sp(e, FR3D_K_OTHER, 0, 0, 0);
                for (int d = 0; d < 3; d++) launch_axpy(e.st, u[d], (med ? dm : dn) + (size_t)d * nl, (long long)nl);
            }
        }
        pz = lz; py = ly; px = lx;
    }

    // :530-541
    const size_t nl = (size_t)pz * py * px;
    for (int b = 0; b < nb; b++) {
        float **u = &uvw[3 * b];
        if (min_level > 0) {
            float *full = e.f32("d_nat", nfull * 3);  // the increments' scratch is free after the last level
            for (int d = 0; d < 3; d++) resize3d(e, u[d], 1, 0, pz, py, px, Z, Y, X, full + (size_t)d * nfull);
            launch_pack3(e.st, full, full + nfull, full + 2 * nfull, (long long)nfull, flow_out[b]);
        } else {
            FR3D_CHECK(nl == nfull, "internal: finest level is not full resolution");
            launch_pack3(e.st, u[0], u[1], u[2], (long long)nfull, flow_out[b]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Verification mode (k_verify.hip): get_displacement with the reference's arithmetic, fp64 end to end, on the
// engine's data path -- compact skewed layout, hyperplane schedule, the pyramid's resampler / prefilter / gather /
// tensor stages.  One volume, no lock-step batch, no timing spans.  flow_out: (Z,Y,X,3) fp64 on the device.
// Mirrors core/optical_flow_3d.py:389-542 statement by statement (the level flow stays fp64 between the stages like
// the reference's `u = u + du`, and is rounded to fp32 only where the reference's resampler does it, :116 of
// util/resize_util_3D.py).
// ---------------------------------------------------------------------------------------------
static void get_displacement_verify(Engine &e, const fr3d_params &p, const std::vector<Level> &lv, int min_level,
                                    const RefPyramid &rp, const float *moving, int Z, int Y, int X, int C,
                                    const float *uvw_init, double *flow_out)
{
    const size_t nfull = (size_t)Z * Y * X;
    double *ud[3] = {nullptr, nullptr, nullptr}, *ud_prev[3] = {nullptr, nullptr, nullptr};
    int pz = 0, py = 0, px = 0, flip = 0;
    for (size_t li = 0; li < lv.size(); li++) {
        const Level &L = lv[li];
        const int lz = L.z, ly = L.y, lx = L.x;
        const size_t nl = (size_t)lz * ly * lx;
        const double hz = (double)Z / lz, hy = (double)Y / ly, hx = (double)X / lx;
        const float *f1l = rp.f1[li];
        const Skew sk = e.compact_skew(lz, ly, lx);
        const size_t ns = (size_t)sk.total;
        const float *f2l = moving;
        if (!(C == 1 && lz == Z && ly == Y && lx == X)) {
            float *buf = e.f32("f2l", nl * C);
            for (int c = 0; c < C; c++) resize3d(e, moving, C, c, Z, Y, X, lz, ly, lx, buf + (size_t)c * nl);
            f2l = buf;
        }
        // level flow: fp32 out of the resampler (:417-434), fp64 from here on
        const std::string sfx = flip ? "_a" : "_b";
        flip ^= 1;
        float *uf = e.f32("vf_uf", nl * 3);
        for (int d = 0; d < 3; d++) {
            ud_prev[d] = ud[d];
            ud[d] = e.f64(std::string("vf_u") + char('0' + d) + sfx, nl);
        }
        const float *warped = f2l;
        if (li == 0) {
            for (int d = 0; d < 3; d++) {
                if (uvw_init) resize3d(e, uvw_init, 3, d, Z, Y, X, lz, ly, lx, uf + (size_t)d * nl);
                else launch_fill(e.st, uf + (size_t)d * nl, 0.0f, (long long)nl);
            }
        } else {
            const size_t nprev = (size_t)pz * py * px;
            float *prevf = e.f32("vf_prev", nprev);
            for (int d = 0; d < 3; d++) {
                launch_cast<double, float>(e.st, ud_prev[d], (long long)nprev, prevf);
                resize3d(e, prevf, 1, 0, pz, py, px, lz, ly, lx, uf + (size_t)d * nl);
            }
            float *wbuf = e.f32("warped", nl * C);
            for (int c = 0; c < C; c++)
                warp_cubic_chan<float, float>(e, f2l + (size_t)c * nl, 1, 0, uf, uf + nl, uf + 2 * nl, 1, hx, hy, hz,
                                              f1l + (size_t)c * nl, lz, ly, lx, wbuf + (size_t)c * nl, 1, 0);
            warped = wbuf;
        }
        for (int d = 0; d < 3; d++) launch_cast<float, double>(e.st, uf + (size_t)d * nl, (long long)nl, ud[d]);

        // operands of the sweep as records on the compact skewed rows
        double *Jnat = e.f64("vf_Jnat", nl * 10);
        double *Jrec = e.f64("vf_Jrec", ns * 10 * C);
        float *wrec = e.f32("vf_wrec", ns * C);
        double *psi = e.f64("vf_psi", ns * C);
        double *Urec = e.f64("vf_Urec", ns * 3);
        double *Drec = e.f64("vf_Drec", ns * 3);
        VerifyArgs a;
        std::memset(&a, 0, sizeof(a));
        a.sk = sk;
        a.C = C;
        a.update_lag = p.update_lag;
        for (int c = 0; c < C; c++) {
            double *Jo[10];
            for (int q = 0; q < 10; q++) Jo[q] = Jnat + (size_t)q * nl;
            launch_motion_tensor<double, double>(e.st, f1l + (size_t)c * nl, warped + (size_t)c * nl, lz, ly, lx, hz, hy, hx, Jo,
                                                 (double *)nullptr, 0, nullptr);
            launch_skew_pack<double, double>(e.st, Jnat, (long long)nl, Jrec + (size_t)c * 10 * ns, 10, sk);
            launch_skew_pack<float, float>(e.st, rp.wl[li] + (size_t)c * nl, 0, wrec + (size_t)c * ns, 1, sk);
            a.J[c] = Jrec + (size_t)c * 10 * ns;
            a.w[c] = wrec + (size_t)c * ns;
            a.psi[c] = psi + (size_t)c * ns;
            a.a_data[c] = p.a_data[c];
        }
        launch_skew_pack<float, double>(e.st, uf, (long long)nl, Urec, 3, sk);
        FR3D_HIP(hipMemsetAsync(Drec, 0, ns * 3 * sizeof(double), e.st));
        a.U = Urec;
        a.D = Drec;
        // alpha schedule (:485-490) and alpha / h^2 (level_solver_3d.py:473-475)
        const double sc = (L.idx == min_level) ? 1.0 : std::pow(p.eta, -0.5 * (double)L.idx);
        a.ax = (sc * p.alpha[0]) / (hx * hx);
        a.ay = (sc * p.alpha[1]) / (hy * hy);
        a.az = (sc * p.alpha[2]) / (hz * hz);
#ifdef FR3D_EXPERIMENTS
        auto dump = [&](const char *tag, int ch, const void *dev, size_t bytes) {
            const char *dir = getenv("FR3D_VERIFY_DUMP");  // debugging aid: per-level intermediates as raw files
            if (!dir) return;
            std::vector<char> host(bytes);
            FR3D_HIP(hipStreamSynchronize(e.st));
            FR3D_HIP(hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost));
            const std::string path = std::string(dir) + "/g_L" + std::to_string(L.idx) + "_" + tag + std::to_string(ch) + ".bin";
            if (FILE *f = fopen(path.c_str(), "wb")) {
                fwrite(host.data(), 1, bytes, f);
                fclose(f);
            }
        };
        dump("warpedf", 0, warped, nl * C * sizeof(float));
        dump("uinitf", 0, uf, nl * 3 * sizeof(float));
#endif
        if (p.a_smooth == 1.0) {
            launch_sor_verify(e.st, a, e.chain_sched(sk, p.iterations));
        } else {
            // a_smooth != 1: psi_smooth of iteration t needs the whole field of t-1, so the iterations run one at a time
            // (speed is not the point of this mode): psi_smooth on the padded grid, then the S hyperplane steps of ONE
            // lexicographic sweep.  Dm2 = the increments two iterations back: the ghost ring the reference's padded
            // arrays hold while psi_smooth is evaluated (set_boundary_3d ran before the previous sweep).
            double *Dm2 = e.f64("vf_Dm2", ns * 3);
            double *Ps = e.f64("vf_Ps", (size_t)(lz + 2) * (ly + 2) * (lx + 2));
            FR3D_HIP(hipMemsetAsync(Dm2, 0, ns * 3 * sizeof(double), e.st));
            a.Ps = Ps;
            const SorChainSched &one = e.chain_sched(sk, 1);
            for (int it = 0; it < p.iterations; it++) {
                launch_psi_smooth_verify(e.st, sk, Urec, Drec, Dm2, p.a_smooth, hx, hy, hz, Ps);
                FR3D_HIP(hipMemcpyAsync(Dm2, Drec, ns * 3 * sizeof(double), hipMemcpyDeviceToDevice, e.st));
                a.t_base = it;
                launch_sor_verify(e.st, a, one);
            }
        }
        // :517-529 in fp64: increments back to the natural order, 5^3 median, u = u + du
        double *dn = e.f64("vf_dnat", nl * 3);
        launch_unskew_unpack<double, double>(e.st, Drec, dn, (long long)nl, 3, sk);
#ifdef FR3D_EXPERIMENTS
        dump("res", 0, dn, nl * 3 * sizeof(double));
#endif
#ifdef FR3D_EXPERIMENTS
        // numerics experiment (which part of the shipped modes' difference is the fp32 level tail?): FR3D_VERIFY_TAIL32=1
        // rounds the increments to fp32 before the median and keeps the level flow in fp32, like the shipped modes
        static const char *tail_env = getenv("FR3D_VERIFY_TAIL32");
        const bool tail32 = tail_env && atoi(tail_env) != 0;
        if (tail32) {
            float *tmpf = e.f32("vf_tail32", nl * 3);
            launch_cast<double, float>(e.st, dn, (long long)nl * 3, tmpf);
            launch_cast<float, double>(e.st, tmpf, (long long)nl * 3, dn);
        }
#endif
        const bool med = std::min(lz, std::min(ly, lx)) > 5;
        double *dm = med ? e.f64("vf_dmed", nl) : nullptr;
        for (int d = 0; d < 3; d++) {
            const double *inc = dn + (size_t)d * nl;
            if (med) {
                launch_median5_f64(e.st, inc, lz, ly, lx, dm);
                inc = dm;
            }
            launch_axpy_f64(e.st, ud[d], inc, (long long)nl);
#ifdef FR3D_EXPERIMENTS
            if (tail32) {
                float *tmpf = e.f32("vf_tail32", nl * 3);
                launch_cast<double, float>(e.st, ud[d], (long long)nl, tmpf);
                launch_cast<float, double>(e.st, tmpf, (long long)nl, ud[d]);
            }
#endif
        }
#ifdef FR3D_EXPERIMENTS
        for (int d = 0; d < 3; d++) dump("u", d, ud[d], nl * sizeof(double));
#endif
        pz = lz; py = ly; px = lx;
    }
    // :530-541
    const size_t nl = (size_t)pz * py * px;
    if (min_level > 0) {
        float *lo = e.f32("vf_prev", nl);
        float *hi = e.f32("vf_uf", nfull);
        double *full = e.f64("vf_dnat", nfull * 3);
        for (int d = 0; d < 3; d++) {
            launch_cast<double, float>(e.st, ud[d], (long long)nl, lo);
            resize3d(e, lo, 1, 0, pz, py, px, Z, Y, X, hi);
            launch_cast<float, double>(e.st, hi, (long long)nfull, full + (size_t)d * nfull);
        }
        launch_pack3_f64(e.st, full, full + nfull, full + 2 * nfull, (long long)nfull, flow_out);
    } else {
        FR3D_CHECK(nl == nfull, "internal: finest level is not full resolution");
        launch_pack3_f64(e.st, ud[0], ud[1], ud[2], (long long)nfull, flow_out);
    }
}

// Which kernel runs the a_smooth == 1 sweep of a level (fr3d_params.solver_sweep; env FR3D_SWEEP=planes|window
// overrides an automatic choice).  Both produce the same bits.  FR3D_SWEEP_AUTO is the plane sweep everywhere: the
// window sweep (k_sor_win.hip) is bound by its instruction stream (~490 per update) and measured behind it in every
// storage format but fp64, where the two are level (256^3 batch 8, same box: 129 against 132 ms of sweeps per volume;
// fp32 storage 101 against 65, packed 171 against 88 -- DESIGN.md section 4, round 4).
static bool use_window_sweep(const fr3d_params &p, int C, bool fp64_storage)
{
    (void)fp64_storage;
    if (!sor_win_supports(C) || p.iterations <= 0 || p.a_smooth != 1.0) return false;
    int sw = p.solver_sweep;
    if (sw == FR3D_SWEEP_AUTO) {
        static const char *env = getenv("FR3D_SWEEP");
        if (env && !strcmp(env, "window")) sw = FR3D_SWEEP_WINDOW;
        else if (env && !strcmp(env, "planes")) sw = FR3D_SWEEP_PLANES;
    }
    return sw == FR3D_SWEEP_WINDOW;
}

// FR3D_SOLVER_AUTO picks the cheapest mode that keeps the flow within 1e-4 voxels (mean end-point error) of the
// reference CPU path WITH MARGIN, as measured against full CPU runs (DESIGN.md section 2,
// tests/test_gpu_fullsize_parity.py, profiles/parity_fullsize.json):
//  * one channel, up to 2^22 voxels: fp32 storage with fp64 update arithmetic (128^3: ~3e-5);
//  * one channel, larger volumes: packed 42-bit storage (256^3: 1.6e-5, 512^3: 3.2e-5).  fp32 storage measures
//    8.6e-5 at 256^3 on the benchmark's own input recipe -- inside the bound by 10 %, too thin for a default -- and
//    1.5e-4 at 512^3;
//  * several channels: fp64 storage.  The two-channel iteration amplifies every rounding of the stored operands
//    (config 5, 256x512x512: 2.5e-4 with packed storage, outside the bound); with fp64 storage and the exact level
//    tail the flow is within 2e-11 of the CPU path there.
static int solver_mode(const fr3d_params &p, int C, long long nvox)
{
    int m = p.solver_fp64;
    // the psi_smooth solver (a_smooth != 1) has no packed form and is less sensitive: fp32 storage measures 1.6e-5 at
    // 256^3 (tests/golden/fullsize_cfg2_asmooth05.npz); fp64 storage above 2^25 voxels and for several channels
    if (m < 0 && p.a_smooth != 1.0) m = (C >= 2 || nvox > (1LL << 25)) ? 2 : 1;
    if (m < 0) m = C >= 2 ? 2 : (nvox > (1LL << 22) ? 3 : 1);
    // packed 42-bit storage exists for the a_smooth == 1 sweep; the psi_smooth solver takes fp64 storage instead
    if (m == 3 && p.a_smooth != 1.0) m = 2;
    return m;
}

static int resolve_mode(const fr3d_params &p, int C, int Z, int Y, int X, const std::vector<Level> &lv);

static void get_displacement_core(Engine &e, const fr3d_params &p_in, const std::vector<Level> &lv, int min_level,
                                  const RefPyramid &rp, int nb, const float *const *moving, int Z, int Y, int X,
                                  int C, const float *uvw_init, float *const *flow_out, int reserve_nb = 1)
{
    fr3d_params p = p_in;
    p.solver_fp64 = resolve_mode(p_in, C, Z, Y, X, lv);
    FR3D_CHECK(p.solver_fp64 >= 0 && p.solver_fp64 <= 3, "solver_fp64 must be FR3D_SOLVER_AUTO, 0, 1, 2 or 3");
    if (p.solver_fp64 == 3)
        get_displacement_core_t<pk42>(e, p, lv, min_level, rp, nb, moving, Z, Y, X, C, uvw_init, flow_out, reserve_nb);
    else if (p.solver_fp64 == 2)
        get_displacement_core_t<double>(e, p, lv, min_level, rp, nb, moving, Z, Y, X, C, uvw_init, flow_out, reserve_nb);
    else
        get_displacement_core_t<float>(e, p, lv, min_level, rp, nb, moving, Z, Y, X, C, uvw_init, flow_out, reserve_nb);
}

// How many volumes to solve in lock step: FR3D_BATCH (default 8: the sweep runs at 0.495 of the roofline with 4
// volumes per launch, 0.508 with 8, 0.514 with 16 at 256^3), bounded by free HBM
// (29 skewed operand arrays per volume and channel set).
static int g_batch_hint = 0;  // fr3d_set_batch()
static bool g_fast_path = true;  // a_smooth == 1 of the call in progress (the psi_smooth path holds 13 more values per voxel)
static double g_storage_bytes = 4.0;  // bytes per stored solver value of the call in progress
static bool g_window = false;         // the call in progress runs the window sweep (its export arrays count in the budget)

// Volumes solved in lock step: fr3d_set_batch(), else FR3D_BATCH, else 8 volumes of up to 2^24 voxels -- and the same
// number of VOXELS for smaller volumes (up to 128 of them): a level of a small volume is a few thousand launches of
// a few microseconds each whatever the batch, so a batch of 8 leaves the device idle (48^3: 384 volumes/s with 8 in lock
// step, 707 with 128; 100^3: 138 -> 181; 64 x 256 x 256: 51 -> 58; profiles/r04/shape_rates.md).  Same workspace by
// construction.
static int batch_wanted(long long nvox)
{
    static const char *env = getenv("FR3D_BATCH");
    if (g_batch_hint > 0) return g_batch_hint;
    if (env) return std::max(1, atoi(env));
    long long f = nvox > 0 ? (1LL << 24) / nvox : 1;
    f = std::max(1LL, std::min(16LL, f));
    return (int)(8 * f);
}

// Bytes one volume of a lock-step batch holds on the finest level with `bytes` per stored solver value, and the HBM
// the solver slabs may occupy right now.
static double solver_bytes_per_volume(const std::vector<Level> &lv, int C, double bytes, bool fast_path, bool window = false)
{
    const Level &F = lv.back();
    std::vector<long long> pb;
    std::vector<int> cp;
    const double total = (double)make_compact_tables(F.z, F.y, F.x, pb, cp);  // packed rows: 1.1-1.3x the voxel count
    const double nfin = (double)F.z * F.y * F.x;
    // skewed solver slabs + the level flows of a volume (two generations of u,v,w)
    // (the window sweep adds the exports of its first WIN_WMAX - 1 slots, records of 3)
    return total * bytes * (12.0 * C + 9.0 + 6.0 + (fast_path ? 0.0 : 13.0) + (window ? 3.0 * (WIN_WMAX - 1) : 0.0)) + nfin * 4.0 * 9.0;
}
// `lanes`: engine lanes the call will run on -- each holds its own per-level scratch
static double solver_budget(const std::vector<Level> &lv, int C, int lanes = 1)
{
    const Level &F = lv.back();
    const double nfin = (double)F.z * F.y * F.x;
    // volume-independent scratch of the finest level, per lane: moving level and its warp (2C), fp64 spline coefficients
    // and the y-pass scratch (~4.2), increments and their median (6); once: reference and weight pyramids (~4C)
    const double scratch = nfin * 4.0 * (lanes * (2.0 * C + 4.2 + 6.0) + 4.0 * C);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return -1.0;
    // what the solver slabs may occupy: the memory that is free now plus what the engine already
    // holds (its buffers are reused), minus a margin for the per-level scratch; never more than
    // 85 % of the device.  Memory the caller holds (a resident series, torch tensors) is respected.
    // (the host-staging windows "stg*" of fr3d_process_batch_raw stay in use during the call: not reusable)
    size_t held = 0;
    for (const Engine *en : {&g_eng, &g_eng2})
        for (const auto &kv : en->bufs)
            if (kv.first.compare(0, 3, "stg") != 0) held += kv.second.cap;
    double avail = (double)free_b + (double)held - scratch - 2.0 * 1073741824.0;
    // FR3D_MEM_CAP_MIB: pretend the device is this small (tests of the batch / lane decisions under a tight budget)
    static const char *cap_env = getenv("FR3D_MEM_CAP_MIB");
    if (cap_env && atof(cap_env) > 0.0) avail = std::min(avail, atof(cap_env) * 1048576.0);
    return std::min(0.85 * (double)total_b, avail);
}

static std::atomic<int> g_last_mode{-1};  // fr3d_last_solver_mode(); both lanes store the same value
static std::atomic<int> g_last_fallback{0};  // fr3d_last_solver_fallback()

// The solver mode of a call: solver_mode(), except that an AUTOMATIC choice of packed storage falls back to fp32
// storage when one volume's packed slabs do not fit the device and its fp32 slabs do (one 1024^3 volume: 164 GB of
// packed slabs against 123 GB) -- a volume that size is solved rather than refused; fr3d_last_solver_mode() tells.
static int resolve_mode(const fr3d_params &p, int C, int Z, int Y, int X, const std::vector<Level> &lv)
{
    int m = solver_mode(p, C, (long long)Z * Y * X);
    int fell = 0;
    if (p.solver_fp64 < 0 && m == 3) {
        const double budget = solver_budget(lv, C);
        if (budget > 0 && solver_bytes_per_volume(lv, C, 16.0 / 3.0, true) > budget &&
            solver_bytes_per_volume(lv, C, 4.0, true) <= budget) {
            m = 1;
            fell = 1;
        }
    }
    g_last_mode = m;
    g_last_fallback = fell;
    return m;
}

static int pick_batch(int T, const std::vector<Level> &lv, int C, int lanes = 1)
{
    int want = batch_wanted((long long)lv.back().z * lv.back().y * lv.back().x);
    if (want > T) want = T;
    if (want < 1) want = 1;
    const double per_vol = solver_bytes_per_volume(lv, C, g_storage_bytes, g_fast_path, g_window);
    const double budget = solver_budget(lv, C, lanes);
    if (budget > 0)
        while (want > 1 && per_vol * want > budget) want--;
    return want;
}

static void check_params(const fr3d_params *p, int Z, int Y, int X, int C)
{
    FR3D_CHECK(p != nullptr, "params is NULL");
    FR3D_CHECK(Z >= 1 && Y >= 1 && X >= 1, "volume dimensions must be >= 1");
    FR3D_CHECK(C >= 1 && C <= FR3D_MAX_CHANNELS, "1..8 channels are supported (FR3D_MAX_CHANNELS)");
    FR3D_CHECK(p->iterations >= 0 && p->update_lag >= 1, "iterations >= 0 and update_lag >= 1 required");
    FR3D_CHECK(p->eta > 0.0 && p->eta <= 1.0, "eta must be in (0,1]");  // eta == 1: `levels` solves at full size, like the reference
    FR3D_CHECK(p->levels >= 1, "levels must be >= 1");
}

static void get_displacement_dev(const fr3d_params *p, const float *fixed, const float *moving, int Z,
                                 int Y, int X, int C, const float *uvw_init, const float *weight,
                                 float *flow_out)
{
    ensure_init();
    check_params(p, Z, Y, X, C);
    FR3D_CHECK(fixed && moving && flow_out, "NULL volume pointer");
    Engine &e = g_eng;
    int min_level = p->min_level;
    std::vector<Level> lv = make_schedule(Z, Y, X, p->eta, p->levels, min_level);
    RefPyramid rp;
    for (auto &kv : g_eng2.bufs) kv.second.release();  // a single volume runs on lane 0: it may need the other lane's memory
    build_ref_pyramid(e, lv, fixed, weight, Z, Y, X, C, rp, "gd_");
    get_displacement_core(e, *p, lv, min_level, rp, 1, &moving, Z, Y, X, C, uvw_init, &flow_out);
    FR3D_HIP(hipStreamSynchronize(e.st));
}

template <typename TV, typename TF, typename TR = TV, typename TO = float>
static void warp_dev_t(const TV *vol, const TF *flow, const TR *ref, int Z, int Y, int X, int C, int order,
                       TO *out)
{
    Engine &e = *g_cur;
    for (int c = 0; c < C; c++) {
        if (order == 3) {
            warp_cubic_chan<TV, TF, TR, TO>(e, vol, C, c, flow + 0, flow + 1, flow + 2, 3, 1.0, 1.0, 1.0, ref, Z, Y,
                                            X, out, C, c);
        } else {
            Span sp(e, FR3D_K_WARP, 24.0 * (double)Z * Y * X, 1, (long long)Z * Y * X);
            launch_warp_linear<TV, TF, TR, TO>(e.st, vol, C, c, flow + 0, flow + 1, flow + 2, 3, ref, Z, Y, X, out,
                                               C, c);
        }
    }
}

// The executor's final compensation warp (parallelization/sequential_3d.py:153-170): the RAW volume in
// its own element type, flow fp32, reference fp32 or fp64, result in the raw type (OutCast, k_warp.hip).
template <typename TV>
static void warp_raw_t(const void *vol, const float *flow, const void *ref, int ref_dtype, int Z, int Y, int X, int C,
                       int order, void *out)
{
    if (ref_dtype == FR3D_F64)
        warp_dev_t<TV, float, double, TV>((const TV *)vol, flow, (const double *)ref, Z, Y, X, C, order, (TV *)out);
    else
        warp_dev_t<TV, float, float, TV>((const TV *)vol, flow, (const float *)ref, Z, Y, X, C, order, (TV *)out);
}

static void warp_raw(const void *vol, int raw_dtype, const float *flow, const void *ref, int ref_dtype, int Z, int Y,
                     int X, int C, int order, void *out)
{
    switch (raw_dtype) {
        case FR3D_F32: warp_raw_t<float>(vol, flow, ref, ref_dtype, Z, Y, X, C, order, out); break;
        case FR3D_F64: warp_raw_t<double>(vol, flow, ref, ref_dtype, Z, Y, X, C, order, out); break;
        case FR3D_U8: warp_raw_t<unsigned char>(vol, flow, ref, ref_dtype, Z, Y, X, C, order, out); break;
        case FR3D_U16: warp_raw_t<unsigned short>(vol, flow, ref, ref_dtype, Z, Y, X, C, order, out); break;
        case FR3D_I16: warp_raw_t<short>(vol, flow, ref, ref_dtype, Z, Y, X, C, order, out); break;
        default: throw Error("unknown dtype code");
    }
}

static void warp_dev(const void *vol, int vdt, const void *flow, int fdt, const void *ref, int Z, int Y,
                     int X, int C, int order, float *out)
{
    ensure_init();
    FR3D_CHECK(order == 1 || order == 3, "Unsupported interpolation method. Use 'linear' or 'cubic'.");
    FR3D_CHECK(Z >= 1 && Y >= 1 && X >= 1 && C >= 1, "bad warp shape");
    FR3D_CHECK(vol && flow && ref && out, "NULL pointer");
    if (vdt == FR3D_F32 && fdt == FR3D_F32)
        warp_dev_t<float, float>((const float *)vol, (const float *)flow, (const float *)ref, Z, Y, X, C, order, out);
    else if (vdt == FR3D_F32 && fdt == FR3D_F64)
        warp_dev_t<float, double>((const float *)vol, (const double *)flow, (const float *)ref, Z, Y, X, C, order, out);
    else if (vdt == FR3D_F64 && fdt == FR3D_F32)
        warp_dev_t<double, float>((const double *)vol, (const float *)flow, (const double *)ref, Z, Y, X, C, order, out);
    else if (vdt == FR3D_F64 && fdt == FR3D_F64)
        warp_dev_t<double, double>((const double *)vol, (const double *)flow, (const double *)ref, Z, Y, X, C, order, out);
    else
        throw Error("unknown dtype code");
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
}

static size_t dtype_size(int dt)
{
    switch (dt) {
        case FR3D_F32: return 4;
        case FR3D_F64: return 8;
        case FR3D_U8: return 1;
        case FR3D_U16: case FR3D_I16: return 2;
        default: throw Error("unknown dtype code");
    }
}

// The per-volume body of the executors for T device-resident volumes.  batch_raw / registered_out have
// the element type `raw_dtype`, ref_raw has `ref_dtype` (FR3D_F32 | FR3D_F64).  `rp_in`: reference and
// weight pyramids built by the caller (the host entry builds them once for all its staging windows).
static void process_batch_dev(const fr3d_params *p, const float *batch_proc, const void *batch_raw, int raw_dtype,
                              const float *ref_proc, const void *ref_raw, int ref_dtype, const float *w_init,
                              const float *weight, int T, int Z, int Y, int X, int C, int order,
                              float *flows_out, void *registered_out, fr3d_progress_fn progress, void *user,
                              const RefPyramid *rp_in = nullptr)
{
    ensure_init();
    check_params(p, Z, Y, X, C);
    FR3D_CHECK(order == 1 || order == 3, "Unsupported interpolation method. Use 'linear' or 'cubic'.");
    FR3D_CHECK(T >= 0, "T must be >= 0");
    FR3D_CHECK(ref_dtype == FR3D_F32 || ref_dtype == FR3D_F64, "reference_raw must be float32 or float64");
    const size_t esz = dtype_size(raw_dtype);
    Engine &e = g_eng;
    int min_level = p->min_level;
    std::vector<Level> lv = make_schedule(Z, Y, X, p->eta, p->levels, min_level);
    RefPyramid rp_own;
    // the fixed-reference pyramid and the weight pyramid are time-invariant: build once
    if (!rp_in) build_ref_pyramid(e, lv, ref_proc, weight, Z, Y, X, C, rp_own, "pb_");
    const RefPyramid &rp = rp_in ? *rp_in : rp_own;
    const size_t nv = (size_t)Z * Y * X;
    {
        const int m = resolve_mode(*p, C, Z, Y, X, lv);
        g_storage_bytes = m == 2 ? 8.0 : (m == 3 ? 16.0 / 3.0 : 4.0);
        g_window = use_window_sweep(*p, C, m == 2) && (m != 2 || C == 1);
    }
    g_fast_path = p->a_smooth == 1.0;
    int B = T > 0 ? pick_batch(T, lv, C) : 1;
    // two lanes: each takes lock-step batches of half the size, alternately (profiling brackets imply one lane: the
    // HIP-event spans of two lanes overlap and are not kernel times).  The budget is that of the layout that runs:
    // the volumes that fit beside TWO lanes' scratch, halved and rounded DOWN (an odd count must not become 2 x the
    // larger half); no room for one volume per lane -> one lane.
    bool two = g_lanes == 2 && g_eng2.inited && !e.prof && T >= 2 && B >= 2;
    if (two) {
        const int B2 = pick_batch(T, lv, C, 2) / 2;
        if (B2 >= 1) B = B2;
        else two = false;
    }
    if (!two)
        for (auto &kv : g_eng2.bufs) kv.second.release();  // one lane: the other lane's workspace is not reusable here
    const int reserve = g_batch_hint > 0 ? std::max(1, pick_batch(g_batch_hint, lv, C, two ? 2 : 1) / (two ? 2 : 1)) : B;
    // Buffers never shrink by themselves: slabs that an earlier call sized for more volumes per lane than this one
    // holds (a one-lane call before a two-lane call: every profiled pass) would sit beside the other lane's new slabs
    // although the budget above counts them as reusable.  Give them back first.
    for (Engine *en : {&g_eng, &g_eng2})
        if (en->slab_slots > std::max(B, reserve)) {
            en->release_slabs();
        }
    // the solver mode is settled here, once: the lanes then never look at the memory budget (and at each other's
    // workspace tables) while they run
    fr3d_params pr = *p;
    pr.solver_fp64 = resolve_mode(*p, C, Z, Y, X, lv);
    // T volumes in ceil(T/B) lock-step batches of (nearly) equal size: 10 volumes at B = 4 run as
    // 4+3+3, not 4+4+2 (the shared launches amortise best over evenly filled batches)
    struct Chunk { int t0, nb; };
    std::vector<Chunk> chunks;
    {
        int chunks_left = T > 0 ? cdiv(T, B) : 0;
        for (int t0 = 0, nb = 0; t0 < T; t0 += nb, chunks_left--) {
            nb = cdiv(T - t0, chunks_left);
            chunks.push_back({t0, nb});
        }
    }
    auto run_chunk = [&](Engine &lane, const Chunk &c) {
        std::vector<const float *> mov(c.nb);
        std::vector<float *> fl(c.nb);
        for (int b = 0; b < c.nb; b++) {
            mov[b] = batch_proc + (size_t)(c.t0 + b) * nv * C;
            fl[b] = flows_out + (size_t)(c.t0 + b) * nv * 3;
        }
        get_displacement_core(lane, pr, lv, min_level, rp, c.nb, mov.data(), Z, Y, X, C, w_init, fl.data(), reserve);
        g_cur = &lane;
        for (int b = 0; b < c.nb; b++) {
            const size_t o = (size_t)(c.t0 + b) * nv * C * esz;
            warp_raw((const char *)batch_raw + o, raw_dtype, fl[b], ref_raw, ref_dtype, Z, Y, X, C, order,
                     (char *)registered_out + o);
        }
        g_cur = &g_eng;
    };
    if (!two) {
        for (const Chunk &c : chunks) {
            run_chunk(e, c);
            if (progress) {
                FR3D_HIP(hipStreamSynchronize(e.st));
                for (int b = 0; b < c.nb; b++) progress(1, user);
            }
        }
        FR3D_HIP(hipStreamSynchronize(e.st));
        return;
    }
    // Two lanes.  Lane 1 starts behind an event that covers everything enqueued on lane 0 so far (the caller's inputs,
    // the reference pyramid) and is fed by a host thread of its own: a batch is thousands of launches, and one thread
    // feeding both streams would block on the first stream's full queue while the second runs dry.  Each thread waits
    // for its batches in order and reports their volumes (the callback is serialised; it may be called from either thread).
    hipEvent_t ev_start = nullptr;
    FR3D_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    std::exception_ptr err0, err1;
    try {
        FR3D_HIP(hipEventRecord(ev_start, e.st));
        FR3D_HIP(hipStreamWaitEvent(g_eng2.st, ev_start, 0));
    } catch (...) {
        (void)hipEventDestroy(ev_start);
        throw;
    }
    const int device = e.device;
    // The progress callback runs on the CALLER's thread only (the reference's executors call it there, and the caller
    // holds this library's lock: a callback that re-enters the library from lane 1's thread would wait for it forever).
    // Lane 1 posts the volumes of its finished batches; the caller's thread delivers them whenever it reports its own
    // and while it waits for lane 1 at the end.
    std::mutex progress_mu;
    std::condition_variable progress_cv;
    int posted = 0;          // volumes finished on lane 1, not yet delivered
    bool lane1_done = false;
    auto deliver_posted = [&]() {  // caller's thread
        int n;
        {
            std::lock_guard<std::mutex> lk(progress_mu);
            n = posted;
            posted = 0;
        }
        if (progress)
            for (int b = 0; b < n; b++) progress(1, user);
    };
    auto feed = [&](Engine &lane, int parity, std::exception_ptr &err) {
        std::vector<std::pair<hipEvent_t, int>> pending;  // completion event and volume count of this lane's batches
        auto report = [&](size_t upto) {  // wait for this lane's batches [reported, upto) and report them
            for (; !pending.empty() && upto > 0; upto--) {
                FR3D_HIP(hipEventSynchronize(pending.front().first));
                if (progress) {
                    if (parity == 0) {
                        deliver_posted();
                        for (int b = 0; b < pending.front().second; b++) progress(1, user);
                    } else {
                        std::lock_guard<std::mutex> lk(progress_mu);
                        posted += pending.front().second;
                        progress_cv.notify_one();
                    }
                }
                (void)hipEventDestroy(pending.front().first);
                pending.erase(pending.begin());
            }
        };
        try {
            FR3D_HIP(hipSetDevice(device));
            for (size_t i = 0; i < chunks.size(); i++) {
                if ((int)(i & 1) != parity) continue;
                run_chunk(lane, chunks[i]);
                hipEvent_t ev;
                FR3D_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                pending.emplace_back(ev, chunks[i].nb);
                FR3D_HIP(hipEventRecord(ev, lane.st));
                // with the next batch already in the queue, wait for the one before it: progress arrives one batch late
                // and the host never runs more than two batches ahead of the device
                if (pending.size() > 1) report(pending.size() - 1);
            }
            report(pending.size());
            FR3D_HIP(hipStreamSynchronize(lane.st));
        } catch (...) {
            err = std::current_exception();
            (void)hipStreamSynchronize(lane.st);
            for (auto &pe : pending) (void)hipEventDestroy(pe.first);
        }
    };
    std::thread other([&]() {
        feed(g_eng2, 1, err1);
        std::lock_guard<std::mutex> lk(progress_mu);
        lane1_done = true;
        progress_cv.notify_one();
    });
    feed(e, 0, err0);
    for (;;) {  // lane 0 is through: hand on lane 1's volumes as they finish
        std::unique_lock<std::mutex> lk(progress_mu);
        progress_cv.wait(lk, [&]() { return posted > 0 || lane1_done; });
        const bool done = lane1_done;
        lk.unlock();
        deliver_posted();
        if (done) break;
    }
    other.join();
    (void)hipEventDestroy(ev_start);
    if (err0) std::rethrow_exception(err0);
    if (err1) std::rethrow_exception(err1);
}

// ---- f-4 update_reference (compensate_recording_3D.py:395-429) -------------------------------------
// new reference_proc = per-channel mean of the last min(100, T) batch_proc volumes warped by their flows.
template <typename TV, typename TR>
static void update_reference_t(Engine &e, const TV *batch_proc, const float *flows, const TR *ref_proc, int T, int Z,
                               int Y, int X, int C, int order, double *new_ref)
{
    const size_t nv = (size_t)Z * Y * X;
    const int n_ref = std::min(100, T);
    if (n_ref < 1) return;
    const int start = T - n_ref;
    float *tmp = e.f32("ur_tmp", nv);
    double *acc = e.f64("ur_acc", nv);
    for (int c = 0; c < C; c++) {
        for (int t = 0; t < n_ref; t++) {
            const TV *vol = batch_proc + (size_t)(start + t) * nv * C;
            const float *fl = flows + (size_t)(start + t) * nv * 3;
            if (order == 3) {
                warp_cubic_chan<TV, float, TR, float>(e, vol, C, c, fl + 0, fl + 1, fl + 2, 3, 1.0, 1.0, 1.0, ref_proc, Z,
                                                      Y, X, tmp, 1, 0);
            } else {
                // the linear launcher indexes vol and ref with the same channel stride / offset (C, c); tmp is planar
                Span sp(e, FR3D_K_WARP, 24.0 * (double)nv, 1, (long long)nv);
                launch_warp_linear<TV, float, TR, float>(e.st, vol, C, c, fl + 0, fl + 1, fl + 2, 3, ref_proc, Z, Y, X, tmp, 1,
                                                         0);
            }
            launch_accum_f64(e.st, acc, tmp, (long long)nv, t == 0);
        }
        launch_mean_store(e.st, acc, (long long)nv, C, c, (double)n_ref, new_ref);
    }
}

static void update_reference_dev(const void *batch_proc, int proc_dtype, const float *flows, const void *ref_proc,
                                 int ref_dtype, int T, int Z, int Y, int X, int C, int order, double *new_ref)
{
    ensure_init();
    FR3D_CHECK(order == 1 || order == 3, "Unsupported interpolation method. Use 'linear' or 'cubic'.");
    FR3D_CHECK(T >= 0 && Z >= 1 && Y >= 1 && X >= 1 && C >= 1, "bad update_reference shape");
    FR3D_CHECK((T == 0 || (batch_proc && flows)) && ref_proc && new_ref, "NULL pointer");
    FR3D_CHECK((proc_dtype == FR3D_F32 || proc_dtype == FR3D_F64) && (ref_dtype == FR3D_F32 || ref_dtype == FR3D_F64),
               "batch_proc / reference_proc must be float32 or float64");
    Engine &e = g_eng;
    if (proc_dtype == FR3D_F64 && ref_dtype == FR3D_F64)
        update_reference_t<double, double>(e, (const double *)batch_proc, flows, (const double *)ref_proc, T, Z, Y, X, C, order, new_ref);
    else if (proc_dtype == FR3D_F64)
        update_reference_t<double, float>(e, (const double *)batch_proc, flows, (const float *)ref_proc, T, Z, Y, X, C, order, new_ref);
    else if (ref_dtype == FR3D_F64)
        update_reference_t<float, double>(e, (const float *)batch_proc, flows, (const double *)ref_proc, T, Z, Y, X, C, order, new_ref);
    else
        update_reference_t<float, float>(e, (const float *)batch_proc, flows, (const float *)ref_proc, T, Z, Y, X, C, order, new_ref);
    FR3D_HIP(hipStreamSynchronize(e.st));
}

// ---- f-1 preprocessing ----------------------------------------------------------------------------
// scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, int(truncate*sigma+0.5)), fp64
static std::vector<double> gaussian_kernel(double sigma, double truncate, int &radius)
{
    radius = (int)(truncate * sigma + 0.5);
    const int n = 2 * radius + 1;
    std::vector<double> w(n);
    const double sigma2 = sigma * sigma;
    for (int i = 0; i < n; i++) {
        const double x = (double)(i - radius);
        w[i] = std::exp(-0.5 / sigma2 * (x * x));
    }
    double s;
    if (n < 8) {
        s = 0.0;
        for (int i = 0; i < n; i++) s += w[i];
    } else {  // numpy pairwise add.reduce (n <= 128 in practice)
        double r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = w[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += w[i + j];
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) s += w[i];
    }
    for (auto &v : w) v = v / s;
    return w;
}

const double *Engine::gauss_kernel(double sigma, double truncate, int &radius)
{
    long long a, b;
    std::memcpy(&a, &sigma, 8);
    std::memcpy(&b, &truncate, 8);
    auto key = std::make_pair(a, b);
    auto it = gkernels.find(key);
    if (it == gkernels.end()) {
        int r = 0;
        std::vector<double> w = sigma > 0.0 ? gaussian_kernel(sigma, truncate, r) : std::vector<double>(1, 1.0);
        double *d = nullptr;
        FR3D_HIP(hipMalloc((void **)&d, w.size() * sizeof(double)));
        FR3D_HIP(hipMemcpy(d, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));  // synchronous, once
        it = gkernels.emplace(key, std::make_pair(d, r)).first;
    }
    radius = it->second.second;
    return it->second.first;
}

template <typename TIN>
static void preprocess_t(Engine &e, const TIN *frames, int T, int Z, int Y, int X, int C, const double *nmin,
                         const double *nden, const double *sigma, double truncate, int mode, void *out, int out_dtype)
{
    const long long n = (long long)T * Z * Y * X;
    const bool fastk = mode == FR3D_BOUNDARY_REFLECT;  // the radius-4 kernels are written for the pipeline's mode
    double *bufA = e.f64("pp_a", (size_t)n);
    double *bufB = e.f64("pp_b", (size_t)n);
    Span sp(e, FR3D_K_PREPROC, 0, 0, 0);
    for (int c = 0; c < C; c++) {
        const double *sg = sigma + 4 * c;            // sx, sy, sz, st
        const double ax_sigma[4] = {sg[3], sg[2], sg[1], sg[0]};  // array axes T, Z, Y, X
        // the passes scipy really runs: axes with sigma > 0 whose kernel has more than one tap (a kernel [1.0]
        // multiplies by exactly 1)
        struct Pass { int axis, radius; const double *w; };
        Pass ps[4];
        int np = 0;
        for (int axis = 0; axis < 4; axis++) {
            if (!(ax_sigma[axis] > 1e-15)) continue; // scipy skips these axes
            int radius;
            const double *dw = e.gauss_kernel(ax_sigma[axis], truncate, radius);
            if (radius == 0) continue;
            ps[np++] = {axis, radius, dw};
        }
        const double *cur = nullptr;                 // nullptr: still reading the caller's frames
        double *dst = bufA;
        int launches = 0;
        double bytes = 0.0;
        bool stored = false;
        for (int p = 0; p < np; p++) {
            const bool first = p == 0, last = p == np - 1;
            bool done = false;
            // fast kernels (radius 4): the last pass writes the caller's array itself
            if (!fastk) {
            } else if (last) {
                if (out_dtype == FR3D_F64)
                    done = first ? launch_gauss_pass4<TIN, double, true>(e.st, frames, C, c, nmin[c], nden[c], T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, (double *)out, C, c)
                                 : launch_gauss_pass4<double, double, false>(e.st, cur, 1, 0, 0.0, 1.0, T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, (double *)out, C, c);
                else
                    done = first ? launch_gauss_pass4<TIN, float, true>(e.st, frames, C, c, nmin[c], nden[c], T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, (float *)out, C, c)
                                 : launch_gauss_pass4<double, float, false>(e.st, cur, 1, 0, 0.0, 1.0, T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, (float *)out, C, c);
                if (done) {
                    stored = true;
                    bytes += (double)n * ((first ? sizeof(TIN) : 8.0) + (out_dtype == FR3D_F64 ? 8.0 : 4.0));
                }
            } else {
                done = first ? launch_gauss_pass4<TIN, double, true>(e.st, frames, C, c, nmin[c], nden[c], T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, dst, 1, 0)
                             : launch_gauss_pass4<double, double, false>(e.st, cur, 1, 0, 0.0, 1.0, T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, dst, 1, 0);
                if (done) bytes += (double)n * ((first ? sizeof(TIN) : 8.0) + 8.0);
            }
            if (!done) {  // any other radius / a very short axis: the general kernel
                if (first) launch_gauss_pass<TIN>(e.st, frames, C, c, nmin[c], nden[c], T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, mode, dst);
                else launch_gauss_pass<double>(e.st, cur, 1, 0, 0.0, 1.0, T, Z, Y, X, ps[p].axis, ps[p].w, ps[p].radius, mode, dst);
                bytes += (double)n * ((first ? sizeof(TIN) : 8.0) + 8.0);
            }
            launches++;
            if (!stored) {
                cur = dst;
                dst = (dst == bufA) ? bufB : bufA;
            }
        }
        if (np == 0) {  // no filtering at all: normalisation only (a radius-0 pass)
            int r0;
            const double *dw = e.gauss_kernel(0.0, truncate, r0);  // radius 0: the single tap 1.0
            launch_gauss_pass<TIN>(e.st, frames, C, c, nmin[c], nden[c], T, Z, Y, X, 3, dw, 0, mode, dst);
            cur = dst;
            launches++;
            bytes += (double)n * (sizeof(TIN) + 8.0);
        }
        if (!stored) {
            if (out_dtype == FR3D_F64) launch_store_channel<double>(e.st, cur, n, C, c, (double *)out);
            else launch_store_channel<float>(e.st, cur, n, C, c, (float *)out);
            launches++;
            bytes += (double)n * (8.0 + (out_dtype == FR3D_F64 ? 8.0 : 4.0));
        }
        // algorithmic bytes as for the resampler: per pass one read and one write of the volume in the
        // pass's own element types
        sp.add(bytes, launches, n);
    }
}

static void preprocess_dev(const void *frames, int dtype, int T, int Z, int Y, int X, int C, const double *nmin,
                           const double *nden, const double *sigma, double truncate, void *out, int out_dtype,
                           int mode = FR3D_BOUNDARY_REFLECT)
{
    FR3D_CHECK(mode >= FR3D_BOUNDARY_REFLECT && mode <= FR3D_BOUNDARY_WRAP, "unknown boundary mode");
    ensure_init();
    FR3D_CHECK(frames && nmin && nden && sigma && out, "NULL pointer");
    FR3D_CHECK(T >= 0 && Z >= 1 && Y >= 1 && X >= 1 && C >= 1 && C <= FR3D_MAX_CHANNELS, "bad preprocess shape");
    FR3D_CHECK(out_dtype == FR3D_F32 || out_dtype == FR3D_F64, "out_dtype must be FR3D_F32 or FR3D_F64");
    FR3D_CHECK(truncate > 0.0, "truncate must be positive");
    for (int c = 0; c < C; c++) FR3D_CHECK(nden[c] != 0.0, "normalisation denominator is zero");
    Engine &e = g_eng;
    switch (dtype) {
        case FR3D_F32: preprocess_t<float>(e, (const float *)frames, T, Z, Y, X, C, nmin, nden, sigma, truncate, mode, out, out_dtype); break;
        case FR3D_F64: preprocess_t<double>(e, (const double *)frames, T, Z, Y, X, C, nmin, nden, sigma, truncate, mode, out, out_dtype); break;
        case FR3D_U8: preprocess_t<unsigned char>(e, (const unsigned char *)frames, T, Z, Y, X, C, nmin, nden, sigma, truncate, mode, out, out_dtype); break;
        case FR3D_U16: preprocess_t<unsigned short>(e, (const unsigned short *)frames, T, Z, Y, X, C, nmin, nden, sigma, truncate, mode, out, out_dtype); break;
        case FR3D_I16: preprocess_t<short>(e, (const short *)frames, T, Z, Y, X, C, nmin, nden, sigma, truncate, mode, out, out_dtype); break;
        default: throw Error("unknown dtype code");
    }
    FR3D_HIP(hipStreamSynchronize(e.st));
}

// host staging helper: device copies of caller (host) arrays for the duration of one entry point.  The
// buffers come from the engine's pool ("stg0", "stg1", ... grown on demand, kept for the next call), so a
// caller that registers volume after volume through fr3d_get_displacement / fr3d_warp does not pay a
// hipMalloc + hipFree pair per array and call.
struct Staged {
    int next = 0;
    void *slot(size_t bytes) { return g_eng.bufs["stg" + std::to_string(next++)].ensure(bytes ? bytes : 1); }
    void *up(const void *host, size_t bytes)
    {
        if (!host) return nullptr;
        void *d = slot(bytes);
        FR3D_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
        return d;
    }
    void *alloc(size_t bytes) { return slot(bytes); }
};

}  // namespace fr3d

// =============================================================================================
// C ABI
// =============================================================================================
using namespace fr3d;

#define FR3D_TRY try { std::lock_guard<std::recursive_mutex> lk__(g_mu);
#define FR3D_CATCH                                                                              \
    }                                                                                           \
    catch (const std::exception &ex) { g_err = ex.what(); return 1; }                           \
    catch (...) { g_err = "unknown error"; return 1; }                                          \
    return 0;

extern "C" {

const char *fr3d_last_error(void) { return g_err.c_str(); }
const char *fr3d_version(void) { return "flowreg3d_amd 0.1 (gfx950)"; }

const char *fr3d_device_info(void)
{
    static std::string info;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    info.clear();
    if (!g_eng.inited) return info.c_str();
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, g_eng.device) != hipSuccess) return info.c_str();
    char buf[384];
    snprintf(buf, sizeof(buf), "%s (%s); %d CUs; core %d MHz; memory %d MHz, %d-bit; %.0f GiB", pr.name,
             pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000, pr.memoryClockRate / 1000,
             pr.memoryBusWidth, (double)pr.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    info = buf;
    return info.c_str();
}

int fr3d_last_solver_mode(void) { return g_last_mode; }
int fr3d_last_solver_fallback(void) { return g_last_fallback; }

int fr3d_set_batch(int nvol)
{
    g_batch_hint = nvol > 0 ? nvol : 0;
    return 0;
}

int fr3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fr3d_init(int device)
{
    FR3D_TRY
    if (g_eng.inited && g_eng.device == device) return 0;
    if (g_eng.inited) throw Error("engine already initialised on another device; call fr3d_shutdown first");
    int n = 0;
    FR3D_HIP(hipGetDeviceCount(&n));
    FR3D_CHECK(n > 0, "no HIP device visible");
    FR3D_CHECK(device >= 0 && device < n, "device index out of range");
    FR3D_HIP(hipSetDevice(device));
    for (Engine *en : {&g_eng, &g_eng2}) {
        FR3D_HIP(hipStreamCreateWithFlags(&en->st, hipStreamNonBlocking));
        en->device = device;
        std::memset(en->acc, 0, sizeof(en->acc));
        en->inited = true;
    }
    if (const char *v = std::getenv("FR3D_LANES")) g_lanes = std::atoi(v) == 1 ? 1 : 2;
    FR3D_CATCH
}

static void release_engine(Engine &en)
{
    if (!en.inited) return;
    (void)hipStreamSynchronize(en.st);
    for (auto &kv : en.bufs) kv.second.release();
    en.bufs.clear();
    for (auto &kv : en.tables) {
        (void)hipFree(kv.second.idx);
        (void)hipFree(kv.second.wt);
    }
    en.tables.clear();
    for (auto &kv : en.scheds) free_sor_schedule(kv.second);
    en.scheds.clear();
    for (auto &kv : en.chain_scheds) free_sor_chain_schedule(kv.second);
    en.chain_scheds.clear();
    for (auto &kv : en.win_scheds) free_win_schedule(kv.second);
    en.win_scheds.clear();
    for (auto &kv : en.gkernels) (void)hipFree(kv.second.first);
    en.gkernels.clear();
    for (auto &kv : en.compacts) {
        (void)hipFree(kv.second.pb);
        (void)hipFree(kv.second.cp);
    }
    en.compacts.clear();
    for (auto &s : en.spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    en.spans.clear();
    for (auto ev : en.ev_pool) (void)hipEventDestroy(ev);
    en.ev_pool.clear();
    (void)hipStreamDestroy(en.st);
    en.st = nullptr;
    en.inited = false;
    en.device = -1;
}

void fr3d_shutdown(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    release_engine(g_eng2);
    release_engine(g_eng);
    g_cur = &g_eng;
}

int fr3d_set_lanes(int lanes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int prev = g_lanes;
    if (lanes == 1 || lanes == 2) g_lanes = lanes;
    return prev;
}

int fr3d_get_displacement_dev(const fr3d_params *p, const float *fixed, const float *moving, int Z, int Y,
                              int X, int C, const float *uvw_init, const float *weight, float *flow_out)
{
    FR3D_TRY
    get_displacement_dev(p, fixed, moving, Z, Y, X, C, uvw_init, weight, flow_out);
    FR3D_CATCH
}

int fr3d_get_displacement(const fr3d_params *p, const float *fixed, const float *moving, int Z, int Y, int X,
                          int C, const float *uvw_init, const float *weight, float *flow_out)
{
    FR3D_TRY
    ensure_init();
    check_params(p, Z, Y, X, C);
    FR3D_CHECK(fixed && moving && flow_out, "NULL volume pointer");
    const size_t nv = (size_t)Z * Y * X;
    Staged s;
    const float *df = (const float *)s.up(fixed, nv * C * 4);
    const float *dm = (const float *)s.up(moving, nv * C * 4);
    const float *du = (const float *)s.up(uvw_init, nv * 3 * 4);
    const float *dw = (const float *)s.up(weight, nv * C * 4);
    float *dout = (float *)s.alloc(nv * 3 * 4);
    get_displacement_dev(p, df, dm, Z, Y, X, C, du, dw, dout);
    FR3D_HIP(hipMemcpy(flow_out, dout, nv * 3 * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_get_displacement_verify(const fr3d_params *p, const float *fixed, const float *moving, int Z, int Y, int X,
                                 int C, const float *uvw_init, const float *weight, double *flow_out)
{
    FR3D_TRY
    ensure_init();
    check_params(p, Z, Y, X, C);
    FR3D_CHECK(fixed && moving && flow_out, "NULL volume pointer");
    Engine &e = g_eng;
    const size_t nv = (size_t)Z * Y * X;
    Staged s;
    const float *df = (const float *)s.up(fixed, nv * C * 4);
    const float *dm = (const float *)s.up(moving, nv * C * 4);
    const float *du = (const float *)s.up(uvw_init, nv * 3 * 4);
    const float *dw = (const float *)s.up(weight, nv * C * 4);
    double *dout = (double *)s.alloc(nv * 3 * 8);
    int min_level = p->min_level;
    std::vector<Level> lv = make_schedule(Z, Y, X, p->eta, p->levels, min_level);
    RefPyramid rp;
    build_ref_pyramid(e, lv, df, dw, Z, Y, X, C, rp, "gd_");
    const bool prof = e.prof;
    e.prof = false;  // not a timed path
    try {
        get_displacement_verify(e, *p, lv, min_level, rp, dm, Z, Y, X, C, du, dout);
    } catch (...) {
        e.prof = prof;
        throw;
    }
    e.prof = prof;
    FR3D_HIP(hipStreamSynchronize(e.st));
    FR3D_HIP(hipMemcpy(flow_out, dout, nv * 3 * 8, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_spline_coefficients(const float *vol, int Z, int Y, int X, double *coef_out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(vol && coef_out && Z > 0 && Y > 0 && X > 0, "bad arguments");
    FR3D_CHECK(prefilter_compact_ok(Z, Y, X), "spline coefficient hook: every axis must be at least 41 voxels long");
    const size_t n = (size_t)Z * Y * X, np = (size_t)(Z + 4) * (Y + 4) * (X + 4);
    Staged s;
    const float *dv = (const float *)s.up(vol, n * 4);
    double *coef = (double *)s.alloc(np * 8);
    double *tmp = (double *)s.alloc((size_t)(Z + 4) * (Y + 4) * X * 8);
    launch_prefilter3_compact<float>(g_eng.st, dv, 1, 0, Z, Y, X, coef, tmp);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(coef_out, coef, np * 8, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_motion_tensor_f64(const float *f1, const float *f2, int Z, int Y, int X, double hz, double hy, double hx,
                           double *J_out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(f1 && f2 && J_out && Z > 0 && Y > 0 && X > 0, "bad arguments");
    const size_t n = (size_t)Z * Y * X;
    Staged s;
    const float *d1 = (const float *)s.up(f1, n * 4);
    const float *d2 = (const float *)s.up(f2, n * 4);
    double *dj = (double *)s.alloc(n * 10 * 8);
    double *Jo[10];
    for (int q = 0; q < 10; q++) Jo[q] = dj + (size_t)q * n;
    launch_motion_tensor<double, double>(g_eng.st, d1, d2, Z, Y, X, hz, hy, hx, Jo, (double *)nullptr, 0, nullptr);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(J_out, dj, n * 10 * 8, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

// level_solver on tensor ENTRIES in the reference's own arithmetic (k_verify.hip): the sweep of the verification mode,
// a_smooth == 1 or not; uvw fp32 or fp64
static void level_solve_entries(const double *J, const float *weight, const void *uvw, bool uvw_f64, int Z, int Y, int X,
                                int C, const double *alpha3, int iterations, int update_lag, const double *a_data,
                                double a_smooth, double hx, double hy, double hz, double *duvw_out,
                                const double *uvw_padded = nullptr)
{
    ensure_init();
    FR3D_CHECK(J && weight && uvw && alpha3 && a_data && duvw_out, "NULL pointer");
    FR3D_CHECK(Z > 0 && Y > 0 && X > 0 && C >= 1 && C <= FR3D_MAX_CHANNELS, "bad solver shape");
    FR3D_CHECK(iterations >= 0 && update_lag >= 1, "iterations >= 0 and update_lag >= 1 required");
    Engine &e = g_eng;
    const size_t n = (size_t)Z * Y * X;
    const Skew sk = e.compact_skew(Z, Y, X);
    const size_t ns = (size_t)sk.total;
    Staged s;
    const double *dJ = (const double *)s.up(J, n * 10 * C * 8);   // (C, 10, Z, Y, X)
    const float *dW = (const float *)s.up(weight, n * C * 4);      // (C, Z, Y, X)
    const void *dU = s.up(uvw, n * 3 * (uvw_f64 ? 8 : 4));         // (3, Z, Y, X)
    double *Jrec = (double *)s.alloc(ns * 10 * C * 8);
    float *wrec = (float *)s.alloc(ns * C * 4);
    double *psi = (double *)s.alloc(ns * C * 8);
    double *Urec = (double *)s.alloc(ns * 3 * 8);
    double *Drec = (double *)s.alloc(ns * 3 * 8);
    double *dn = (double *)s.alloc(n * 3 * 8);
    VerifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.sk = sk;
    a.C = C;
    a.update_lag = update_lag;
    for (int c = 0; c < C; c++) {
        launch_skew_pack<double, double>(e.st, dJ + (size_t)c * 10 * n, (long long)n, Jrec + (size_t)c * 10 * ns, 10, sk);
        launch_skew_pack<float, float>(e.st, dW + (size_t)c * n, 0, wrec + (size_t)c * ns, 1, sk);
        a.J[c] = Jrec + (size_t)c * 10 * ns;
        a.w[c] = wrec + (size_t)c * ns;
        a.psi[c] = psi + (size_t)c * ns;
        a.a_data[c] = a_data[c];
    }
    if (uvw_f64) launch_skew_pack<double, double>(e.st, (const double *)dU, (long long)n, Urec, 3, sk);
    else launch_skew_pack<float, double>(e.st, (const float *)dU, (long long)n, Urec, 3, sk);
    FR3D_HIP(hipMemsetAsync(Drec, 0, ns * 3 * 8, e.st));
    a.U = Urec;
    a.D = Drec;
    a.Ug = (const double *)s.up(uvw_padded, (size_t)(Z + 2) * (Y + 2) * (X + 2) * 3 * 8);  // nullptr stays nullptr
    a.ax = alpha3[0] / (hx * hx);
    a.ay = alpha3[1] / (hy * hy);
    a.az = alpha3[2] / (hz * hz);
    if (a_smooth == 1.0) {
        launch_sor_verify(e.st, a, e.chain_sched(sk, iterations));
    } else {
        // one iteration at a time, as in get_displacement_verify: psi_smooth of t from the increments of t-1 (interior)
        // and t-2 (ghost ring), then one lexicographic sweep
        double *Dm2 = (double *)s.alloc(ns * 3 * 8);
        double *Ps = (double *)s.alloc((size_t)(Z + 2) * (Y + 2) * (X + 2) * 8);
        FR3D_HIP(hipMemsetAsync(Dm2, 0, ns * 3 * 8, e.st));
        a.Ps = Ps;
        const SorChainSched &one = e.chain_sched(sk, 1);
        for (int it = 0; it < iterations; it++) {
            launch_psi_smooth_verify(e.st, sk, Urec, Drec, Dm2, a_smooth, hx, hy, hz, Ps, a.Ug);
            FR3D_HIP(hipMemcpyAsync(Dm2, Drec, ns * 3 * 8, hipMemcpyDeviceToDevice, e.st));
            a.t_base = it;
            launch_sor_verify(e.st, a, one);
        }
    }
    launch_unskew_unpack<double, double>(e.st, Drec, dn, (long long)n, 3, sk);
    FR3D_HIP(hipStreamSynchronize(e.st));
    FR3D_HIP(hipMemcpy(duvw_out, dn, n * 3 * 8, hipMemcpyDeviceToHost));
}

int fr3d_level_solve_verify(const double *J, const float *weight, const float *uvw, int Z, int Y, int X, int C,
                            const double *alpha3, int iterations, int update_lag, const double *a_data, double hx,
                            double hy, double hz, double *duvw_out)
{
    FR3D_TRY
    level_solve_entries(J, weight, uvw, false, Z, Y, X, C, alpha3, iterations, update_lag, a_data, 1.0, hx, hy, hz, duvw_out);
    FR3D_CATCH
}

int fr3d_level_solve_tensor(const double *J, const float *weight, const double *uvw, const double *uvw_padded, int Z, int Y,
                            int X, int C, const double *alpha3, int iterations, int update_lag, const double *a_data,
                            double a_smooth, double hx, double hy, double hz, double *duvw_out)
{
    FR3D_TRY
    level_solve_entries(J, weight, uvw, true, Z, Y, X, C, alpha3, iterations, update_lag, a_data, a_smooth, hx, hy, hz, duvw_out,
                        uvw_padded);
    FR3D_CATCH
}

int fr3d_portable_pow(const double *x, const double *y, size_t n, double *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(x && y && out, "NULL pointer");
    Staged s;
    const double *dx = (const double *)s.up(x, n * 8);
    const double *dy = (const double *)s.up(y, n * 8);
    double *dout = (double *)s.alloc(n * 8);
    launch_ppow(g_eng.st, dx, dy, (long long)n, dout);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_warp_dev(const void *vol, int vol_dtype, const void *flow, int flow_dtype, const void *ref, int Z,
                  int Y, int X, int C, int order, float *out)
{
    FR3D_TRY
    warp_dev(vol, vol_dtype, flow, flow_dtype, ref, Z, Y, X, C, order, out);
    FR3D_CATCH
}

int fr3d_warp(const void *vol, int vol_dtype, const void *flow, int flow_dtype, const void *ref, int Z, int Y,
              int X, int C, int order, float *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(order == 1 || order == 3, "Unsupported interpolation method. Use 'linear' or 'cubic'.");
    FR3D_CHECK(Z >= 1 && Y >= 1 && X >= 1 && C >= 1, "bad warp shape");
    FR3D_CHECK(vol && flow && ref && out, "NULL pointer");
    FR3D_CHECK((vol_dtype == FR3D_F32 || vol_dtype == FR3D_F64) && (flow_dtype == FR3D_F32 || flow_dtype == FR3D_F64),
               "unknown dtype code");
    const size_t nv = (size_t)Z * Y * X;
    const size_t vb = vol_dtype == FR3D_F64 ? 8 : 4, fb = flow_dtype == FR3D_F64 ? 8 : 4;
    Staged s;
    const void *dv = s.up(vol, nv * C * vb);
    const void *dr = s.up(ref, nv * C * vb);
    const void *dfl = s.up(flow, nv * 3 * fb);
    float *dout = (float *)s.alloc(nv * C * 4);
    warp_dev(dv, vol_dtype, dfl, flow_dtype, dr, Z, Y, X, C, order, dout);
    FR3D_HIP(hipMemcpy(out, dout, nv * C * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_process_batch_raw_dev(const fr3d_params *p, const float *batch_proc, const void *batch_raw, int raw_dtype,
                               const float *ref_proc, const void *ref_raw, int ref_dtype, const float *w_init,
                               const float *weight, int T, int Z, int Y, int X, int C, int order, float *flows_out,
                               void *registered_out, fr3d_progress_fn progress, void *user)
{
    FR3D_TRY
    FR3D_CHECK(batch_proc && batch_raw && ref_proc && ref_raw && flows_out && registered_out, "NULL pointer");
    process_batch_dev(p, batch_proc, batch_raw, raw_dtype, ref_proc, ref_raw, ref_dtype, w_init, weight, T, Z, Y, X, C,
                      order, flows_out, registered_out, progress, user);
    FR3D_CATCH
}

int fr3d_process_batch_dev(const fr3d_params *p, const float *batch_proc, const float *batch_raw,
                           const float *ref_proc, const float *ref_raw, const float *w_init,
                           const float *weight, int T, int Z, int Y, int X, int C, int order, float *flows_out,
                           float *registered_out, fr3d_progress_fn progress, void *user)
{
    return fr3d_process_batch_raw_dev(p, batch_proc, batch_raw, FR3D_F32, ref_proc, ref_raw, FR3D_F32, w_init, weight,
                                      T, Z, Y, X, C, order, flows_out, registered_out, progress, user);
}

int fr3d_process_batch_raw(const fr3d_params *p, const float *batch_proc, const void *batch_raw, int raw_dtype,
                           const float *ref_proc, const void *ref_raw, int ref_dtype, const float *w_init,
                           const float *weight, int T, int Z, int Y, int X, int C, int order, float *flows_out,
                           void *registered_out, fr3d_progress_fn progress, void *user)
{
    FR3D_TRY
    ensure_init();
    check_params(p, Z, Y, X, C);
    FR3D_CHECK(batch_proc && batch_raw && ref_proc && ref_raw && flows_out && registered_out, "NULL pointer");
    FR3D_CHECK(T >= 0, "T must be >= 0");
    FR3D_CHECK(ref_dtype == FR3D_F32 || ref_dtype == FR3D_F64, "reference_raw must be float32 or float64");
    const size_t nv = (size_t)Z * Y * X;
    const size_t rsz = dtype_size(raw_dtype), fsz = dtype_size(ref_dtype);
#ifdef FR3D_EXPERIMENTS  // FR3D_HOST_TRACE=1: milliseconds since entry of every phase of this call, on stderr
    const bool trace = getenv("FR3D_HOST_TRACE") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    std::mutex trace_mu;
    auto mark = [&](const char *what, int k) {
        if (!trace) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
        std::lock_guard<std::mutex> lk(trace_mu);
        fprintf(stderr, "[host trace] %8.2f ms  %s %d\n", ms, what, k);
    };
#else
    auto mark = [](const char *, int) {};
#endif
    Staged s;
    const float *drp = (const float *)s.up(ref_proc, nv * C * 4);
    const void *drr = s.up(ref_raw, nv * C * fsz);
    const float *dwi = (const float *)s.up(w_init, nv * 3 * 4);
    const float *dwt = (const float *)s.up(weight, nv * C * 4);
    // The series passes through the device in windows of one lock-step batch, double-buffered: a
    // copier thread uploads window k+2 and downloads window k on its own stream while the engine
    // stream computes window k+1, so the PCIe traffic (and the page faults of freshly allocated
    // host output arrays) hide behind the solver.  T is bounded by host memory only.
    const size_t per_vol = nv * ((size_t)C * (4 + 2 * rsz) + 12);
    const int lock = batch_wanted((long long)nv);
    const char *cap_env = getenv("FR3D_STAGE_KIB");  // staging budget override (tests use it to force windows)
    const size_t cap = cap_env ? (size_t)std::max(1, atoi(cap_env)) << 10 : (8ull << 30);
    int win = (int)std::max<size_t>(1, std::min<size_t>((size_t)lock, cap / per_vol));
    if (win > T) win = T;
    // a series that fits one window would upload, compute and download one after the other: two half windows
    // overlap the transfers with the solver (a lock-step batch of 4 runs within 2 % of a batch of 8 per volume).
    // More, smaller windows (8 volumes as 3 + 3 + 2) were measured and lost: 8.6 against 9.3 volumes/s at 256^3.
    if (win == T && T >= 4) win = cdiv(T, 2);
    const int nwin = win > 0 ? cdiv(T, win) : 0;
    const int nset = nwin > 1 ? 2 : 1;
    float *dbp[2], *dfl[2];
    char *dbr[2], *dre[2];
    for (int q = 0; q < nset; q++) {
        dbp[q] = (float *)s.alloc(nv * C * 4 * (size_t)win);
        dbr[q] = (char *)s.alloc(nv * C * rsz * (size_t)win);
        dfl[q] = (float *)s.alloc(nv * 3 * 4 * (size_t)win);
        dre[q] = (char *)s.alloc(nv * C * rsz * (size_t)win);
    }
    // the reference and weight pyramids are the same for every window: build them once
    int min_level = p->min_level;
    std::vector<Level> lv = make_schedule(Z, Y, X, p->eta, p->levels, min_level);
    RefPyramid rp;
    mark("reference arrays uploaded", 0);
    if (T > 0) build_ref_pyramid(g_eng, lv, drp, dwt, Z, Y, X, C, rp, "pb_");
    mark("reference pyramid enqueued", 0);
    std::mutex mu;
    std::condition_variable cv;
    int uploaded = 0, computed = 0;  // windows finished by each side
    bool abort_all = false;
    std::exception_ptr copier_error;
    const int device = g_eng.device;
    auto count_of = [&](int k) { return std::min(win, T - k * win); };
    bool pinned_fl = false, pinned_re = false;
    std::thread copier([&]() {
        hipStream_t cs = nullptr;
        try {
            FR3D_HIP(hipSetDevice(device));
            FR3D_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            auto upload = [&](int k) {
                const int q = k % nset, nt = count_of(k);
                const size_t o = (size_t)k * win * nv * C;
                FR3D_HIP(hipMemcpyAsync(dbp[q], batch_proc + o, nv * C * 4 * (size_t)nt, hipMemcpyHostToDevice, cs));
                FR3D_HIP(hipMemcpyAsync(dbr[q], (const char *)batch_raw + o * rsz, nv * C * rsz * (size_t)nt,
                                        hipMemcpyHostToDevice, cs));
                FR3D_HIP(hipStreamSynchronize(cs));
                mark("window uploaded", k);
                std::lock_guard<std::mutex> lk(mu);
                uploaded = k + 1;
                cv.notify_all();
            };
            for (int k = 0; k < std::min(nset, nwin); k++) upload(k);
            // Results go back into the caller's pageable arrays: unpinned, a device-to-host copy is staged through
            // a host memcpy at a few GB/s on this thread (the 2 GB of a 256^3 batch of 8: 0.2 s, most of it after the
            // last window, where nothing hides it).  Pinning the output arrays -- here, while the first window
            // computes -- lets the copies run as plain DMA.  Not being able to pin (memlock limits) is not an error.
            const size_t fl_bytes = (size_t)T * nv * 3 * 4, re_bytes = (size_t)T * nv * C * rsz;
            if (fl_bytes + re_bytes >= ((size_t)64 << 20)) {
                pinned_fl = hipHostRegister(flows_out, fl_bytes, hipHostRegisterDefault) == hipSuccess;
                pinned_re = hipHostRegister(registered_out, re_bytes, hipHostRegisterDefault) == hipSuccess;
                (void)hipGetLastError();
                mark("output arrays pinned", 0);
            }
            for (int k = 0; k < nwin; k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return computed > k || abort_all; });
                    if (abort_all) break;
                }
                const int q = k % nset, nt = count_of(k);
                FR3D_HIP(hipMemcpyAsync(flows_out + (size_t)k * win * nv * 3, dfl[q], nv * 3 * 4 * (size_t)nt,
                                        hipMemcpyDeviceToHost, cs));
                FR3D_HIP(hipMemcpyAsync((char *)registered_out + (size_t)k * win * nv * C * rsz, dre[q],
                                        nv * C * rsz * (size_t)nt, hipMemcpyDeviceToHost, cs));
                FR3D_HIP(hipStreamSynchronize(cs));
                mark("window downloaded", k);
                if (k + nset < nwin) upload(k + nset);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            copier_error = std::current_exception();
            abort_all = true;
            cv.notify_all();
        }
        if (cs) (void)hipStreamDestroy(cs);
        if (pinned_fl) (void)hipHostUnregister(flows_out);
        if (pinned_re) (void)hipHostUnregister(registered_out);
        mark("copier done", 0);
    });
    std::exception_ptr main_error;
    try {
        for (int k = 0; k < nwin; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return uploaded > k || abort_all; });
                if (abort_all) break;
            }
            const int q = k % nset;
            process_batch_dev(p, dbp[q], dbr[q], raw_dtype, drp, drr, ref_dtype, dwi, dwt, count_of(k), Z, Y, X, C, order,
                              dfl[q], dre[q], progress, user, &rp);  // returns with the engine stream drained
            mark("window computed", k);
            std::lock_guard<std::mutex> lk(mu);
            computed = k + 1;
            cv.notify_all();
        }
    } catch (...) {
        main_error = std::current_exception();
        std::lock_guard<std::mutex> lk(mu);
        abort_all = true;
        cv.notify_all();
    }
    copier.join();
    mark("exit", 0);
    if (main_error) std::rethrow_exception(main_error);
    if (copier_error) std::rethrow_exception(copier_error);
    FR3D_CATCH
}

int fr3d_process_batch(const fr3d_params *p, const float *batch_proc, const float *batch_raw,
                       const float *ref_proc, const float *ref_raw, const float *w_init, const float *weight,
                       int T, int Z, int Y, int X, int C, int order, float *flows_out, float *registered_out,
                       fr3d_progress_fn progress, void *user)
{
    return fr3d_process_batch_raw(p, batch_proc, batch_raw, FR3D_F32, ref_proc, ref_raw, FR3D_F32, w_init, weight, T, Z,
                                  Y, X, C, order, flows_out, registered_out, progress, user);
}

int fr3d_update_reference_dev(const void *batch_proc, int proc_dtype, const float *flows, const void *ref_proc,
                              int ref_dtype, int T, int Z, int Y, int X, int C, int order, double *new_ref)
{
    FR3D_TRY
    update_reference_dev(batch_proc, proc_dtype, flows, ref_proc, ref_dtype, T, Z, Y, X, C, order, new_ref);
    FR3D_CATCH
}

int fr3d_update_reference(const void *batch_proc, int proc_dtype, const float *flows, const void *ref_proc,
                          int ref_dtype, int T, int Z, int Y, int X, int C, int order, double *new_ref)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(T >= 0 && Z >= 1 && Y >= 1 && X >= 1 && C >= 1, "bad update_reference shape");
    FR3D_CHECK((T == 0 || (batch_proc && flows)) && ref_proc && new_ref, "NULL pointer");
    FR3D_CHECK((proc_dtype == FR3D_F32 || proc_dtype == FR3D_F64) && (ref_dtype == FR3D_F32 || ref_dtype == FR3D_F64),
               "batch_proc / reference_proc must be float32 or float64");
    const size_t nv = (size_t)Z * Y * X;
    const int n_ref = std::min(100, T);
    if (n_ref < 1) return 0;  // the reference returns without touching reference_proc (:398-399)
    const size_t psz = dtype_size(proc_dtype), rsz = dtype_size(ref_dtype);
    Staged s;
    // only the last n_ref volumes are read
    const void *dbp = s.up((const char *)batch_proc + (size_t)(T - n_ref) * nv * C * psz, (size_t)n_ref * nv * C * psz);
    const float *dfl = (const float *)s.up(flows + (size_t)(T - n_ref) * nv * 3, (size_t)n_ref * nv * 3 * 4);
    const void *drp = s.up(ref_proc, nv * C * rsz);
    double *dout = (double *)s.alloc(nv * C * 8);
    update_reference_dev(dbp, proc_dtype, dfl, drp, ref_dtype, n_ref, Z, Y, X, C, order, dout);
    FR3D_HIP(hipMemcpy(new_ref, dout, nv * C * 8, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_mean_stack_dev(const float *stack, int count, size_t n, float *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(stack && out && count >= 1, "bad mean_stack arguments");
    launch_mean_stack_f32(g_eng.st, stack, count, (long long)n, out);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_CATCH
}

int fr3d_gaussian_filter_dev(const void *frames, int dtype, int T, int Z, int Y, int X, int C, const double *norm_min,
                             const double *norm_den, const double *sigma, double truncate, int mode, void *out,
                             int out_dtype)
{
    FR3D_TRY
    preprocess_dev(frames, dtype, T, Z, Y, X, C, norm_min, norm_den, sigma, truncate, out, out_dtype, mode);
    FR3D_CATCH
}

int fr3d_gaussian_filter(const void *frames, int dtype, int T, int Z, int Y, int X, int C, const double *norm_min,
                         const double *norm_den, const double *sigma, double truncate, int mode, void *out,
                         int out_dtype)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(frames && out, "NULL pointer");
    FR3D_CHECK(T >= 0 && Z >= 1 && Y >= 1 && X >= 1 && C >= 1, "bad preprocess shape");
    FR3D_CHECK(out_dtype == FR3D_F32 || out_dtype == FR3D_F64, "out_dtype must be FR3D_F32 or FR3D_F64");
    const size_t n = (size_t)T * Z * Y * X * C;
    Staged s;
    const void *din = s.up(frames, n * dtype_size(dtype));
    void *dout = s.alloc(n * dtype_size(out_dtype));
    preprocess_dev(din, dtype, T, Z, Y, X, C, norm_min, norm_den, sigma, truncate, dout, out_dtype, mode);
    FR3D_HIP(hipMemcpy(out, dout, n * dtype_size(out_dtype), hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_preprocess_dev(const void *frames, int dtype, int T, int Z, int Y, int X, int C, const double *norm_min,
                        const double *norm_den, const double *sigma, double truncate, void *out, int out_dtype)
{
    return fr3d_gaussian_filter_dev(frames, dtype, T, Z, Y, X, C, norm_min, norm_den, sigma, truncate,
                                    FR3D_BOUNDARY_REFLECT, out, out_dtype);
}

int fr3d_preprocess(const void *frames, int dtype, int T, int Z, int Y, int X, int C, const double *norm_min,
                    const double *norm_den, const double *sigma, double truncate, void *out, int out_dtype)
{
    return fr3d_gaussian_filter(frames, dtype, T, Z, Y, X, C, norm_min, norm_den, sigma, truncate, FR3D_BOUNDARY_REFLECT,
                                out, out_dtype);
}

static void flow_stats_dev(const float *flows, int T, int Z, int Y, int X, double *out)
{
    ensure_init();
    FR3D_CHECK(flows && out && T >= 0 && Z >= 1 && Y >= 1 && X >= 1, "bad flow_stats arguments");
    Engine &e = g_eng;
    const long long n = (long long)Z * Y * X;
    const int nb = (int)std::min<long long>(1024, (n + 255) / 256);
    double *dpart = e.f64("stats_part", (size_t)nb * 6);
    std::vector<double> hpart((size_t)nb * 6);
    for (int t = 0; t < T; t++) {
        launch_flow_stats(e.st, flows + (size_t)t * n * 3, Z, Y, X, nb, dpart);
        FR3D_HIP(hipMemcpyAsync(hpart.data(), dpart, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, e.st));
        FR3D_HIP(hipStreamSynchronize(e.st));
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nb; b++)
            for (int q = 0; q < 6; q++) {
                if (q == 1) acc[q] = std::max(acc[q], hpart[(size_t)b * 6 + q]);
                else acc[q] += hpart[(size_t)b * 6 + q];
            }
        double *o = out + (size_t)t * 6;
        o[0] = acc[0] / (double)n; o[1] = acc[1]; o[2] = acc[2] / (double)n;
        o[3] = acc[3] / (double)n; o[4] = acc[4] / (double)n; o[5] = acc[5] / (double)n;
    }
}

int fr3d_flow_stats_dev(const float *flows, int T, int Z, int Y, int X, double *out)
{
    FR3D_TRY
    flow_stats_dev(flows, T, Z, Y, X, out);
    FR3D_CATCH
}

int fr3d_flow_stats(const float *flows, int T, int Z, int Y, int X, double *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(flows && out && T >= 0 && Z >= 1 && Y >= 1 && X >= 1, "bad flow_stats arguments");
    Staged s;
    const float *d = (const float *)s.up(flows, (size_t)T * Z * Y * X * 3 * 4);
    flow_stats_dev(d, T, Z, Y, X, out);
    FR3D_CATCH
}

// ---- kernel-level entry points ------------------------------------------------------------------

int fr3d_resize3d(const float *src, int D, int H, int W, int od, int oh, int ow, float *dst)
{
    return fr3d_resize3d_ex(src, D, H, W, od, oh, ow, 0.6, 0, dst);
}

int fr3d_resize3d_ex(const float *src, int D, int H, int W, int od, int oh, int ow, double sigma_coeff, int per_axis,
                     float *dst)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(src && dst && D > 0 && H > 0 && W > 0 && od > 0 && oh > 0 && ow > 0, "bad resize arguments");
    FR3D_CHECK(sigma_coeff >= 0.0 && sigma_coeff * std::max(D, std::max(H, W)) < 4096.0, "sigma_coeff out of range");
    Staged s;
    const float *ds = (const float *)s.up(src, (size_t)D * H * W * 4);
    float *dd = (float *)s.alloc((size_t)od * oh * ow * 4);
    resize3d(g_eng, ds, 1, 0, D, H, W, od, oh, ow, dd, sigma_coeff, per_axis != 0);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(dst, dd, (size_t)od * oh * ow * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_motion_tensor(const float *f1, const float *f2, int Z, int Y, int X, double hz, double hy, double hx,
                       float *J, float *A)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(f1 && f2 && J && Z > 0 && Y > 0 && X > 0, "bad tensor arguments");
    const size_t n = (size_t)Z * Y * X;
    Staged s;
    const float *d1 = (const float *)s.up(f1, n * 4);
    const float *d2 = (const float *)s.up(f2, n * 4);
    float *dj = (float *)s.alloc(n * 10 * 4);
    float *da = A ? (float *)s.alloc(n * 12 * 4) : nullptr;
    float *Jo[10];
    for (int a = 0; a < 10; a++) Jo[a] = dj + (size_t)a * n;
    launch_motion_tensor<float>(g_eng.st, d1, d2, Z, Y, X, hz, hy, hx, Jo, da, (long long)n, nullptr);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(J, dj, n * 10 * 4, hipMemcpyDeviceToHost));
    if (A) FR3D_HIP(hipMemcpy(A, da, n * 12 * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_level_solve(const float *A, const float *weight, const float *uvw, int Z, int Y, int X, int C,
                     const double *alpha3, int iterations, int update_lag, const double *a_data, double a_smooth,
                     double hx, double hy, double hz, int solver_fp64, float *duvw_out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(A && weight && uvw && alpha3 && a_data && duvw_out, "NULL pointer");
    FR3D_CHECK(Z > 0 && Y > 0 && X > 0 && C >= 1 && C <= FR3D_MAX_CHANNELS, "bad solver shape");
    FR3D_CHECK(iterations >= 0 && update_lag >= 1, "iterations >= 0 and update_lag >= 1 required");
    Engine &e = g_eng;
    const size_t n = (size_t)Z * Y * X;
    const bool fast = a_smooth == 1.0;  // k_sor.hip / k_sor_smooth.hip; compact layout and records in both
    const Skew sk = e.compact_skew(Z, Y, X);
    const size_t ns = (size_t)sk.total;
    Staged s;
    const float *dA = (const float *)s.up(A, n * 12 * C * 4);
    const float *dW = (const float *)s.up(weight, n * C * 4);
    const float *dU = (const float *)s.up(uvw, n * 3 * 4);
    float *Msk = (float *)s.alloc(ns * 9 * 4);
    float *Ask = (float *)s.alloc(ns * 12 * C * 4);
    float *wsk = (float *)s.alloc(ns * C * 4);
    float *Lb = (float *)s.alloc(ns * 3 * 4);
    float *db = (float *)s.alloc(ns * 3 * 4);
    float *dn = (float *)s.alloc(n * 3 * 4);
    SorArgs a;
    std::memset(&a, 0, sizeof(a));
    a.sk = sk;
    a.C = C;
    a.M = Msk;
    a.L = Lb;
    a.d = db;
    // A arrives as (12, C, Z, Y, X): factor q of channel c at (q*C + c)*n
    for (int c = 0; c < C; c++) {
        float *Adst = Ask + (size_t)c * 12 * ns;
        launch_skew_pack<float, float>(e.st, dA + (size_t)c * n, (long long)C * n, Adst, 12, sk);
        a.A[c] = Adst;
        launch_skew_pack<float, float>(e.st, dW + (size_t)c * n, 0, wsk + (size_t)c * ns, 1, sk);
        a.weight[c] = wsk + (size_t)c * ns;
        a.a_data[c] = a_data[c];
    }
    a.ax = alpha3[0] / (hx * hx);
    a.ay = alpha3[1] / (hy * hy);
    a.az = alpha3[2] / (hz * hz);
    FR3D_HIP(hipMemsetAsync(db, 0, ns * 3 * 4, e.st));
    a.iterations = iterations;
    a.update_lag = update_lag;
    if (fast) {
        launch_laplace_rec<float>(e.st, dU, dU + n, dU + 2 * n, sk, a.ax, a.ay, a.az, Lb);
        launch_sor<float>(e.st, a, solver_fp64 != 0, e.chain_sched(sk, iterations));
    } else {
        float *smU = (float *)s.alloc(ns * 3 * 4), *smD = (float *)s.alloc(ns * 9 * 4), *smP = (float *)s.alloc(ns * 4);
        SmoothArgs<float> sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.view.Z = Z; sa.view.Y = Y; sa.view.X = X; sa.view.sk = sk;
        smooth_set_spacing(sa.view, hx, hy, hz);
        sa.view.a_smooth = a_smooth;
        launch_skew_pack<float, float>(e.st, dU, (long long)n, smU, 3, sk);
        FR3D_HIP(hipMemsetAsync(smD, 0, ns * 9 * 4, e.st));
        sa.view.U = smU;
        sa.nvol = 1;
        for (int m = 0; m < 3; m++) sa.D[m] = smD + (size_t)m * 3 * ns;
        sa.Ps = smP;
        sa.M = Msk;
        for (int c = 0; c < C; c++) {
            sa.A[c] = a.A[c];
            sa.weight[c] = a.weight[c];
            sa.a_data[c] = a_data[c];
        }
        sa.ax = a.ax; sa.ay = a.ay; sa.az = a.az;
        sa.C = C; sa.iterations = iterations; sa.update_lag = update_lag; sa.S_planes = sk.S;
        launch_sor_smooth<float>(e.st, sa, e.sched(sk, iterations, SM_LAG));
        if (iterations > 0)
            FR3D_HIP(hipMemcpyAsync(db, sa.D[(iterations - 1) % 3], ns * 3 * 4, hipMemcpyDeviceToDevice, e.st));
    }
    launch_unskew_unpack<float, float>(e.st, db, dn, (long long)n, 3, sk);
    FR3D_HIP(hipStreamSynchronize(e.st));
    FR3D_HIP(hipMemcpy(duvw_out, dn, n * 3 * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_median5(const float *in, int Z, int Y, int X, float *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(in && out && Z > 0 && Y > 0 && X > 0, "bad median arguments");
    const size_t n = (size_t)Z * Y * X;
    Staged s;
    const float *di = (const float *)s.up(in, n * 4);
    float *dout = (float *)s.alloc(n * 4);
    launch_median5(g_eng.st, di, Z, Y, X, dout);
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    FR3D_CATCH
}

int fr3d_schedule(int Z, int Y, int X, double eta, int levels, int min_level, int *sizes, int max_out,
                  int *min_level_eff)
{
    try {
        if (Z < 1 || Y < 1 || X < 1 || !(eta > 0.0 && eta <= 1.0) || levels < 1) {
            g_err = "bad schedule arguments";
            return -1;
        }
        std::vector<Level> lv = make_schedule(Z, Y, X, eta, levels, min_level);
        for (size_t i = 0; i < lv.size() && (int)i < max_out && sizes; i++) {
            sizes[3 * i + 0] = lv[i].z;
            sizes[3 * i + 1] = lv[i].y;
            sizes[3 * i + 2] = lv[i].x;
        }
        if (min_level_eff) *min_level_eff = min_level;
        return (int)lv.size();
    } catch (...) {
        g_err = "schedule failed";
        return -1;
    }
}

// ---- memory helpers -----------------------------------------------------------------------------

long long fr3d_sor_schedule_check(int Z, int Y, int X, int iterations, int tile_rows, int chain, long long *n_updates)
{
    try {
        if (Z < 1 || Y < 1 || X < 1 || iterations < 0 || tile_rows < 0 || chain < 0) {
            g_err = "bad schedule arguments";
            return -1;
        }
        if (tile_rows == 0 || chain == 0) sor_tile_shape(make_skew(Z, Y, X), tile_rows, chain);
        return check_chain_schedule(Z, Y, X, iterations, tile_rows, chain, n_updates);
    } catch (const std::exception &ex) {
        g_err = ex.what();
        return -1;
    }
}

void *fr3d_dev_malloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        g_err = "hipMalloc failed";
        return nullptr;
    }
    return p;
}
void fr3d_dev_free(void *p)
{
    if (p) (void)hipFree(p);
}
int fr3d_h2d(void *dst, const void *src, size_t bytes)
{
    FR3D_TRY
    FR3D_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    FR3D_CATCH
}
int fr3d_d2h(void *dst, const void *src, size_t bytes)
{
    FR3D_TRY
    FR3D_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    FR3D_CATCH
}
int fr3d_sync(void)
{
    FR3D_TRY
    ensure_init();
    FR3D_HIP(hipStreamSynchronize(g_eng.st));
    FR3D_CATCH
}

// ---- measurement --------------------------------------------------------------------------------

static void fold_spans()
{
    for (Engine *en : {&g_eng, &g_eng2}) {
        Engine &e = *en;
        if (e.spans.empty()) continue;
        FR3D_HIP(hipStreamSynchronize(e.st));
        for (auto &s : e.spans) {
            float ms = 0.0f;
            FR3D_HIP(hipEventElapsedTime(&ms, s.a, s.b));
            e.acc[s.kid].ms += ms;
            e.ev_pool.push_back(s.a);
            e.ev_pool.push_back(s.b);
        }
        e.spans.clear();
    }
}

int fr3d_prof_enable(int on)
{
    FR3D_TRY
    ensure_init();
    fold_spans();
    g_eng.prof = g_eng2.prof = on != 0;
    FR3D_CATCH
}
int fr3d_prof_reset(void)
{
    FR3D_TRY
    ensure_init();
    fold_spans();
    std::memset(g_eng.acc, 0, sizeof(g_eng.acc));
    std::memset(g_eng2.acc, 0, sizeof(g_eng2.acc));
    FR3D_CATCH
}
int fr3d_stream_probe(size_t n_floats, int reps, double *gbytes_per_s)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(n_floats > 0 && reps > 0 && gbytes_per_s, "bad stream_probe arguments");
    Engine &e = g_eng;
    Staged s;
    float *x = (float *)s.alloc(n_floats * 4), *y = (float *)s.alloc(n_floats * 4);
    FR3D_HIP(hipMemsetAsync(x, 0, n_floats * 4, e.st));
    FR3D_HIP(hipMemsetAsync(y, 0, n_floats * 4, e.st));
    launch_axpy(e.st, y, x, (long long)n_floats);  // first touch outside the timed part
    hipEvent_t a = e.get_event(), b = e.get_event();
    FR3D_HIP(hipEventRecord(a, e.st));
    for (int r = 0; r < reps; r++) launch_axpy(e.st, y, x, (long long)n_floats);
    FR3D_HIP(hipEventRecord(b, e.st));
    FR3D_HIP(hipEventSynchronize(b));
    float ms = 0.0f;
    FR3D_HIP(hipEventElapsedTime(&ms, a, b));
    e.ev_pool.push_back(a);
    e.ev_pool.push_back(b);
    *gbytes_per_s = 12.0 * (double)n_floats * reps / ((double)ms * 1e-3) / 1e9;
    FR3D_CATCH
}
int fr3d_read_probe(size_t n_floats, int reps, double *gbytes_per_s)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(n_floats > 0 && reps > 0 && gbytes_per_s, "bad read_probe arguments");
    Engine &e = g_eng;
    Staged s;
    float *x = (float *)s.alloc(n_floats * 8 * 4), *sink = (float *)s.alloc(64 * 4);
    FR3D_HIP(hipMemsetAsync(x, 0, n_floats * 8 * 4, e.st));
    launch_read8(e.st, x, (long long)n_floats, sink);  // first touch outside the timed part
    hipEvent_t a = e.get_event(), b = e.get_event();
    FR3D_HIP(hipEventRecord(a, e.st));
    for (int r = 0; r < reps; r++) launch_read8(e.st, x, (long long)n_floats, sink);
    FR3D_HIP(hipEventRecord(b, e.st));
    FR3D_HIP(hipEventSynchronize(b));
    float ms = 0.0f;
    FR3D_HIP(hipEventElapsedTime(&ms, a, b));
    e.ev_pool.push_back(a);
    e.ev_pool.push_back(b);
    *gbytes_per_s = 32.0 * (double)n_floats * reps / ((double)ms * 1e-3) / 1e9;
    FR3D_CATCH
}
}  // extern "C"
namespace fr3d { void launch_xcd_probe(hipStream_t st, int gx, int gy, int *out); }  // k_misc.hip (diagnostic)
extern "C" {
int fr3d_xcd_probe(int grid_x, int grid_y, int *xcc_of_block)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(grid_x > 0 && grid_y > 0 && grid_y <= 65535 && xcc_of_block, "bad xcd_probe arguments");
    Engine &e = g_eng;
    Staged s;
    const size_t n = (size_t)grid_x * grid_y;
    int *d = (int *)s.alloc(n * sizeof(int));
    FR3D_HIP(hipMemsetAsync(d, 0xff, n * sizeof(int), e.st));
    launch_xcd_probe(e.st, grid_x, grid_y, d);
    FR3D_HIP(hipStreamSynchronize(e.st));
    FR3D_HIP(hipMemcpy(xcc_of_block, d, n * sizeof(int), hipMemcpyDeviceToHost));
    FR3D_CATCH
}
int fr3d_prof_get(fr3d_kernel_stat *out)
{
    FR3D_TRY
    ensure_init();
    FR3D_CHECK(out, "NULL pointer");
    fold_spans();
    // both lanes added up: with two lanes the spans of one lane run while the other lane's kernels share the device, so
    // the times are per-lane stream times, not exclusive kernel times (per-kernel accounting: fr3d_set_lanes(1))
    for (int k = 0; k < FR3D_K_COUNT; k++) {
        out[k] = g_eng.acc[k];
        out[k].ms += g_eng2.acc[k].ms;
        out[k].algo_bytes += g_eng2.acc[k].algo_bytes;
        out[k].launches += g_eng2.acc[k].launches;
        out[k].units += g_eng2.acc[k].units;
    }
    FR3D_CATCH
}

}  // extern "C"
