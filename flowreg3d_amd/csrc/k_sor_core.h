// k_sor_core.h -- the per-voxel arithmetic of the a_smooth == 1 SOR sweep (k_sor.hip), kept apart
// from the launch geometry so that alternative sweep kernels reuse it and stay bit-identical.  Reference: core/level_solver_3d.py:356-377 (psi_data),
// :472-493 (stencil), :503-540 (du -> dv -> dw relaxation with omega = 1.95).
#pragma once

#include <type_traits>

#include "fr3d_internal.h"

namespace fr3d {

#define SOR_OMEGA 1.95

// The per-voxel arithmetic below also compiles for the host: the window kernel's CPU emulator
// (tools/emu/sor_win_emu.hip) runs the same code thread by thread.
#define FR3D_HD __host__ __device__ __forceinline__

template <typename R> FR3D_HD R fma_(R a, R b, R c);
template <> FR3D_HD float fma_<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> FR3D_HD double fma_<double>(double a, double b, double c) { return fma(a, b, c); }

#ifdef FR3D_EXPERIMENTS
// numerics what-if (FR3D_SOR_DBG bits 64 / 128, mantissa bits in bits 8..15): round a group of values to a BLOCK
// floating-point format -- one shared exponent (that of the largest magnitude) and `bits` magnitude bits each
template <typename V, int N>
__device__ __forceinline__ void bfp_round(V (&v)[N], int first, int n, int bits)
{
    double mx = 0.0;
    for (int q = 0; q < n; q++) mx = fmax(mx, fabs((double)v[first + q]));
    if (!(mx > 0.0) || !(mx < 1e300)) return;
    int ex;
    (void)frexp(mx, &ex);  // mx < 2^ex
    const double scale = ldexp(1.0, ex - bits);
    for (int q = 0; q < n; q++) v[first + q] = (V)(rint((double)v[first + q] / scale) * scale);
}
#endif

FR3D_HD double bits_double(unsigned hi, unsigned lo)
{
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
FR3D_HD unsigned long long double_bits(double x) { return __builtin_bit_cast(unsigned long long, x); }

// ---- storage formats of the solver operands -------------------------------------------------------------
// float / double: plain arrays.  pk42 (fr3d_params.solver_fp64 == 3): three values share 16 bytes -- the upper 42
// bits of each value's fp64 pattern (sign, 11 exponent bits, 30 mantissa bits: 31 significant bits, 128x finer than
// fp32) -- dwords 0..2 hold the high words, dword 3 the three 10-bit continuations.  Decoding is two integer
// operations per value and no conversion (the register pair IS the double); a record of N values (N a multiple
// of 3: every record of the sweep is) takes 4N/3 dwords and is fetched as N/3 dwordx4 loads, 16-B aligned.
// Why: at 512^3 every operand group held in fp32 costs ~1e-4 of flow error (profiles/r02/numerics_512_*), fp64
// storage doubles the sweep's bytes; 42 bits keep the error at the fp64 level for 2/3 of its bytes.
template <typename S> struct Sto;
template <> struct Sto<float> {
    using val = float;   // type of a decoded value
    using wt = float;    // storage type of the per-voxel channel weights
    static constexpr bool wide = false;  // fp64-grade storage: fp64 pow and fp64 update arithmetic
    __host__ __device__ static constexpr long long elems(long long nvals) { return nvals; }
    static constexpr double bytes_per_value = 4.0;
    FR3D_HD static float quant(float x) { return x; }
    FR3D_HD static float quant(double x) { return (float)x; }
};
template <> struct Sto<double> {
    using val = double;
    using wt = double;
    static constexpr bool wide = true;
    __host__ __device__ static constexpr long long elems(long long nvals) { return nvals; }
    static constexpr double bytes_per_value = 8.0;
    FR3D_HD static double quant(double x) { return x; }
};
#define PK42_ROUND (1ull << 21)
#define PK42_MASK (~((1ull << 22) - 1ull))
template <> struct Sto<pk42> {
    using val = double;
    using wt = float;  // weights are resampled fp32 values: exact in float
    static constexpr bool wide = true;
    __host__ __device__ static constexpr long long elems(long long nvals) { return nvals / 3 * 4; }
    static constexpr double bytes_per_value = 16.0 / 3.0;
    // round to nearest (ties away from zero) at bit 22 of the fp64 pattern; inf stays inf
    FR3D_HD static double quant(double x)
    {
        const unsigned long long q = (double_bits(x) + PK42_ROUND) & PK42_MASK;
        return bits_double((unsigned)(q >> 32), (unsigned)q);
    }
};

// A record of N consecutive values of one voxel; loading it as one object lets the compiler emit wide
// global loads (dwordx3 / dwordx4) instead of N dword loads.  `v` holds DECODED values.
template <typename S, int N>
struct Rec {
    typename Sto<S>::val v[N];
};
template <typename S, int N>
FR3D_HD Rec<S, N> ldrec(const S *base, long long voxel)
{
    if constexpr (std::is_same<S, pk42>::value) {
        static_assert(N % 3 == 0, "pk42 records hold triples");
        const uint4 *p = reinterpret_cast<const uint4 *>(base) + voxel * (N / 3);
        Rec<S, N> r;
#pragma unroll
        for (int g = 0; g < N / 3; g++) {
            const uint4 q = p[g];
            r.v[3 * g + 0] = bits_double(q.x, (q.w & 0x3FFu) << 22);
            r.v[3 * g + 1] = bits_double(q.y, ((q.w >> 10) & 0x3FFu) << 22);
            r.v[3 * g + 2] = bits_double(q.z, (q.w >> 20) << 22);
        }
        return r;
    } else {
        return *reinterpret_cast<const Rec<S, N> *>(base + voxel * N);
    }
}
// A record as it lies in memory: loading it does not touch the loaded bits (ldrec decodes packed values at once, which
// makes the consumer of a PREFETCH wait for the data where the load is issued); dec() decodes at the point of use.
template <typename S, int N>
struct RawRec {
    typename Sto<S>::val w[N];
    FR3D_HD Rec<S, N> dec() const
    {
        Rec<S, N> r;
#pragma unroll
        for (int n = 0; n < N; n++) r.v[n] = w[n];
        return r;
    }
};
template <int N>
struct RawRec<pk42, N> {
    static_assert(N % 3 == 0, "pk42 records hold triples");
    uint4 w[N / 3];
    FR3D_HD Rec<pk42, N> dec() const
    {
        Rec<pk42, N> r;
#pragma unroll
        for (int g = 0; g < N / 3; g++) {
            const uint4 q = w[g];
            r.v[3 * g + 0] = bits_double(q.x, (q.w & 0x3FFu) << 22);
            r.v[3 * g + 1] = bits_double(q.y, ((q.w >> 10) & 0x3FFu) << 22);
            r.v[3 * g + 2] = bits_double(q.z, (q.w >> 20) << 22);
        }
        return r;
    }
};
template <typename S, int N>
FR3D_HD RawRec<S, N> ldraw(const S *base, long long voxel)
{
    if constexpr (std::is_same<S, pk42>::value)
        return *reinterpret_cast<const RawRec<S, N> *>(reinterpret_cast<const uint4 *>(base) + voxel * (N / 3));
    else
        return *reinterpret_cast<const RawRec<S, N> *>(base + voxel * N);
}
// values must already be representable (Sto<S>::quant) -- the bits below the format are dropped
template <typename S, int N>
FR3D_HD void strec(S *base, long long voxel, const Rec<S, N> &r)
{
    if constexpr (std::is_same<S, pk42>::value) {
        uint4 *p = reinterpret_cast<uint4 *>(base) + voxel * (N / 3);
#pragma unroll
        for (int g = 0; g < N / 3; g++) {
            const unsigned long long b0 = double_bits(r.v[3 * g + 0]);
            const unsigned long long b1 = double_bits(r.v[3 * g + 1]);
            const unsigned long long b2 = double_bits(r.v[3 * g + 2]);
            uint4 q;
            q.x = (unsigned)(b0 >> 32);
            q.y = (unsigned)(b1 >> 32);
            q.z = (unsigned)(b2 >> 32);
            q.w = ((unsigned)b0 >> 22) | (((unsigned)b1 >> 22) << 10) | (((unsigned)b2 >> 22) << 20);
            p[g] = q;
        }
    } else {
        *reinterpret_cast<Rec<S, N> *>(base + voxel * N) = r;
    }
}

// The 3x3 system of one voxel for the current psi window: m[0..5] = M11,M22,M33,M12,M13,M23 with
// M = sum_c w_c psi_c J_c, m[6..8] = b = L - sum_c w_c psi_c (J14,J24,J34)_c.  psi is frozen between
// psi-update iterations (level_solver_3d.py:356), so M and b are too: an update iteration builds them from the
// square-root factors (sor_accum_channel per channel, then sor_finish_system) and stores them; the other
// iterations stream the 9 stored values -- independent of the channel count.
template <typename R>
struct SorAcc {
    R M11 = 0, M22 = 0, M33 = 0, M12 = 0, M13 = 0, M23 = 0, bu = 0, bv = 0, bw = 0;
};
// psi_data update (level_solver_3d.py:356-377) of one channel from the increments of iteration t-1, and the channel's
// share of M and of sum_c w psi (J14,J24,J34).  The quadratic form is evaluated as the sum of three squared residuals
// of the tensor's square-root factors (see k_tensor.hip) -- algebraically the reference's expression, but stable
// with fp32 storage.  `wt` = the channel weight of the voxel, `adc` = a_data of the channel.
template <typename R, typename S>
FR3D_HD void sor_accum_channel(const Rec<S, 12> &fr, double wt, double adc, R du0, R dv0, R dw0, SorAcc<R> &acc,
                               [[maybe_unused]] int dbg = 0)
{
    const typename Sto<S>::val *f = fr.v;
    if (adc != 1.0) {
        const double u_ = (double)du0, v_ = (double)dv0, w_ = (double)dw0;
        double val = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double r = fma((double)f[4 * k], u_, fma((double)f[4 * k + 1], v_,
                           fma((double)f[4 * k + 2], w_, (double)f[4 * k + 3])));
            val = fma(r, r, val);
        }
        // fp32 powf (~1 ulp) with fp32 storage: the products below are stored in fp32 anyway, and the fp64
        // pow's ~600-instruction dependent chain set a ~6 us latency floor on every launch
#ifdef FR3D_EXPERIMENTS
        if (dbg & 32) wt *= adc * (val + 1e-6);  // timing experiment: psi without the pow
        else
#endif
        if (Sto<S>::wide) wt *= adc * pow(val + 1e-6, adc - 1.0);  // reference-grade modes
        else wt *= adc * (double)powf((float)(val + 1e-6), (float)(adc - 1.0));
    }
    const R w = (R)Sto<S>::quant(wt);
    const R x0 = (R)f[0], x1 = (R)f[1], x2 = (R)f[2], x3 = (R)f[3];
    const R y0 = (R)f[4], y1 = (R)f[5], y2 = (R)f[6], y3 = (R)f[7];
    const R z0 = (R)f[8], z1 = (R)f[9], z2 = (R)f[10], z3 = (R)f[11];
    acc.M11 = fma_<R>(w, fma_<R>(z0, z0, fma_<R>(y0, y0, x0 * x0)), acc.M11);
    acc.M22 = fma_<R>(w, fma_<R>(z1, z1, fma_<R>(y1, y1, x1 * x1)), acc.M22);
    acc.M33 = fma_<R>(w, fma_<R>(z2, z2, fma_<R>(y2, y2, x2 * x2)), acc.M33);
    acc.M12 = fma_<R>(w, fma_<R>(z0, z1, fma_<R>(y0, y1, x0 * x1)), acc.M12);
    acc.M13 = fma_<R>(w, fma_<R>(z0, z2, fma_<R>(y0, y2, x0 * x2)), acc.M13);
    acc.M23 = fma_<R>(w, fma_<R>(z1, z2, fma_<R>(y1, y2, x1 * x2)), acc.M23);
    acc.bu = fma_<R>(w, fma_<R>(z0, z3, fma_<R>(y0, y3, x0 * x3)), acc.bu);
    acc.bv = fma_<R>(w, fma_<R>(z1, z3, fma_<R>(y1, y3, x1 * x3)), acc.bv);
    acc.bw = fma_<R>(w, fma_<R>(z2, z3, fma_<R>(y2, y3, x2 * x3)), acc.bw);
}
// the frozen system in storage precision: M and b = L - sum_c w psi (J14,J24,J34)
template <typename R, typename S>
FR3D_HD Rec<S, 9> sor_finish_system(const SorAcc<R> &acc, const Rec<S, 3> &lr)
{
    const R b_u = (R)lr.v[0] - acc.bu;
    const R b_v = (R)lr.v[1] - acc.bv;
    const R b_w = (R)lr.v[2] - acc.bw;
    Rec<S, 9> mr;
    mr.v[0] = Sto<S>::quant(acc.M11); mr.v[1] = Sto<S>::quant(acc.M22); mr.v[2] = Sto<S>::quant(acc.M33);
    mr.v[3] = Sto<S>::quant(acc.M12); mr.v[4] = Sto<S>::quant(acc.M13); mr.v[5] = Sto<S>::quant(acc.M23);
    mr.v[6] = Sto<S>::quant(b_u); mr.v[7] = Sto<S>::quant(b_v); mr.v[8] = Sto<S>::quant(b_w);
    return mr;
}

// The sweep kernel of k_sor.hip keeps its own monolithic copy of the same arithmetic (splitting it changed the
// register allocation: 76 -> 81 VGPRs, one wave per SIMD less); tests/test_gpu_sor_window.py holds the two bit-identical.
// `e` is the voxel's index inside one volume's arrays, vM/vA/vL the (wave-uniform) element offsets of the
// volume's slab; `upd`: build (and, when `store`, write) the system, else read the stored one.
template <typename R, typename S, int C>
__device__ __forceinline__ void sor_system(const SorArgsT<S> &a, bool upd, bool store, long long vM, long long vA,
                                           long long vL, long long e, R du0, R dv0, R dw0, R (&m)[9])
{
    using V = typename Sto<S>::val;
    if (upd) {
        R M11 = 0, M22 = 0, M33 = 0, M12 = 0, M13 = 0, M23 = 0, bu = 0, bv = 0, bw = 0;
        // one channel at a time: unrolling over channels keeps 12C factors live
        const int nch = C > 0 ? C : a.C;  // C == 0: channel count at run time (5..FR3D_MAX_CHANNELS channels)
#pragma unroll 1
        for (int c = 0; c < nch; c++) {
            // psi_data update (level_solver_3d.py:356-377) from the increments of iteration t-1.
            // The quadratic form is evaluated as the sum of three squared residuals of the tensor's
            // square-root factors (see k_tensor.hip) -- algebraically the reference's expression,
            // but stable with fp32 storage.
            Rec<S, 12> fr = ldrec<S, 12>(a.A[c] + vA, e);
#ifdef FR3D_EXPERIMENTS
            if (a.dbg & 8) {
#pragma unroll
                for (int q = 0; q < 12; q++) fr.v[q] = (V)(float)fr.v[q];
            }
#endif
            const V *f = fr.v;
            double wt = (double)a.weight[c][e];
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
                // fp32 powf (~1 ulp) with fp32 storage: the products below are stored in fp32 anyway, and the fp64
                // pow's ~600-instruction dependent chain set a ~6 us latency floor on every launch
#ifdef FR3D_EXPERIMENTS
                if (a.dbg & 32) wt *= adc * (val + 1e-6);  // timing experiment: psi without the pow
                else
#endif
                if (Sto<S>::wide) wt *= adc * pow(val + 1e-6, adc - 1.0);  // reference-grade modes
                else wt *= adc * (double)powf((float)(val + 1e-6), (float)(adc - 1.0));
            }
            const R w = (R)Sto<S>::quant(wt);
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
        Rec<S, 3> lr = ldrec<S, 3>(a.L + vL, e);
#ifdef FR3D_EXPERIMENTS
        if (a.dbg & 4) { lr.v[0] = (V)(float)lr.v[0]; lr.v[1] = (V)(float)lr.v[1]; lr.v[2] = (V)(float)lr.v[2]; }
#endif
        const R b_u = (R)lr.v[0] - bu;
        const R b_v = (R)lr.v[1] - bv;
        const R b_w = (R)lr.v[2] - bw;
        Rec<S, 9> mr;
        mr.v[0] = Sto<S>::quant(M11); mr.v[1] = Sto<S>::quant(M22); mr.v[2] = Sto<S>::quant(M33);
        mr.v[3] = Sto<S>::quant(M12); mr.v[4] = Sto<S>::quant(M13); mr.v[5] = Sto<S>::quant(M23);
        mr.v[6] = Sto<S>::quant(b_u); mr.v[7] = Sto<S>::quant(b_v); mr.v[8] = Sto<S>::quant(b_w);
#ifdef FR3D_EXPERIMENTS
        if (a.dbg & 2) {
#pragma unroll
            for (int q = 0; q < 9; q++) mr.v[q] = (V)(float)mr.v[q];
        }
        if (a.dbg & 64) {  // M11..M23 with one exponent, b with another
            bfp_round(mr.v, 0, 6, (a.dbg >> 8) & 0xff);
            bfp_round(mr.v, 6, 3, (a.dbg >> 8) & 0xff);
        }
#endif
        if (store) strec<S, 9>(a.M + vM, e, mr);
        // use the stored (rounded) values so update and non-update iterations see one system
#pragma unroll
        for (int q = 0; q < 9; q++) m[q] = (R)mr.v[q];
    } else {
        const Rec<S, 9> mr = ldrec<S, 9>(a.M + vM, e);
#pragma unroll
        for (int q = 0; q < 9; q++) m[q] = (R)mr.v[q];
    }
}

// One relaxation of the voxel: s?_x/y/z are the sums of the two neighbour increments along each
// axis (a ghost neighbour contributes the voxel's own old value, set_boundary_3d :246-259).
// du uses old dv,dw; dv uses new du, old dw; dw uses new du,dv (level_solver_3d.py:503-540).
template <typename R>
FR3D_HD void sor_relax(const R (&m)[9], double axd, double ayd, double azd, R su_x, R sv_x,
                                          R sw_x, R su_y, R sv_y, R sw_y, R su_z, R sv_z, R sw_z, R du0, R dv0,
                                          R dw0, R &du1, R &dv1, R &dw1)
{
    const R ax = (R)axd, ay = (R)ayd, az = (R)azd;
    const R num_u = fma_<R>(az, su_z, fma_<R>(ay, su_y, fma_<R>(ax, su_x, m[6])));
    const R num_v = fma_<R>(az, sv_z, fma_<R>(ay, sv_y, fma_<R>(ax, sv_x, m[7])));
    const R num_w = fma_<R>(az, sw_z, fma_<R>(ay, sw_y, fma_<R>(ax, sw_x, m[8])));
    const R diag = (R)(2.0 * axd + 2.0 * ayd + 2.0 * azd);
    const R den_u = diag + m[0], den_v = diag + m[1], den_w = diag + m[2];
    const R om = (R)SOR_OMEGA, om1 = (R)(1.0 - SOR_OMEGA);
    R n2 = num_u - fma_<R>(m[4], dw0, m[3] * dv0);
    du1 = fma_<R>(om, (den_u != (R)0 ? n2 / den_u : (R)0), om1 * du0);
    n2 = num_v - fma_<R>(m[5], dw0, m[3] * du1);
    dv1 = fma_<R>(om, (den_v != (R)0 ? n2 / den_v : (R)0), om1 * dv0);
    n2 = num_w - fma_<R>(m[5], dv1, m[4] * du1);
    dw1 = fma_<R>(om, (den_w != (R)0 ? n2 / den_w : (R)0), om1 * dw0);
}


// ---- phase-ordered form of the sweep's loads (k_sor.hip, used with packed storage): a psi-update wave builds its
// system BEFORE the neighbour loads are issued, fences keep the groups of loads together ----
// pin(): the empty asm "uses" every value of the records, so the loads that produce them are issued (and waited
// for) before this point and cannot be sunk into later conditional code; ONE statement per group, because the
// scheduler may move an independent load below a fence it does not feed (at most 30 operands).
template <typename S>
__device__ __forceinline__ void pin(Rec<S, 3> &a)
{
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]));
}
template <typename S>
__device__ __forceinline__ void pin(Rec<S, 3> &a, Rec<S, 3> &b, Rec<S, 3> &c, Rec<S, 3> &d, Rec<S, 3> &e, Rec<S, 3> &f)
{
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(c.v[0]),
                 "+v"(c.v[1]), "+v"(c.v[2]), "+v"(d.v[0]), "+v"(d.v[1]), "+v"(d.v[2]), "+v"(e.v[0]), "+v"(e.v[1]),
                 "+v"(e.v[2]), "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]));
}
template <typename S>
__device__ __forceinline__ void pin(Rec<S, 3> &q, Rec<S, 9> &m, Rec<S, 3> &a, Rec<S, 3> &b, Rec<S, 3> &c, Rec<S, 3> &d,
                                    Rec<S, 3> &e, Rec<S, 3> &f)
{
    asm volatile("" : "+v"(q.v[0]), "+v"(q.v[1]), "+v"(q.v[2]), "+v"(m.v[0]), "+v"(m.v[1]), "+v"(m.v[2]), "+v"(m.v[3]),
                 "+v"(m.v[4]), "+v"(m.v[5]), "+v"(m.v[6]), "+v"(m.v[7]), "+v"(m.v[8]), "+v"(a.v[0]), "+v"(a.v[1]),
                 "+v"(a.v[2]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(c.v[0]), "+v"(c.v[1]), "+v"(c.v[2]),
                 "+v"(d.v[0]), "+v"(d.v[1]), "+v"(d.v[2]), "+v"(e.v[0]), "+v"(e.v[1]), "+v"(e.v[2]), "+v"(f.v[0]),
                 "+v"(f.v[1]), "+v"(f.v[2]));
}
// sor_relax with the quotients formed unconditionally and selected afterwards (den == 0 never occurs with
// alpha > 0; same values): no branch for the compiler to sink a neighbour load into
template <typename R>
FR3D_HD void sor_relax_sel(const R (&m)[9], double axd, double ayd, double azd, R su_x, R sv_x,
                                              R sw_x, R su_y, R sv_y, R sw_y, R su_z, R sv_z, R sw_z, R du0, R dv0,
                                              R dw0, R &du1, R &dv1, R &dw1)
{
    const R ax = (R)axd, ay = (R)ayd, az = (R)azd;
    const R num_u = fma_<R>(az, su_z, fma_<R>(ay, su_y, fma_<R>(ax, su_x, m[6])));
    const R num_v = fma_<R>(az, sv_z, fma_<R>(ay, sv_y, fma_<R>(ax, sv_x, m[7])));
    const R num_w = fma_<R>(az, sw_z, fma_<R>(ay, sw_y, fma_<R>(ax, sw_x, m[8])));
    const R diag = (R)(2.0 * axd + 2.0 * ayd + 2.0 * azd);
    const R den_u = diag + m[0], den_v = diag + m[1], den_w = diag + m[2];
    const R om = (R)SOR_OMEGA, om1 = (R)(1.0 - SOR_OMEGA);
    R n2 = num_u - fma_<R>(m[4], dw0, m[3] * dv0);
    R q = n2 / den_u;
    du1 = fma_<R>(om, (den_u != (R)0 ? q : (R)0), om1 * du0);
    n2 = num_v - fma_<R>(m[5], dw0, m[3] * du1);
    q = n2 / den_v;
    dv1 = fma_<R>(om, (den_v != (R)0 ? q : (R)0), om1 * dv0);
    n2 = num_w - fma_<R>(m[5], dv1, m[4] * du1);
    q = n2 / den_w;
    dw1 = fma_<R>(om, (den_w != (R)0 ? q : (R)0), om1 * dw0);
}

}  // namespace fr3d
