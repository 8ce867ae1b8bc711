/* portable_pow.h -- x^y from +, -, *, / and integer operations on the bit pattern only, so that the SAME source gives
 * the SAME bits under gcc on the host and hipcc on gfx950 (both compile without FMA contraction; there is no libm
 * call, no fma, no table).
 *
 * Who uses it: the engine's verification mode (fr3d_params.solver_fp64 == FR3D_SOLVER_VERIFY, k_sor_verify.hip) and
 * the `ppow` build of the CPU restatement of the reference (test infrastructure of this repository), where it replaces pow() in the
 * psi = a (x + eps)^(a-1) nonlinearities (core/level_solver_3d.py:310,377).  glibc's and the GPU math library's pow
 * differ in the last bit now and then, and the lagged-nonlinearity iteration amplifies a last-bit difference to 1e-3
 * voxels at single voxels -- with one pow on both sides, GPU and CPU agree bit for bit and
 * every other difference shows.
 *
 * Domain: x positive, finite, normal; |y ln x| < 700.  Accuracy: <= 2 ulp (measured against libm in
 * tests/test_portable_pow.py); it is a vehicle for bit-identity, not a replacement for pow() in the shipped modes. */
#ifndef FR3D_PORTABLE_POW_H
#define FR3D_PORTABLE_POW_H

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FR3D_PPOW_FN __host__ __device__ static inline
#else
#define FR3D_PPOW_FN static inline
#endif

FR3D_PPOW_FN double fr3d_ppow_bits2d(unsigned long long u)
{
    union { double d; unsigned long long u; } b;
    b.u = u;
    return b.d;
}
FR3D_PPOW_FN unsigned long long fr3d_ppow_d2bits(double d)
{
    union { double d; unsigned long long u; } b;
    b.d = d;
    return b.u;
}

/* floor for |v| < 2^51 without libm: truncate, step down for negative non-integers */
FR3D_PPOW_FN double fr3d_ppow_floor(double v)
{
    const double t = (double)(long long)v;
    return t > v ? t - 1.0 : t;
}

FR3D_PPOW_FN double fr3d_ppow(double x, double y)
{
    const double LN2_HI = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000: ln 2 with 21 trailing zero bits */
    const double LN2_LO = 1.90821492927058770002e-10; /* ln 2 - LN2_HI */
    const double INV_LN2 = 1.44269504088896338700e+00;
    /* x = m 2^e, m in (sqrt(1/2), sqrt(2)] */
    unsigned long long ux = fr3d_ppow_d2bits(x);
    int e = (int)((ux >> 52) & 0x7ffULL) - 1023;
    double m = fr3d_ppow_bits2d((ux & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    /* ln m = 2 atanh(t), t = (m - 1)/(m + 1), |t| <= 0.1716: odd series to t^25 */
    const double t = (m - 1.0) / (m + 1.0);
    const double t2 = t * t;
    double s = 0.04;                      /* 1/25 */
    s = s * t2 + 0.043478260869565216;    /* 1/23 */
    s = s * t2 + 0.047619047619047616;    /* 1/21 */
    s = s * t2 + 0.052631578947368418;    /* 1/19 */
    s = s * t2 + 0.058823529411764705;    /* 1/17 */
    s = s * t2 + 0.066666666666666666;    /* 1/15 */
    s = s * t2 + 0.076923076923076927;    /* 1/13 */
    s = s * t2 + 0.090909090909090912;    /* 1/11 */
    s = s * t2 + 0.11111111111111110;     /* 1/9 */
    s = s * t2 + 0.14285714285714285;     /* 1/7 */
    s = s * t2 + 0.20000000000000001;     /* 1/5 */
    s = s * t2 + 0.33333333333333331;     /* 1/3 */
    const double lnm = 2.0 * t + 2.0 * t * (t2 * s);
    const double ed = (double)e;
    /* z = y ln x.  The large part y (e LN2_HI) as an exact product p + pe (Dekker's two-product through Veltkamp
     * splitting: plain arithmetic, exact without an fma), the small parts added behind it. */
    const double big = ed * LN2_HI; /* exact: |e| < 2^11, LN2_HI has 21 trailing zero bits */
    const double z_hi = y * big;
    const double ya = y * 134217729.0, yh = ya - (ya - y), yl = y - yh;
    const double ba = big * 134217729.0, bh = ba - (ba - big), bl = big - bh;
    const double z_he = ((yh * bh - z_hi) + yh * bl + yl * bh) + yl * bl; /* y big = z_hi + z_he exactly */
    const double z_lo = y * (lnm + ed * LN2_LO) + z_he;
    /* exp(z_hi + z_lo): k = round((z_hi + z_lo)/ln 2), r = z - k ln 2 in [-0.35, 0.35] */
    const double k = fr3d_ppow_floor((z_hi + z_lo) * INV_LN2 + 0.5);
    const double r = ((z_hi - k * LN2_HI) + z_lo) - k * LN2_LO;
    double p = 1.6059043836821613e-10;    /* 1/13! */
    p = p * r + 2.08767569878681e-09;     /* 1/12! */
    p = p * r + 2.505210838544172e-08;    /* 1/11! */
    p = p * r + 2.755731922398589e-07;    /* 1/10! */
    p = p * r + 2.7557319223985893e-06;   /* 1/9! */
    p = p * r + 2.48015873015873e-05;     /* 1/8! */
    p = p * r + 1.984126984126984e-04;    /* 1/7! */
    p = p * r + 1.388888888888889e-03;    /* 1/6! */
    p = p * r + 8.333333333333333e-03;    /* 1/5! */
    p = p * r + 4.1666666666666664e-02;   /* 1/4! */
    p = p * r + 1.6666666666666666e-01;   /* 1/3! */
    p = p * r + 0.5;
    const double er = 1.0 + (r + r * r * p);
    const long long ki = (long long)k;
    /* scale by 2^k in two steps so that a subnormal result is formed by one rounding multiplication */
    const long long k1 = ki / 2, k2 = ki - k1;
    const double s1 = fr3d_ppow_bits2d((unsigned long long)(k1 + 1023) << 52);
    const double s2 = fr3d_ppow_bits2d((unsigned long long)(k2 + 1023) << 52);
    return (er * s1) * s2;
}

#endif
