/*
 * fr3d_oracle.c -- CPU restatement of flowreg3d's get_displacement path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see fr3d_oracle.h).  Build with -O2 -ffp-contract=off (no FMA
 * contraction, no fast-math) so the fp64/fp32 operation order below is what executes.
 *
 * Citations are relative to /root/reference/src/flowreg3d/.
 */
#include "fr3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The psi nonlinearities a (x + eps)^(a-1) (core/level_solver_3d.py:310,377) use the C library's pow -- what the
 * reference's Python floats do -- in the default build.  The `ppow` build (-DFR3D_ORACLE_PORTABLE_POW,
 * _build/libfr3d_oracle_ppow.so) takes a pow written in plain arithmetic instead, the same source the engine's
 * verification mode compiles for the GPU, so that the two sides can be compared BIT FOR BIT
 * (tests/test_gpu_verify_mode.py); it differs from the default build only through the last bits of psi. */
#ifdef FR3D_ORACLE_PORTABLE_POW
#include "../flowreg3d_amd/csrc/portable_pow.h"
#define PSI_POW(x, y) fr3d_ppow((x), (y))
#else
#define PSI_POW(x, y) pow((x), (y))
#endif

#define IDX3(z, y, x, Y, X) (((size_t)(z) * (size_t)(Y) + (size_t)(y)) * (size_t)(X) + (size_t)(x))

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* K1  fused Gauss (x) Keys-cubic separable resampler       util/resize_util_3D.py            */
/* ------------------------------------------------------------------------------------------ */

/* util/resize_util_3D.py:53-61, A = -0.75 (:5); evaluated in fp64 with libm pow like CPython */
static double cubic_keys(double x)
{
    const double A = -0.75;
    double ax = fabs(x);
    if (ax < 1.0) return (A + 2.0) * pow(ax, 3.0) - (A + 3.0) * pow(ax, 2.0) + 1.0;
    if (ax < 2.0) return A * pow(ax, 3.0) - 5.0 * A * pow(ax, 2.0) + 8.0 * A * ax - 4.0 * A;
    return 0.0;
}

/* util/resize_util_3D.py:64-73 (symmetric reflect: -1 -> 0, n -> n-1) */
static int reflect_idx(int j, int n)
{
    if (n <= 1) return 0;
    while (j < 0 || j >= n) {
        if (j < 0) j = -j - 1;
        else j = 2 * n - 1 - j;
    }
    return j;
}

/* NumPy's float32 np.exp as dispatched on AVX2/AVX512 x86 hosts (numpy/_core/src/umath/
 * loops_exponent_log.dispatch.c.src, NumPy 2.2): Cody-Waite reduction + P5/Q2 rational
 * minimax, all in fp32 with FMAs.  The Gaussian taps of the resampler go through it
 * (util/resize_util_3D.py:106), and libm's expf differs from it by 1 ulp often enough to move
 * the end-to-end flow by ~1e-5, so the restatement follows NumPy here.  Verified bit-identical
 * to np.exp on 2e6 random float32 inputs in [-20, 0] (tools/gen_golden.py environment). */
static float np_expf(float x)
{
    const float log2e = 0x1.715476p+0f, magic = 0x1.800000p+23f;
    const float c1 = -0x1.62e400p-1f, c2 = -0x1.7f7d1cp-20f;
    const float P0 = 9.999999999980870924916e-01f, P1 = 7.257664613233124478488e-01f,
                P2 = 2.473615434895520810817e-01f, P3 = 5.114512081637298353406e-02f,
                P4 = 6.757896990527504603057e-03f, P5 = 5.082762527590693718096e-04f;
    const float Q0 = 1.0f, Q1 = -2.742335390411667452936e-01f, Q2 = 2.159509375685829852307e-02f;
    if (x < -87.0f) return 0.0f; /* far below any tap the resampler evaluates */
    float q = x * log2e;
    q = (q + magic) - magic;
    float r = fmaf(q, c1, x);
    r = fmaf(q, c2, r);
    float num = fmaf(P5, r, P4);
    num = fmaf(num, r, P3);
    num = fmaf(num, r, P2);
    num = fmaf(num, r, P1);
    num = fmaf(num, r, P0);
    float den = fmaf(Q2, r, Q1);
    den = fmaf(den, r, Q0);
    return ldexpf(num / den, (int)q);
}

/* numpy's float32 pairwise add.reduce for n <= 128 (loops_utils.h.src: pairwise_sum) */
static float np_sum_f32(const float *a, int n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
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

int fr3d_oracle_resize_tables(int in_len, int out_len, double sigma, int *idx, float *wt)
{
    /* util/resize_util_3D.py:98-111 */
    double scale = (double)out_len / (double)in_len;
    int R;
    float *g;
    if (sigma <= 0.0) {
        R = 0;
        g = (float *)xmalloc(sizeof(float));
        g[0] = 1.0f;
    } else {
        R = (int)ceil(2.0 * sigma);
        int n = 2 * R + 1;
        g = (float *)xmalloc(sizeof(float) * (size_t)n);
        float sig32 = (float)sigma; /* fp32 array / python float -> fp32 (NEP 50) */
        for (int k = 0; k < n; k++) {
            float x = (float)(k - R);
            float q = x / sig32;
            float e = -0.5f * (q * q);
            g[k] = np_expf(e);
        }
        float s = np_sum_f32(g, n);
        for (int k = 0; k < n; k++) g[k] = g[k] / s;
    }
    int P = 2 * R + 4;
    if (!idx || !wt) {
        free(g);
        return P;
    }
    /* util/resize_util_3D.py:76-95 */
    for (int i = 0; i < out_len; i++) {
        double x = ((double)i + 0.5) / scale - 0.5;
        int left = (int)floor(x - 2.0) - R;
        float ssum = 0.0f;
        for (int p = 0; p < P; p++) {
            int j = left + p;
            idx[(size_t)i * P + p] = reflect_idx(j, in_len);
            double d = x - (double)j;
            float acc = 0.0f;
            for (int u = -R; u <= R; u++) {
                float cv = (float)cubic_keys(d - (double)u); /* weak python float -> fp32 */
                acc += g[u + R] * cv;
            }
            wt[(size_t)i * P + p] = acc;
            ssum += acc;
        }
        float inv = 1.0f / ssum;
        for (int p = 0; p < P; p++) wt[(size_t)i * P + p] *= inv;
    }
    free(g);
    return P;
}

void fr3d_oracle_resize3d(const float *src, int D, int H, int W, int od, int oh, int ow,
                          double sigma_coeff, float *dst)
{
    fr3d_oracle_resize3d_ex(src, D, H, W, od, oh, ow, sigma_coeff, 0, dst);
}

void fr3d_oracle_resize3d_ex(const float *src, int D, int H, int W, int od, int oh, int ow,
                             double sigma_coeff, int per_axis, float *dst)
{
    /* util/resize_util_3D.py:114-138 */
    double sz = (double)od / (double)D, sy = (double)oh / (double)H, sx = (double)ow / (double)W;
    double s = sx;
    if (sy < s) s = sy;
    if (sz < s) s = sz;
    double sig = (s < 1.0) ? (sigma_coeff / s) : 0.0;
    double sigx = sig, sigy = sig, sigz = sig;
    if (per_axis) { /* :120-123 */
        sigx = sx < 1.0 ? sigma_coeff / sx : 0.0;
        sigy = sy < 1.0 ? sigma_coeff / sy : 0.0;
        sigz = sz < 1.0 ? sigma_coeff / sz : 0.0;
    }

    int Px = fr3d_oracle_resize_tables(W, ow, sigx, NULL, NULL);
    int Py = fr3d_oracle_resize_tables(H, oh, sigy, NULL, NULL);
    int Pz = fr3d_oracle_resize_tables(D, od, sigz, NULL, NULL);
    int *ix = (int *)xmalloc(sizeof(int) * (size_t)ow * Px);
    int *iy = (int *)xmalloc(sizeof(int) * (size_t)oh * Py);
    int *iz = (int *)xmalloc(sizeof(int) * (size_t)od * Pz);
    float *wx = (float *)xmalloc(sizeof(float) * (size_t)ow * Px);
    float *wy = (float *)xmalloc(sizeof(float) * (size_t)oh * Py);
    float *wz = (float *)xmalloc(sizeof(float) * (size_t)od * Pz);
    fr3d_oracle_resize_tables(W, ow, sigx, ix, wx);
    fr3d_oracle_resize_tables(H, oh, sigy, iy, wy);
    fr3d_oracle_resize_tables(D, od, sigz, iz, wz);

    float *t1 = (float *)xmalloc(sizeof(float) * (size_t)D * H * ow);
    float *t2 = (float *)xmalloc(sizeof(float) * (size_t)D * oh * ow);
    /* :8-20  X pass */
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++) {
            const float *row = src + IDX3(z, y, 0, H, W);
            float *o = t1 + IDX3(z, y, 0, H, ow);
            for (int i = 0; i < ow; i++) {
                float a = 0.0f;
                for (int p = 0; p < Px; p++) a += row[ix[(size_t)i * Px + p]] * wx[(size_t)i * Px + p];
                o[i] = a;
            }
        }
    /* :23-35  Y pass */
    for (int z = 0; z < D; z++)
        for (int i = 0; i < oh; i++)
            for (int x = 0; x < ow; x++) {
                float a = 0.0f;
                for (int p = 0; p < Py; p++)
                    a += t1[IDX3(z, iy[(size_t)i * Py + p], x, H, ow)] * wy[(size_t)i * Py + p];
                t2[IDX3(z, i, x, oh, ow)] = a;
            }
    /* :38-50  Z pass */
    for (int i = 0; i < od; i++)
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++) {
                float a = 0.0f;
                for (int p = 0; p < Pz; p++)
                    a += t2[IDX3(iz[(size_t)i * Pz + p], y, x, oh, ow)] * wz[(size_t)i * Pz + p];
                dst[IDX3(i, y, x, oh, ow)] = a;
            }
    free(t1); free(t2); free(ix); free(iy); free(iz); free(wx); free(wy); free(wz);
}

/* resize of one channel of a (D,H,W,C) fp64 array -> contiguous fp64 (od,oh,ow); fp32 inside
 * (util/resize_util_3D.py:116 astype(float32), :156 astype(img.dtype)) */
static void resize_chan_f64(const double *src, int D, int H, int W, int C, int ch, int od, int oh,
                            int ow, double *dst)
{
    size_t nin = (size_t)D * H * W, nout = (size_t)od * oh * ow;
    float *a = (float *)xmalloc(sizeof(float) * nin);
    float *b = (float *)xmalloc(sizeof(float) * nout);
    for (size_t i = 0; i < nin; i++) a[i] = (float)src[i * (size_t)C + (size_t)ch];
    fr3d_oracle_resize3d(a, D, H, W, od, oh, ow, 0.6, b);
    for (size_t i = 0; i < nout; i++) dst[i] = (double)b[i];
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------------ */
/* K2  backward warp: scipy.ndimage.map_coordinates(order, mode="nearest")                    */
/*     core/optical_flow_3d.py:22-74 ; SciPy 1.15 _interpolation.py (_prepad_for_spline_filter */
/*     npad=12 'edge', spline_filter -> float64) and ni_splines.c (reflect initialisation)       */
/* ------------------------------------------------------------------------------------------ */

static void spline_line_order3(double *c, int n, int stride)
{
    /* ni_splines.c apply_filter(): gain, causal (mirror init), anticausal (mirror init) */
    const double z = sqrt(3.0) - 2.0;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    if (n < 2) return;
    for (int i = 0; i < n; i++) c[(size_t)i * stride] *= gain;
    /* _init_causal_reflect: SciPy maps mode='nearest' to the half-sample-symmetric ("reflect")
     * initialisation (checked against scipy.ndimage.spline_filter1d 1.15.3) */
    {
        /* in place on c[0] exactly like SciPy: the last term (i = n-1) reads the partially
         * accumulated c[0], an O(z^2n) quirk that is visible for lines shorter than ~12 */
        double z_i = z;
        const double z_n = pow(z, (double)n);
        const double c00 = c[0];
        c[0] = c[0] + z_n * c[(size_t)(n - 1) * stride];
        for (int i = 1; i < n; i++) {
            c[0] += z_i * (c[(size_t)i * stride] + z_n * c[(size_t)(n - 1 - i) * stride]);
            z_i *= z;
        }
        c[0] *= z / (1.0 - z_n * z_n);
        c[0] += c00;
    }
    for (int i = 1; i < n; i++) c[(size_t)i * stride] += z * c[(size_t)(i - 1) * stride];
    /* _init_anticausal_reflect */
    c[(size_t)(n - 1) * stride] *= z / (z - 1.0);
    for (int i = n - 2; i >= 0; i--)
        c[(size_t)i * stride] = z * (c[(size_t)(i + 1) * stride] - c[(size_t)i * stride]);
}

void fr3d_oracle_spline_filter3(double *c, int Z, int Y, int X)
{
    /* spline_filter: for axis in range(ndim): spline_filter1d(...) */
    if (Z > 1)
        for (int y = 0; y < Y; y++)
            for (int x = 0; x < X; x++) spline_line_order3(c + IDX3(0, y, x, Y, X), Z, Y * X);
    if (Y > 1)
        for (int z = 0; z < Z; z++)
            for (int x = 0; x < X; x++) spline_line_order3(c + IDX3(z, 0, x, Y, X), Y, X);
    if (X > 1)
        for (int z = 0; z < Z; z++)
            for (int y = 0; y < Y; y++) spline_line_order3(c + IDX3(z, y, 0, Y, X), X, 1);
}

/* ni_splines.c get_spline_interpolation_weights(), orders 1 and 3 */
static void spline_weights(double x, int order, double *wts)
{
    x -= floor(x);
    double y = x, z = 1.0 - x;
    if (order == 1) {
        wts[0] = 1.0 - x;
    } else {
        wts[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        wts[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        wts[0] = z * z * z / 6.0;
    }
    wts[order] = 1.0;
    for (int i = 0; i < order; i++) wts[order] -= wts[i];
}

static int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* Output conversion of the executor's final warp of a RAW volume (parallelization/sequential_3d.py:
 * 153-170): scipy.ndimage.map_coordinates allocates its output in the INPUT's dtype, so the interpolated
 * double is converted by NI_GeometricTransform's output cast (ni_interpolation.c CASE_INTERP_OUT_*:
 * unsigned: t > 0 ? t + 0.5 : 0, clamped to the type's maximum, truncated; signed: t +- 0.5, clamped,
 * truncated; float/double: plain cast), then stored into the float32 `warped` array
 * (core/optical_flow_3d.py:59-66) and finally into `registered` of the batch's dtype.  Out-of-bounds
 * voxels take f1 through float32 (:69-70) and NumPy's float -> integer cast (truncation).
 * out_dtype: 0 f32, 1 f64, 2 u8, 3 u16, 4 i16. */
static void store_typed(void *out, size_t idx, int out_dtype, double t, int interpolated)
{
    if (out_dtype == 0) { ((float *)out)[idx] = (float)t; return; }
    if (out_dtype == 1) { ((double *)out)[idx] = (double)(float)t; return; }
    if (!interpolated) {
        float f = (float)t;  /* warped[...] = f1 (float32), then registered[t] = warped */
        if (out_dtype == 2) ((unsigned char *)out)[idx] = (unsigned char)f;
        else if (out_dtype == 3) ((unsigned short *)out)[idx] = (unsigned short)f;
        else ((short *)out)[idx] = (short)f;
        return;
    }
    if (out_dtype == 2 || out_dtype == 3) {
        const double mx = out_dtype == 2 ? 255.0 : 65535.0;
        t = t > 0 ? t + 0.5 : 0;
        t = t > mx ? mx : t;
        t = t < 0 ? 0 : t;
        if (out_dtype == 2) ((unsigned char *)out)[idx] = (unsigned char)t;
        else ((unsigned short *)out)[idx] = (unsigned short)t;
    } else {
        t = t > 0 ? t + 0.5 : t - 0.5;
        t = t > 32767.0 ? 32767.0 : t;
        t = t < -32768.0 ? -32768.0 : t;
        ((short *)out)[idx] = (short)t;
    }
}

void fr3d_oracle_imregister(const double *f2, const double *u, const double *v, const double *w,
                            const double *f1, int Z, int Y, int X, int C, int order, float *out)
{
    fr3d_oracle_imregister_typed(f2, u, v, w, f1, Z, Y, X, C, order, 0, out);
}

void fr3d_oracle_imregister_typed(const double *f2, const double *u, const double *v, const double *w,
                                  const double *f1, int Z, int Y, int X, int C, int order, int out_dtype,
                                  void *out)
{
    const int npad = (order > 1) ? 12 : 0;
    const int PZ = Z + 2 * npad, PY = Y + 2 * npad, PX = X + 2 * npad;
    const size_t nvox = (size_t)Z * Y * X;
    double *coef = (double *)xmalloc(sizeof(double) * (size_t)PZ * PY * PX);

    for (int c = 0; c < C; c++) {
        /* np.pad(input, 12, mode='edge') then spline_filter(..., output=float64, mode='nearest') */
        for (int z = 0; z < PZ; z++)
            for (int y = 0; y < PY; y++)
                for (int x = 0; x < PX; x++) {
                    int sz = clampi(z - npad, Z), sy = clampi(y - npad, Y), sx = clampi(x - npad, X);
                    coef[IDX3(z, y, x, PY, PX)] = f2[IDX3(sz, sy, sx, Y, X) * C + c];
                }
        if (order > 1) fr3d_oracle_spline_filter3(coef, PZ, PY, PX);

        for (int z = 0; z < Z; z++)
            for (int y = 0; y < Y; y++)
                for (int x = 0; x < X; x++) {
                    size_t i = IDX3(z, y, x, Y, X);
                    /* core/optical_flow_3d.py:32-34: (grid + disp).astype(float32) */
                    float mx = (float)((double)x + u[i]);
                    float my = (float)((double)y + v[i]);
                    float mz = (float)((double)z + w[i]);
                    /* :37-44 */
                    int oob = (mx < 0.0f) || (mx >= (float)X) || (my < 0.0f) || (my >= (float)Y) ||
                              (mz < 0.0f) || (mz >= (float)Z);
                    if (oob) {
                        /* :69-70 */
                        store_typed(out, i * C + c, out_dtype, f1[i * C + c], 0);
                        continue;
                    }
                    /* :47-49 np.clip in fp32 (in-bounds points are unchanged except > N-1) */
                    float cx = mx > (float)(X - 1) ? (float)(X - 1) : mx;
                    float cy = my > (float)(Y - 1) ? (float)(Y - 1) : my;
                    float cz = mz > (float)(Z - 1) ? (float)(Z - 1) : mz;
                    double cc[3] = {(double)cz + npad, (double)cy + npad, (double)cx + npad};
                    int start[3];
                    double wt[3][4];
                    for (int d = 0; d < 3; d++) {
                        start[d] = (int)floor(cc[d]) - order / 2;
                        spline_weights(cc[d], order, wt[d]);
                    }
                    double t = 0.0;
                    for (int a = 0; a <= order; a++) {
                        int zi = clampi(start[0] + a, PZ);
                        for (int b = 0; b <= order; b++) {
                            int yi = clampi(start[1] + b, PY);
                            for (int e = 0; e <= order; e++) {
                                int xi = clampi(start[2] + e, PX);
                                double cf = coef[IDX3(zi, yi, xi, PY, PX)];
                                cf *= wt[0][a];
                                cf *= wt[1][b];
                                cf *= wt[2][e];
                                t += cf;
                            }
                        }
                    }
                    store_typed(out, i * C + c, out_dtype, t, 1);
                }
    }
    (void)nvox;
    free(coef);
}

/* f-4  BatchMotionCorrector._update_reference       motion_correction/compensate_recording_3D.py:395-429
 * batch_proc (T,Z,Y,X,C) fp64, flows (T,Z,Y,X,3) fp32, ref_proc (Z,Y,X,C) fp64 -> new_ref (Z,Y,X,C) fp64.
 * Per channel: the last n = min(100,T) volumes are warped one channel at a time by imregister_wrapper
 * (fp32 result) into an fp64 stack whose mean over the stack axis is taken: np.mean(axis=0) on a
 * C-contiguous (n,Z,Y,X) array adds the volumes in order t = 0..n-1 and divides the sum by n. */
void fr3d_oracle_update_reference(const double *batch_proc, const float *flows, const double *ref_proc, int T,
                                  int Z, int Y, int X, int C, int order, double *new_ref)
{
    const size_t nv = (size_t)Z * Y * X;
    const int n_ref = T < 100 ? T : 100;
    if (n_ref < 1) return;
    const int start = T - n_ref;
    double *vol = (double *)xmalloc(sizeof(double) * nv), *refc = (double *)xmalloc(sizeof(double) * nv);
    double *u = (double *)xmalloc(sizeof(double) * nv), *v = (double *)xmalloc(sizeof(double) * nv),
           *w = (double *)xmalloc(sizeof(double) * nv), *acc = (double *)xmalloc(sizeof(double) * nv);
    float *out = (float *)xmalloc(sizeof(float) * nv);
    for (int c = 0; c < C; c++) {
        for (size_t q = 0; q < nv; q++) { refc[q] = ref_proc[q * C + c]; acc[q] = 0.0; }
        for (int t = 0; t < n_ref; t++) {
            const double *bp = batch_proc + (size_t)(start + t) * nv * C;
            const float *fl = flows + (size_t)(start + t) * nv * 3;
            for (size_t q = 0; q < nv; q++) {
                vol[q] = bp[q * C + c];
                u[q] = (double)fl[q * 3 + 0]; v[q] = (double)fl[q * 3 + 1]; w[q] = (double)fl[q * 3 + 2];
            }
            fr3d_oracle_imregister(vol, u, v, w, refc, Z, Y, X, 1, order, out);
            for (size_t q = 0; q < nv; q++) acc[q] = (t == 0) ? (double)out[q] : acc[q] + (double)out[q];
        }
        for (size_t q = 0; q < nv; q++) new_ref[q * C + c] = acc[q] / (double)n_ref;
    }
    free(vol); free(refc); free(u); free(v); free(w); free(acc); free(out);
}

/* ------------------------------------------------------------------------------------------ */
/* K3  gradient-constancy motion tensor                 core/optical_flow_3d.py:92-152         */
/* ------------------------------------------------------------------------------------------ */

void fr3d_oracle_motion_tensor_gc(const double *f1, const double *f2, int Z, int Y, int X,
                                  double hz, double hy, double hx, double *const J[10])
{
    const int PZ = Z + 2, PY = Y + 2, PX = X + 2;
    const size_t np_ = (size_t)PZ * PY * PX, n = (size_t)Z * Y * X;
    for (int a = 0; a < 10; a++) memset(J[a], 0, sizeof(double) * np_);

    /* First derivatives on the interior (np.gradient central differences on the symmetric-padded
     * volumes, :93-100), stored unpadded; the symmetric re-pad of :101-104 is a clamped read. */
    double *fx = (double *)xmalloc(sizeof(double) * n);
    double *fy = (double *)xmalloc(sizeof(double) * n);
    double *ft = (double *)xmalloc(sizeof(double) * n);
    double *fz = (double *)xmalloc(sizeof(double) * n);
#define F(arr, z, y, x) arr[IDX3(clampi(z, Z), clampi(y, Y), clampi(x, X), Y, X)]
    for (int z = 0; z < Z; z++)
        for (int y = 0; y < Y; y++)
            for (int x = 0; x < X; x++) {
                size_t i = IDX3(z, y, x, Y, X);
                double gx1 = (F(f1, z, y, x + 1) - F(f1, z, y, x - 1)) / (2.0 * hx);
                double gx2 = (F(f2, z, y, x + 1) - F(f2, z, y, x - 1)) / (2.0 * hx);
                double gy1 = (F(f1, z, y + 1, x) - F(f1, z, y - 1, x)) / (2.0 * hy);
                double gy2 = (F(f2, z, y + 1, x) - F(f2, z, y - 1, x)) / (2.0 * hy);
                double gz1 = (F(f1, z + 1, y, x) - F(f1, z - 1, y, x)) / (2.0 * hz);
                double gz2 = (F(f2, z + 1, y, x) - F(f2, z - 1, y, x)) / (2.0 * hz);
                fx[i] = 0.5 * (gx1 + gx2);
                fy[i] = 0.5 * (gy1 + gy2);
                fz[i] = 0.5 * (gz1 + gz2);
                ft[i] = f2[i] - f1[i];
            }
    const double hx2 = pow(hx, 2.0), hy2 = pow(hy, 2.0), hz2 = pow(hz, 2.0);
    for (int z = 0; z < Z; z++)
        for (int y = 0; y < Y; y++)
            for (int x = 0; x < X; x++) {
                /* :106-113 second np.gradient (central, on the re-padded fields) */
                double fxy = (F(fx, z, y + 1, x) - F(fx, z, y - 1, x)) / (2.0 * hy);
                double fxz = (F(fx, z + 1, y, x) - F(fx, z - 1, y, x)) / (2.0 * hz);
                double fyz = (F(fy, z + 1, y, x) - F(fy, z - 1, y, x)) / (2.0 * hz);
                double fzt = (F(ft, z + 1, y, x) - F(ft, z - 1, y, x)) / (2.0 * hz);
                double fyt = (F(ft, z, y + 1, x) - F(ft, z, y - 1, x)) / (2.0 * hy);
                double fxt = (F(ft, z, y, x + 1) - F(ft, z, y, x - 1)) / (2.0 * hx);
                /* :115-128 */
                double fxx1 = (F(f1, z, y, x - 1) - 2.0 * F(f1, z, y, x) + F(f1, z, y, x + 1)) / hx2;
                double fxx2 = (F(f2, z, y, x - 1) - 2.0 * F(f2, z, y, x) + F(f2, z, y, x + 1)) / hx2;
                double fyy1 = (F(f1, z, y - 1, x) - 2.0 * F(f1, z, y, x) + F(f1, z, y + 1, x)) / hy2;
                double fyy2 = (F(f2, z, y - 1, x) - 2.0 * F(f2, z, y, x) + F(f2, z, y + 1, x)) / hy2;
                double fzz1 = (F(f1, z - 1, y, x) - 2.0 * F(f1, z, y, x) + F(f1, z + 1, y, x)) / hz2;
                double fzz2 = (F(f2, z - 1, y, x) - 2.0 * F(f2, z, y, x) + F(f2, z + 1, y, x)) / hz2;
                double fxx = 0.5 * (fxx1 + fxx2);
                double fyy = 0.5 * (fyy1 + fyy2);
                double fzz = 0.5 * (fzz1 + fzz2);
                /* :130-132 */
                double sxn = sqrt(fxx * fxx + fxy * fxy + fxz * fxz);
                double syn = sqrt(fxy * fxy + fyy * fyy + fyz * fyz);
                double szn = sqrt(fxz * fxz + fyz * fyz + fzz * fzz);
                double rx = 1.0 / (sxn * sxn + 1e-6);
                double ry = 1.0 / (syn * syn + 1e-6);
                double rz = 1.0 / (szn * szn + 1e-6);
                size_t o = IDX3(z + 1, y + 1, x + 1, PY, PX);
                /* :134-143 */
                J[0][o] = rx * (fxx * fxx) + ry * (fxy * fxy) + rz * (fxz * fxz);
                J[1][o] = rx * (fxy * fxy) + ry * (fyy * fyy) + rz * (fyz * fyz);
                J[2][o] = rx * (fxz * fxz) + ry * (fyz * fyz) + rz * (fzz * fzz);
                J[3][o] = rx * (fxt * fxt) + ry * (fyt * fyt) + rz * (fzt * fzt);
                J[4][o] = rx * fxx * fxy + ry * fxy * fyy + rz * fxz * fyz;
                J[5][o] = rx * fxx * fxz + ry * fxy * fyz + rz * fxz * fzz;
                J[6][o] = rx * fxy * fxz + ry * fyy * fyz + rz * fyz * fzz;
                J[7][o] = rx * fxx * fxt + ry * fxy * fyt + rz * fxz * fzt;
                J[8][o] = rx * fxy * fxt + ry * fyy * fyt + rz * fyz * fzt;
                J[9][o] = rx * fxz * fxt + ry * fyz * fyt + rz * fzz * fzt;
            }
#undef F
    free(fx); free(fy); free(fz); free(ft);
}

/* ------------------------------------------------------------------------------------------ */
/* K4-K7  lagged-nonlinearity lexicographic SOR        core/level_solver_3d.py:246-546         */
/* ------------------------------------------------------------------------------------------ */

/* core/level_solver_3d.py:246-259 */
static void set_boundary_3d(double *f, int p, int m, int n)
{
    for (int k = 0; k < p; k++) {
        for (int i = 0; i < n; i++) {
            f[IDX3(k, 0, i, m, n)] = f[IDX3(k, 1, i, m, n)];
            f[IDX3(k, m - 1, i, m, n)] = f[IDX3(k, m - 2, i, m, n)];
        }
        for (int j = 0; j < m; j++) {
            f[IDX3(k, j, 0, m, n)] = f[IDX3(k, j, 1, m, n)];
            f[IDX3(k, j, n - 1, m, n)] = f[IDX3(k, j, n - 2, m, n)];
        }
    }
    for (int j = 0; j < m; j++)
        for (int i = 0; i < n; i++) {
            f[IDX3(0, j, i, m, n)] = f[IDX3(1, j, i, m, n)];
            f[IDX3(p - 1, j, i, m, n)] = f[IDX3(p - 2, j, i, m, n)];
        }
}

/* core/level_solver_3d.py:262-311 */
static void nonlinearity_smoothness_3d(double *psi, const double *u, const double *du,
                                       const double *v, const double *dv, const double *w,
                                       const double *dw, int p, int m, int n, double a, double hx,
                                       double hy, double hz)
{
    const double eps = 1e-5;
    size_t tot = (size_t)p * m * n;
    double *uu = (double *)xmalloc(sizeof(double) * tot);
    double *vv = (double *)xmalloc(sizeof(double) * tot);
    double *ww = (double *)xmalloc(sizeof(double) * tot);
    for (size_t i = 0; i < tot; i++) {
        uu[i] = u[i] + du[i];
        vv[i] = v[i] + dv[i];
        ww[i] = w[i] + dw[i];
    }
    for (int k = 0; k < p; k++)
        for (int j = 0; j < m; j++)
            for (int i = 0; i < n; i++) {
                int ixm = i > 0 ? i - 1 : 0, ixp = i < n - 1 ? i + 1 : n - 1;
                int jym = j > 0 ? j - 1 : 0, jyp = j < m - 1 ? j + 1 : m - 1;
                int kzm = k > 0 ? k - 1 : 0, kzp = k < p - 1 ? k + 1 : p - 1;
                double ux = (uu[IDX3(k, j, ixp, m, n)] - uu[IDX3(k, j, ixm, m, n)]) / (2 * hx);
                double uy = (uu[IDX3(k, jyp, i, m, n)] - uu[IDX3(k, jym, i, m, n)]) / (2 * hy);
                double uz = (uu[IDX3(kzp, j, i, m, n)] - uu[IDX3(kzm, j, i, m, n)]) / (2 * hz);
                double vx = (vv[IDX3(k, j, ixp, m, n)] - vv[IDX3(k, j, ixm, m, n)]) / (2 * hx);
                double vy = (vv[IDX3(k, jyp, i, m, n)] - vv[IDX3(k, jym, i, m, n)]) / (2 * hy);
                double vz = (vv[IDX3(kzp, j, i, m, n)] - vv[IDX3(kzm, j, i, m, n)]) / (2 * hz);
                double wx = (ww[IDX3(k, j, ixp, m, n)] - ww[IDX3(k, j, ixm, m, n)]) / (2 * hx);
                double wy = (ww[IDX3(k, jyp, i, m, n)] - ww[IDX3(k, jym, i, m, n)]) / (2 * hy);
                double wz = (ww[IDX3(kzp, j, i, m, n)] - ww[IDX3(kzm, j, i, m, n)]) / (2 * hz);
                double g = ux * ux + uy * uy + uz * uz + vx * vx + vy * vy + vz * vz + wx * wx +
                           wy * wy + wz * wz;
                if (g < 0.0) g = 0.0;
                psi[IDX3(k, j, i, m, n)] = a * PSI_POW(g + eps, a - 1.0);
            }
    free(uu); free(vv); free(ww);
}

void fr3d_oracle_compute_flow_3d(const double *const J[10], const double *weight,
                                 const double *u, const double *v, const double *w,
                                 int p, int m, int n, int C,
                                 double alpha_x, double alpha_y, double alpha_z,
                                 int iterations, int update_lag, const double *a_data,
                                 double a_smooth, double hx, double hy, double hz, double *out)
{
    const double *J11 = J[0], *J22 = J[1], *J33 = J[2], *J44 = J[3], *J12 = J[4], *J13 = J[5],
                 *J23 = J[6], *J14 = J[7], *J24 = J[8], *J34 = J[9];
    const size_t tot = (size_t)p * m * n;
    double *du = (double *)calloc(tot, sizeof(double));
    double *dv = (double *)calloc(tot, sizeof(double));
    double *dw = (double *)calloc(tot, sizeof(double));
    double *psi_s = (double *)xmalloc(sizeof(double) * tot);
    double *psi = (double *)xmalloc(sizeof(double) * tot * C);
    if (!du || !dv || !dw) abort();
    for (size_t i = 0; i < tot; i++) psi_s[i] = 1.0;
    for (size_t i = 0; i < tot * C; i++) psi[i] = 1.0;
    const double alpha[3] = {alpha_x, alpha_y, alpha_z};
    const double OMEGA = 1.95, eps = 1e-6;

    for (int it = 0; it < iterations; it++) {
        if (a_smooth != 1.0)
            nonlinearity_smoothness_3d(psi_s, u, du, v, dv, w, dw, p, m, n, a_smooth, hx, hy, hz);
        if (it % update_lag == 0) {
            for (int c = 0; c < C; c++) {
                double adc = a_data[c];
                if (adc != 1.0) {
                    for (size_t q = 0; q < tot; q++) {
                        size_t qc = q * C + c;
                        double val = J11[qc] * du[q] * du[q] + J22[qc] * dv[q] * dv[q] +
                                     J33[qc] * dw[q] * dw[q] + 2.0 * J12[qc] * du[q] * dv[q] +
                                     2.0 * J13[qc] * du[q] * dw[q] + 2.0 * J23[qc] * dv[q] * dw[q] +
                                     2.0 * J14[qc] * du[q] + 2.0 * J24[qc] * dv[q] +
                                     2.0 * J34[qc] * dw[q] + J44[qc];
                        if (val < 0.0) val = 0.0;
                        psi[qc] = adc * PSI_POW(val + eps, adc - 1.0);
                    }
                }
            }
        }
        set_boundary_3d(du, p, m, n);
        set_boundary_3d(dv, p, m, n);
        set_boundary_3d(dw, p, m, n);

        for (int k = 1; k < p - 1; k++)
            for (int j = 1; j < m - 1; j++)
                for (int i = 1; i < n - 1; i++) {
                    double denom_u = 0.0, denom_v = 0.0, denom_w = 0.0;
                    double num_u = 0.0, num_v = 0.0, num_w = 0.0;
                    size_t c0 = IDX3(k, j, i, m, n);
                    size_t km = IDX3(k - 1, j, i, m, n), kp = IDX3(k + 1, j, i, m, n);
                    size_t jm = IDX3(k, j - 1, i, m, n), jp = IDX3(k, j + 1, i, m, n);
                    size_t im = c0 - 1, ip = c0 + 1;
                    if (a_smooth != 1.0) {
                        const size_t nb[6] = {km, kp, jm, jp, im, ip};
                        const double sc[6] = {alpha[2] / (hz * hz), alpha[2] / (hz * hz),
                                              alpha[1] / (hy * hy), alpha[1] / (hy * hy),
                                              alpha[0] / (hx * hx), alpha[0] / (hx * hx)};
                        for (int q = 0; q < 6; q++) {
                            double tmp = 0.5 * (psi_s[c0] + psi_s[nb[q]]) * sc[q];
                            num_u += tmp * (u[nb[q]] + du[nb[q]] - u[c0]);
                            denom_u += tmp;
                            num_v += tmp * (v[nb[q]] + dv[nb[q]] - v[c0]);
                            denom_v += tmp;
                            num_w += tmp * (w[nb[q]] + dw[nb[q]] - w[c0]);
                            denom_w += tmp;
                        }
                    } else {
                        double ax = alpha[0] / (hx * hx);
                        double ay = alpha[1] / (hy * hy);
                        double az = alpha[2] / (hz * hz);
                        num_u += ax * (u[ip] + du[ip] + u[im] + du[im] - 2 * u[c0]);
                        denom_u += 2 * ax;
                        num_v += ax * (v[ip] + dv[ip] + v[im] + dv[im] - 2 * v[c0]);
                        denom_v += 2 * ax;
                        num_w += ax * (w[ip] + dw[ip] + w[im] + dw[im] - 2 * w[c0]);
                        denom_w += 2 * ax;
                        num_u += ay * (u[jp] + du[jp] + u[jm] + du[jm] - 2 * u[c0]);
                        denom_u += 2 * ay;
                        num_v += ay * (v[jp] + dv[jp] + v[jm] + dv[jm] - 2 * v[c0]);
                        denom_v += 2 * ay;
                        num_w += ay * (w[jp] + dw[jp] + w[jm] + dw[jm] - 2 * w[c0]);
                        denom_w += 2 * ay;
                        num_u += az * (u[kp] + du[kp] + u[km] + du[km] - 2 * u[c0]);
                        denom_u += 2 * az;
                        num_v += az * (v[kp] + dv[kp] + v[km] + dv[km] - 2 * v[c0]);
                        denom_v += 2 * az;
                        num_w += az * (w[kp] + dw[kp] + w[km] + dw[km] - 2 * w[c0]);
                        denom_w += 2 * az;
                    }
                    for (int c = 0; c < C; c++) {
                        double ww = weight[c0 * C + c];
                        if (a_data[c] != 1.0) ww *= psi[c0 * C + c];
                        denom_u += ww * J11[c0 * C + c];
                        denom_v += ww * J22[c0 * C + c];
                        denom_w += ww * J33[c0 * C + c];
                    }
                    double num_u2 = num_u;
                    for (int c = 0; c < C; c++) {
                        double ww = weight[c0 * C + c];
                        if (a_data[c] != 1.0) ww *= psi[c0 * C + c];
                        num_u2 -= ww * (J14[c0 * C + c] + J12[c0 * C + c] * dv[c0] +
                                        J13[c0 * C + c] * dw[c0]);
                    }
                    double du_kp1 = denom_u != 0.0 ? num_u2 / denom_u : 0.0;
                    du[c0] = (1.0 - OMEGA) * du[c0] + OMEGA * du_kp1;

                    double num_v2 = num_v;
                    for (int c = 0; c < C; c++) {
                        double ww = weight[c0 * C + c];
                        if (a_data[c] != 1.0) ww *= psi[c0 * C + c];
                        num_v2 -= ww * (J24[c0 * C + c] + J12[c0 * C + c] * du[c0] +
                                        J23[c0 * C + c] * dw[c0]);
                    }
                    double dv_kp1 = denom_v != 0.0 ? num_v2 / denom_v : 0.0;
                    dv[c0] = (1.0 - OMEGA) * dv[c0] + OMEGA * dv_kp1;

                    double num_w2 = num_w;
                    for (int c = 0; c < C; c++) {
                        double ww = weight[c0 * C + c];
                        if (a_data[c] != 1.0) ww *= psi[c0 * C + c];
                        num_w2 -= ww * (J34[c0 * C + c] + J13[c0 * C + c] * du[c0] +
                                        J23[c0 * C + c] * dv[c0]);
                    }
                    double dw_kp1 = denom_w != 0.0 ? num_w2 / denom_w : 0.0;
                    dw[c0] = (1.0 - OMEGA) * dw[c0] + OMEGA * dw_kp1;
                }
    }
    for (size_t q = 0; q < tot; q++) {
        out[q * 3 + 0] = du[q];
        out[q * 3 + 1] = dv[q];
        out[q * 3 + 2] = dw[q];
    }
    (void)J44;
    free(du); free(dv); free(dw); free(psi_s); free(psi);
}

/* debugging aid of the tests: FR3D_ORACLE_DUMP=<dir> writes per-level intermediate arrays as raw float64 */
#include <stdio.h>
static void dbg_dump(const char *tag, int level, int ch, const double *a, size_t n)
{
    const char *dir = getenv("FR3D_ORACLE_DUMP");
    if (!dir) return;
    char path[512];
    snprintf(path, sizeof(path), "%s/o_L%d_%s%d.bin", dir, level, tag, ch);
    FILE *f = fopen(path, "wb");
    if (!f) return;
    fwrite(a, sizeof(double), n, f);
    fclose(f);
}

/* ------------------------------------------------------------------------------------------ */
/* K8  exact 5x5x5 median, mirror boundary       scipy.ndimage.median_filter (rank 62 of 125)  */
/* ------------------------------------------------------------------------------------------ */

static int mirror_idx(int i, int n)
{
    /* scipy 'mirror': d c b | a b c d | c b a  (whole-sample symmetric, edge not repeated) */
    if (n == 1) return 0;
    int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

static double select_k(double *a, int n, int k)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        double piv = a[(lo + hi) >> 1];
        int i = lo, j = hi;
        while (i <= j) {
            while (a[i] < piv) i++;
            while (a[j] > piv) j--;
            if (i <= j) {
                double t = a[i]; a[i] = a[j]; a[j] = t;
                i++; j--;
            }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return a[k];
}

void fr3d_oracle_median5(const double *in, int Z, int Y, int X, double *out)
{
    double buf[125];
    int zi[5], yi[5], xi[5];
    for (int z = 0; z < Z; z++) {
        for (int a = 0; a < 5; a++) zi[a] = mirror_idx(z + a - 2, Z);
        for (int y = 0; y < Y; y++) {
            for (int a = 0; a < 5; a++) yi[a] = mirror_idx(y + a - 2, Y);
            for (int x = 0; x < X; x++) {
                for (int a = 0; a < 5; a++) xi[a] = mirror_idx(x + a - 2, X);
                int q = 0;
                for (int a = 0; a < 5; a++)
                    for (int b = 0; b < 5; b++) {
                        const double *row = in + IDX3(zi[a], yi[b], 0, Y, X);
                        for (int c = 0; c < 5; c++) buf[q++] = row[xi[c]];
                    }
                out[IDX3(z, y, x, Y, X)] = select_k(buf, 125, 62);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Driver                                        core/optical_flow_3d.py:77-89, 319-542        */
/* ------------------------------------------------------------------------------------------ */

/* Python round(): half to even.  nearbyint() under the default rounding mode does the same. */
static long py_round(double x) { return (long)nearbyint(x); }

int fr3d_oracle_warping_depth(double eta, int levels, int p, int m, int n)
{
    double min_dim = (double)(p < m ? (p < n ? p : n) : (m < n ? m : n));
    int depth = 0;
    for (int q = 0; q < levels; q++) {
        depth += 1;
        min_dim *= eta;
        if (py_round(min_dim) < 10) break;
    }
    return depth;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void schedule_levels(int p, int m, int n, double eta, int levels, int *min_level,
                            int *mlz, int *mly, int *mlx)
{
    /* :389-399 */
    int max_level_z = fr3d_oracle_warping_depth(eta, levels, p, m, n);
    int max_level_y = fr3d_oracle_warping_depth(eta, levels, m, n, p);
    int max_level_x = fr3d_oracle_warping_depth(eta, levels, n, p, m);
    int max_level = imin(max_level_x, imin(max_level_y, max_level_z)) * 4;
    max_level_z = imin(max_level_z, max_level);
    max_level_y = imin(max_level_y, max_level);
    max_level_x = imin(max_level_x, max_level);
    int top = imax(max_level_x, imax(max_level_y, max_level_z));
    if (top <= *min_level) *min_level = top - 1;
    if (*min_level < 0) *min_level = 0;
    *mlz = max_level_z; *mly = max_level_y; *mlx = max_level_x;
}

static void level_size_of(int p, int m, int n, double eta, int i, int mlz, int mly, int mlx,
                          int *sz)
{
    /* :404-408 */
    sz[0] = (int)py_round((double)p * pow(eta, (double)imin(i, mlz)));
    sz[1] = (int)py_round((double)m * pow(eta, (double)imin(i, mly)));
    sz[2] = (int)py_round((double)n * pow(eta, (double)imin(i, mlx)));
}

int fr3d_oracle_schedule(int p, int m, int n, double eta, int levels, int min_level, int *sizes,
                         int max_out, int *min_level_eff)
{
    int mlz, mly, mlx;
    schedule_levels(p, m, n, eta, levels, &min_level, &mlz, &mly, &mlx);
    int top = imax(mlx, imax(mly, mlz));
    int cnt = 0;
    for (int i = top; i >= min_level; i--) {
        if (cnt < max_out) level_size_of(p, m, n, eta, i, mlz, mly, mlx, sizes + 3 * cnt);
        cnt++;
    }
    if (min_level_eff) *min_level_eff = min_level;
    return cnt;
}

/* np.pad(f, 1, mode="edge") of a (z,y,x) array -> (z+2,y+2,x+2)   (:88-89) */
static void add_boundary(const double *src, int z, int y, int x, double *dst)
{
    for (int a = 0; a < z + 2; a++)
        for (int b = 0; b < y + 2; b++)
            for (int c = 0; c < x + 2; c++)
                dst[IDX3(a, b, c, y + 2, x + 2)] =
                    src[IDX3(clampi(a - 1, z), clampi(b - 1, y), clampi(c - 1, x), y, x)];
}

int fr3d_oracle_get_displacement(const double *fixed, const double *moving, int p, int m, int n,
                                 int C, const double *alpha3, int update_lag, int iterations,
                                 int min_level, int levels, double eta, double a_smooth,
                                 const double *a_data, const double *uvw, const double *weight,
                                 double *flow)
{
    if (p < 1 || m < 1 || n < 1 || C < 1 || iterations < 0 || update_lag < 1) return 1;
    int mlz, mly, mlx;
    schedule_levels(p, m, n, eta, levels, &min_level, &mlz, &mly, &mlx);
    const int top = imax(mlx, imax(mly, mlz));
    const size_t nfull = (size_t)p * m * n;
    /* a level whose rounded size is 0 (axis of length 1 with eta = 0.5, ...) makes the reference
     * raise ZeroDivisionError in its resampler (util/resize_util_3D.py:116-128): report bad input */
    for (int i = top; i >= min_level; i--) {
        int ls[3];
        level_size_of(p, m, n, eta, i, mlz, mly, mlx, ls);
        if (ls[0] < 1 || ls[1] < 1 || ls[2] < 1) return 1;
    }

    /* :343-350 */
    double *init[3];
    for (int d = 0; d < 3; d++) {
        init[d] = (double *)xmalloc(sizeof(double) * nfull);
        for (size_t i = 0; i < nfull; i++) init[d][i] = uvw ? uvw[i * 3 + d] : 0.0;
    }

    double *u = NULL, *v = NULL, *w = NULL; /* padded (lz+2,ly+2,lx+2) */
    int pz = 0, py = 0, px = 0;             /* previous level interior size */

    for (int i = top; i >= min_level; i--) {
        int ls[3];
        level_size_of(p, m, n, eta, i, mlz, mly, mlx, ls);
        const int lz = ls[0], ly = ls[1], lx = ls[2];
        const size_t nl = (size_t)lz * ly * lx;
        const int P = lz + 2, M = ly + 2, N = lx + 2;
        const size_t npad = (size_t)P * M * N;

        /* :409-416 */
        double *f1l = (double *)xmalloc(sizeof(double) * nl * C);
        double *f2l = (double *)xmalloc(sizeof(double) * nl * C);
        double *tmpc = (double *)xmalloc(sizeof(double) * nl);
        for (int c = 0; c < C; c++) {
            resize_chan_f64(fixed, p, m, n, C, c, lz, ly, lx, tmpc);
            for (size_t q = 0; q < nl; q++) f1l[q * C + c] = tmpc[q];
            resize_chan_f64(moving, p, m, n, C, c, lz, ly, lx, tmpc);
            for (size_t q = 0; q < nl; q++) f2l[q * C + c] = tmpc[q];
        }
        const double hz = (double)p / (double)lz, hy = (double)m / (double)ly,
                     hx = (double)n / (double)lx;

        double *un = (double *)xmalloc(sizeof(double) * npad);
        double *vn = (double *)xmalloc(sizeof(double) * npad);
        double *wn = (double *)xmalloc(sizeof(double) * npad);
        double *warped = (double *)xmalloc(sizeof(double) * nl * C);
        double *ui = (double *)xmalloc(sizeof(double) * nl);
        double *vi = (double *)xmalloc(sizeof(double) * nl);
        double *wi = (double *)xmalloc(sizeof(double) * nl);
        if (i == top) {
            /* :417-421 */
            resize_chan_f64(init[0], p, m, n, 1, 0, lz, ly, lx, ui);
            resize_chan_f64(init[1], p, m, n, 1, 0, lz, ly, lx, vi);
            resize_chan_f64(init[2], p, m, n, 1, 0, lz, ly, lx, wi);
            add_boundary(ui, lz, ly, lx, un);
            add_boundary(vi, lz, ly, lx, vn);
            add_boundary(wi, lz, ly, lx, wn);
            memcpy(warped, f2l, sizeof(double) * nl * C);
        } else {
            /* :424-434 */
            size_t nprev = (size_t)pz * py * px;
            double *prev = (double *)xmalloc(sizeof(double) * nprev);
            double *srcs[3] = {u, v, w};
            double *dsts[3] = {ui, vi, wi};
            for (int d = 0; d < 3; d++) {
                for (int a = 0; a < pz; a++)
                    for (int b = 0; b < py; b++)
                        for (int c = 0; c < px; c++)
                            prev[IDX3(a, b, c, py, px)] =
                                srcs[d][IDX3(a + 1, b + 1, c + 1, py + 2, px + 2)];
                resize_chan_f64(prev, pz, py, px, 1, 0, lz, ly, lx, dsts[d]);
            }
            free(prev);
            add_boundary(ui, lz, ly, lx, un);
            add_boundary(vi, lz, ly, lx, vn);
            add_boundary(wi, lz, ly, lx, wn);
            double *us = (double *)xmalloc(sizeof(double) * nl);
            double *vs = (double *)xmalloc(sizeof(double) * nl);
            double *ws = (double *)xmalloc(sizeof(double) * nl);
            for (size_t q = 0; q < nl; q++) {
                us[q] = ui[q] / hx;
                vs[q] = vi[q] / hy;
                ws[q] = wi[q] / hz;
            }
            float *wf = (float *)xmalloc(sizeof(float) * nl * C);
            fr3d_oracle_imregister(f2l, us, vs, ws, f1l, lz, ly, lx, C, 3, wf);
            for (size_t q = 0; q < nl * C; q++) warped[q] = (double)wf[q];
            free(wf); free(us); free(vs); free(ws);
        }
        free(u); free(v); free(w);
        u = un; v = vn; w = wn;
        free(ui); free(vi); free(wi);
        dbg_dump("warped", i, 0, warped, nl * C);
        dbg_dump("uinit", i, 0, u, npad);

        /* :440-473  motion tensor per channel into (P,M,N,C) */
        double *J[10], *Jc[10];
        for (int a = 0; a < 10; a++) {
            J[a] = (double *)xmalloc(sizeof(double) * npad * C);
            Jc[a] = (double *)xmalloc(sizeof(double) * npad);
        }
        double *c1 = (double *)xmalloc(sizeof(double) * nl);
        double *c2 = (double *)xmalloc(sizeof(double) * nl);
        for (int c = 0; c < C; c++) {
            for (size_t q = 0; q < nl; q++) {
                c1[q] = f1l[q * C + c];
                c2[q] = warped[q * C + c];
            }
            fr3d_oracle_motion_tensor_gc(c1, c2, lz, ly, lx, hz, hy, hx, Jc);
            for (int a = 0; a < 10; a++)
                for (size_t q = 0; q < npad; q++) J[a][q * C + c] = Jc[a][q];
        }
        free(c1); free(c2);
        for (int a = 0; a < 10; a++) free(Jc[a]);

        /* :475-483 */
        double *wl = (double *)calloc(npad * C, sizeof(double));
        if (!wl) abort();
        for (int c = 0; c < C; c++) {
            resize_chan_f64(weight, p, m, n, C, c, lz, ly, lx, tmpc);
            for (int a = 0; a < lz; a++)
                for (int b = 0; b < ly; b++)
                    for (int e = 0; e < lx; e++)
                        wl[IDX3(a + 1, b + 1, e + 1, M, N) * C + c] = tmpc[IDX3(a, b, e, ly, lx)];
        }

        /* :485-490 */
        double alpha_scaling = (i == min_level) ? 1.0 : pow(eta, -0.5 * (double)i);
        double at[3] = {alpha_scaling * alpha3[0], alpha_scaling * alpha3[1],
                        alpha_scaling * alpha3[2]};

        /* :492-516 */
        double *res = (double *)xmalloc(sizeof(double) * npad * 3);
        fr3d_oracle_compute_flow_3d((const double *const *)J, wl, u, v, w, P, M, N, C, at[0], at[1],
                                    at[2], iterations, update_lag, a_data, a_smooth, hx, hy, hz,
                                    res);
        dbg_dump("res", i, 0, res, npad * 3);
        /* :517-529 */
        double *dint = (double *)xmalloc(sizeof(double) * nl);
        double *dmed = (double *)xmalloc(sizeof(double) * nl);
        double *uvwp[3] = {u, v, w};
        int do_med = imin(lz, imin(ly, lx)) > 5;
        for (int d = 0; d < 3; d++) {
            if (do_med) {
                for (int a = 0; a < lz; a++)
                    for (int b = 0; b < ly; b++)
                        for (int e = 0; e < lx; e++)
                            dint[IDX3(a, b, e, ly, lx)] =
                                res[IDX3(a + 1, b + 1, e + 1, M, N) * 3 + d];
                fr3d_oracle_median5(dint, lz, ly, lx, dmed);
                for (int a = 0; a < lz; a++)
                    for (int b = 0; b < ly; b++)
                        for (int e = 0; e < lx; e++)
                            res[IDX3(a + 1, b + 1, e + 1, M, N) * 3 + d] =
                                dmed[IDX3(a, b, e, ly, lx)];
            }
            for (size_t q = 0; q < npad; q++) uvwp[d][q] = uvwp[d][q] + res[q * 3 + d];
            dbg_dump("u", i, d, uvwp[d], npad);
        }
        free(dint); free(dmed); free(res); free(wl);
        for (int a = 0; a < 10; a++) free(J[a]);
        free(f1l); free(f2l); free(tmpc); free(warped);
        pz = lz; py = ly; px = lx;
    }

    /* :530-541 */
    {
        size_t nl = (size_t)pz * py * px;
        double *comp = (double *)xmalloc(sizeof(double) * nl);
        double *full = (double *)xmalloc(sizeof(double) * nfull);
        double *uvwp[3] = {u, v, w};
        for (int d = 0; d < 3; d++) {
            for (int a = 0; a < pz; a++)
                for (int b = 0; b < py; b++)
                    for (int c = 0; c < px; c++)
                        comp[IDX3(a, b, c, py, px)] =
                            uvwp[d][IDX3(a + 1, b + 1, c + 1, py + 2, px + 2)];
            if (min_level > 0) {
                resize_chan_f64(comp, pz, py, px, 1, 0, p, m, n, full);
                for (size_t q = 0; q < nfull; q++) flow[q * 3 + d] = full[q];
            } else {
                for (size_t q = 0; q < nfull; q++) flow[q * 3 + d] = comp[q];
            }
        }
        free(comp); free(full);
    }
    free(u); free(v); free(w);
    for (int d = 0; d < 3; d++) free(init[d]);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* f-1  preprocessing: normalize + Gaussian filter      util/image_processing_3D.py:12-162      */
/*      (scipy.ndimage.gaussian_filter(mode="reflect", truncate=4.0), fp64)                     */
/* ------------------------------------------------------------------------------------------ */

/* numpy's float64 pairwise add.reduce for n <= 128 (same blocking as the float32 one above) */
static double np_sum_f64(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius); w has 2*radius+1 entries */
int fr3d_oracle_gaussian_kernel(double sigma, double truncate, double *w, int max_len)
{
    int radius = (int)(truncate * sigma + 0.5);
    int n = 2 * radius + 1;
    if (!w) return radius;
    if (n > max_len) return -1;
    double sigma2 = sigma * sigma;
    for (int i = 0; i < n; i++) {
        double x = (double)(i - radius);
        w[i] = exp(-0.5 / sigma2 * (x * x));
    }
    double s = n <= 128 ? np_sum_f64(w, n) : 0.0;
    if (n > 128) for (int i = 0; i < n; i++) s += w[i];
    for (int i = 0; i < n; i++) w[i] = w[i] / s;
    return radius;
}

/* 'reflect' (half-sample symmetric: d c b a | a b c d | d c b a), any distance */
static int reflect_hs(int i, int n)
{
    if (n == 1) return 0;
    int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

/* scipy ni_filters.c NI_Correlate1D, symmetric branch: centre tap first, then the pairs from the
 * outermost inwards.  data: contiguous (n0,n1,n2) fp64, filtered along `axis`, out != data. */
void fr3d_oracle_correlate1d_sym(const double *data, int n0, int n1, int n2, int axis, const double *w,
                                 int radius, double *out)
{
    const int dims[3] = {n0, n1, n2};
    const size_t strides[3] = {(size_t)n1 * n2, (size_t)n2, 1};
    const int n = dims[axis];
    const size_t st = strides[axis];
    const double *fw = w + radius;
    for (int a = 0; a < n0; a++)
        for (int b = 0; b < n1; b++)
            for (int c = 0; c < n2; c++) {
                const int idx[3] = {a, b, c};
                const int l = idx[axis];
                const double *line = data + (size_t)a * strides[0] + (size_t)b * strides[1] + c - (size_t)l * st;
                double tmp = line[(size_t)l * st] * fw[0];
                for (int jj = -radius; jj < 0; jj++)
                    tmp += (line[(size_t)reflect_hs(l + jj, n) * st] + line[(size_t)reflect_hs(l - jj, n) * st]) * fw[jj];
                out[(size_t)a * strides[0] + (size_t)b * strides[1] + c] = tmp;
            }
}

/* gaussian_filter(vol, sigma=(s0,s1,s2), mode="reflect", truncate) on (n0,n1,n2) fp64, in place;
 * axes with sigma <= 1e-15 are skipped (scipy _filters.py gaussian_filter). */
void fr3d_oracle_gaussian_filter3(double *vol, int n0, int n1, int n2, const double *sigma3, double truncate)
{
    size_t n = (size_t)n0 * n1 * n2;
    double *tmp = (double *)xmalloc(sizeof(double) * n);
    for (int axis = 0; axis < 3; axis++) {
        if (!(sigma3[axis] > 1e-15)) continue;
        int radius = fr3d_oracle_gaussian_kernel(sigma3[axis], truncate, NULL, 0);
        double *w = (double *)xmalloc(sizeof(double) * (size_t)(2 * radius + 1));
        fr3d_oracle_gaussian_kernel(sigma3[axis], truncate, w, 2 * radius + 1);
        fr3d_oracle_correlate1d_sym(vol, n0, n1, n2, axis, w, radius, tmp);
        memcpy(vol, tmp, sizeof(double) * n);
        free(w);
    }
    free(tmp);
}
