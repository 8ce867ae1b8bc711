/*
 * fr3d_oracle.h -- CPU restatement of flowreg3d's 3-D variational optical-flow path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / CPU baseline.  The product path (flowreg3d_amd/) never links,
 * imports or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/flowreg3d/).  Arithmetic types follow the reference run as plain
 * Python (numba's njit replaced by a no-op, NumPy 2.2 / SciPy 1.15 promotion rules), which is
 * how tests/golden/ was generated (tools/gen_golden.py): fp32 tables and fp32 accumulation in
 * the resampler, fp64 everywhere else, fp32 warp coordinates and warp output.
 *
 * Parity status: PINNED by tests/golden/ (npz files; outputs of the reference itself, stage by
 * stage and end to end); the reference's own test-suite holds no golden values for this path.
 */
#ifndef FR3D_ORACLE_H
#define FR3D_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* util/resize_util_3D.py:98-111 (+ :53-95).  idx/wt are (out_len, P) row-major, P = 2R+4.
 * Returns P; with idx == NULL only returns P. */
int fr3d_oracle_resize_tables(int in_len, int out_len, double sigma, int *idx, float *wt);

/* util/resize_util_3D.py:114-156 for one channel (D,H,W) fp32 -> (od,oh,ow) fp32. */
void fr3d_oracle_resize3d(const float *src, int D, int H, int W, int od, int oh, int ow,
                          double sigma_coeff, float *dst);
/* the same with per_axis (:120-123: one Gaussian sigma per axis from that axis' own scale) */
void fr3d_oracle_resize3d_ex(const float *src, int D, int H, int W, int od, int oh, int ow,
                             double sigma_coeff, int per_axis, float *dst);

/* scipy.ndimage.spline_filter(order=3, mode='nearest', output=float64) on a contiguous
 * (Z,Y,X) fp64 array, in place (axis 0, then 1, then 2). */
void fr3d_oracle_spline_filter3(double *c, int Z, int Y, int X);

/* core/optical_flow_3d.py:22-74.  f2,f1: (Z,Y,X,C) fp64; u,v,w: (Z,Y,X) fp64 in level voxels.
 * order: 3 = cubic, 1 = linear.  out: (Z,Y,X,C) fp32. */
void fr3d_oracle_imregister(const double *f2, const double *u, const double *v, const double *w,
                            const double *f1, int Z, int Y, int X, int C, int order, float *out);

/* The same with the output stored as the executor's final warp stores it when the raw volume has dtype
 * out_dtype (0 f32, 1 f64, 2 u8, 3 u16, 4 i16; SciPy allocates map_coordinates' output in the input's
 * dtype): parallelization/sequential_3d.py:153-170.  f2 holds the raw values exactly (fp64). */
void fr3d_oracle_imregister_typed(const double *f2, const double *u, const double *v, const double *w,
                                  const double *f1, int Z, int Y, int X, int C, int order, int out_dtype,
                                  void *out);

/* BatchMotionCorrector._update_reference (motion_correction/compensate_recording_3D.py:395-429):
 * batch_proc (T,Z,Y,X,C) fp64, flows (T,Z,Y,X,3) fp32, ref_proc (Z,Y,X,C) fp64 -> new_ref (Z,Y,X,C) fp64 =
 * per-channel mean of the last min(100,T) volumes warped by their flows (order 3 cubic / 1 linear). */
void fr3d_oracle_update_reference(const double *batch_proc, const float *flows, const double *ref_proc, int T,
                                  int Z, int Y, int X, int C, int order, double *new_ref);

/* core/optical_flow_3d.py:92-152.  f1,f2: (Z,Y,X) fp64.  J: 10 arrays (Z+2,Y+2,X+2) in the
 * reference's return order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34 (outer ring zero). */
void fr3d_oracle_motion_tensor_gc(const double *f1, const double *f2, int Z, int Y, int X,
                                  double hz, double hy, double hx, double *const J[10]);

/* core/level_solver_3d.py:314-546 (+ :246-311).  J*: (P,M,N,C); weight (P,M,N,C); u,v,w (P,M,N);
 * out (P,M,N,3).  Lexicographic SOR, omega = 1.95, fp64. */
void fr3d_oracle_compute_flow_3d(const double *const J[10], const double *weight,
                                 const double *u, const double *v, const double *w,
                                 int P, int M, int N, int C,
                                 double alpha_x, double alpha_y, double alpha_z,
                                 int iterations, int update_lag, const double *a_data,
                                 double a_smooth, double hx, double hy, double hz, double *out);

/* scipy.ndimage.median_filter(size=(5,5,5), mode='mirror') as used at
 * core/optical_flow_3d.py:517-526.  in/out (Z,Y,X) fp64, out != in. */
void fr3d_oracle_median5(const double *in, int Z, int Y, int X, double *out);

/* core/optical_flow_3d.py:77-85 */
int fr3d_oracle_warping_depth(double eta, int levels, int p, int m, int n);

/* Pyramid schedule of core/optical_flow_3d.py:389-408.  sizes: up to max_out rows of 3 ints
 * (z,y,x) from the coarsest to the finest solved level.  Returns the number of solves;
 * *min_level_eff receives the clamped min_level (:396-399). */
int fr3d_oracle_schedule(int p, int m, int n, double eta, int levels, int min_level,
                         int *sizes, int max_out, int *min_level_eff);

/* core/optical_flow_3d.py:319-542.  fixed,moving: (Z,Y,X,C) fp64; uvw: (Z,Y,X,3) fp64 or NULL;
 * weight: (Z,Y,X,C) fp64 (already expanded as :351-381 does) ; a_data: C values.
 * flow: (Z,Y,X,3) fp64.  Returns 0 on success. */
int fr3d_oracle_get_displacement(const double *fixed, const double *moving, int Z, int Y, int X,
                                 int C, const double *alpha3, int update_lag, int iterations,
                                 int min_level, int levels, double eta, double a_smooth,
                                 const double *a_data, const double *uvw, const double *weight,
                                 double *flow);

/* scipy.ndimage gaussian kernel / symmetric correlate1d / gaussian_filter(mode="reflect") as
 * util/image_processing_3D.py:95-162 uses them (fp64). */
int fr3d_oracle_gaussian_kernel(double sigma, double truncate, double *w, int max_len);
void fr3d_oracle_correlate1d_sym(const double *data, int n0, int n1, int n2, int axis, const double *w,
                                 int radius, double *out);
void fr3d_oracle_gaussian_filter3(double *vol, int n0, int n1, int n2, const double *sigma3, double truncate);

#ifdef __cplusplus
}
#endif
#endif
