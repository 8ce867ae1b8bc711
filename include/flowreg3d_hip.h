/*
 * flowreg3d_hip.h -- C ABI of the MI355X (gfx950) engine behind flowreg3d's
 * get_displacement / imregister_wrapper / executor.process_batch path.
 *
 * The reference is pure Python, so there is no existing FFI to mirror; each entry point names
 * the reference function (file:line under /root/reference/src/flowreg3d/) whose arithmetic it
 * replaces and is what a ctypes/cffi binding inside flowreg3d would call.
 *
 * Conventions
 *  - plain C types only; arrays are C-contiguous, channels-last exactly as NumPy hands them:
 *    volumes (Z,Y,X,C), flow (Z,Y,X,3) with components [dx,dy,dz] = [u,v,w].
 *  - "_dev" entry points take device pointers (hipMalloc'ed, e.g. torch tensors' data_ptr());
 *    the others take host pointers and stage through internal device buffers.
 *  - every function returns 0 on success, non-zero on failure; fr3d_last_error() gives the
 *    thread-local message.  Nothing is retained from caller pointers after a call returns.
 *  - all work is issued on one internal HIP stream per process and is complete on return
 *    unless stated otherwise.  Calls are serialised by an internal mutex (re-entrant safe).
 */
#ifndef FLOWREG3D_HIP_H
#define FLOWREG3D_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR3D_MAX_CHANNELS 8

/* element types of caller buffers */
#define FR3D_F32 0
#define FR3D_F64 1
#define FR3D_U8 2
#define FR3D_U16 3
#define FR3D_I16 4

/* Solver parameters = keyword arguments of get_displacement
 * (core/optical_flow_3d.py:319-333) as the executors pass them in `flow_params`
 * (motion_correction/compensate_recording_3D.py:301-315). */
typedef struct fr3d_params {
    double alpha[3];                   /* alpha_x, alpha_y, alpha_z */
    int update_lag;
    int iterations;
    int min_level;
    int levels;
    double eta;
    double a_smooth;                   /* 1.0: constant diffusion (fast path); otherwise psi_smooth is
                                          re-evaluated every iteration (k_sor_smooth.hip) */
    double a_data[FR3D_MAX_CHANNELS];  /* per channel */
    int solver_fp64;                   /* 0: fp32 storage + fp32 update arithmetic;
                                          1: fp32 storage, fp64 update arithmetic;
                                          2: fp64 storage and arithmetic in the solver (2x the
                                             bytes; for configurations where the reference's own
                                             iteration is ill-conditioned);
                                          3: packed 42-bit storage (the upper 42 bits of each fp64 value, three
                                             values per 16 bytes: 31 significant bits for 4/3 of the fp32 bytes),
                                             fp64 arithmetic; a_smooth == 1 sweep only (mode 2 otherwise);
                                          FR3D_SOLVER_AUTO (-1): the cheapest mode measured to stay within
                                             1e-4 voxels of the reference CPU path with margin -- one channel: 1 up
                                             to 2^22 voxels, 3 above; several channels: 2 (what the Python mirror
                                             passes by default) */
    int solver_sweep;                  /* which kernel runs the a_smooth == 1 sweep (results are bit-identical):
                                          FR3D_SWEEP_AUTO (0): the engine's choice (PLANES: it measured faster in
                                             every storage format; env FR3D_SWEEP overrides);
                                          FR3D_SWEEP_PLANES (1): one launch per hyperplane step, every iteration's
                                             operands streamed from HBM (k_sor.hip);
                                          FR3D_SWEEP_WINDOW (2): up to 5 iterations of a psi period per workgroup
                                             on chip, tiles of lines marching along x (k_sor_win.hip; 1 or 2
                                             channels -- 1 with fp64 storage --, PLANES otherwise; 12 more solver
                                             values per voxel of workspace) */
    int reserved[6];
} fr3d_params;
#define FR3D_SOLVER_AUTO (-1)
#define FR3D_SWEEP_AUTO 0
#define FR3D_SWEEP_PLANES 1
#define FR3D_SWEEP_WINDOW 2

/* The solver mode (0..3) the most recent flow solve of this process ran in: what FR3D_SOLVER_AUTO resolved to.  An
 * automatic choice of packed storage falls back to fp32 storage when one volume's packed solver slabs do not fit the
 * device's free memory and the fp32 slabs do.  -1 before the first solve. */
int fr3d_last_solver_mode(void);
/* 1 if that solve was an automatic choice that DEGRADED from packed 42-bit to fp32 storage for lack of device memory
 * (volumes of the 1024^3 class): the flow is then outside the 1e-4 parity bound the automatic modes are chosen for
 * (fp32 storage measures 1.5e-4 at 512^3 already; profiles/parity_fullsize.json).  The Python mirror warns. */
int fr3d_last_solver_fallback(void);

/* ---- lifetime ------------------------------------------------------------------------- */
int fr3d_init(int device);              /* hipSetDevice + stream + workspace; idempotent */
void fr3d_shutdown(void);               /* frees every device buffer and the stream */
const char *fr3d_last_error(void);
int fr3d_device_count(void);            /* hipGetDeviceCount without initialising a context */
const char *fr3d_version(void);
/* One-line description of the initialised device ("name; N CUs; core MHz; memory MHz, bus bits;
 * GiB") for logs and bench output; "" before fr3d_init. */
const char *fr3d_device_info(void);
/* Volumes of a batch that fr3d_process_batch solves in lock step (their SOR launches are shared; env FR3D_BATCH).
 * Default: 8 volumes of up to 2^24 voxels, and the same number of VOXELS for smaller volumes (at most 128 volumes:
 * small volumes are bound by their launch count, 48^3 runs 1.8x faster with 128 in lock step than with 8).  Also reserves
 * solver workspace for that many volumes on first use, so a later full batch does not reallocate.  0 restores the
 * default. */
int fr3d_set_batch(int nvol);
/* Engine lanes of fr3d_process_batch* (1 or 2; default 2, env FR3D_LANES; returns the previous value).  With 2 the
 * lock-step batches of a series are dealt alternately to two engine lanes -- two HIP streams, each with its own workspace
 * (batches of half the size: the same memory in total), lane 1 fed by a host thread of its own -- so that the stages of
 * one batch that do not fill the memory system (median, warps, motion tensor, the short launches of the coarse levels,
 * the ramp and tail of every sweep launch) run under the other batch's sweep: +9..12 % at 256^3.  Results are
 * bit-identical to one lane.  The progress callback always runs on the CALLER's thread (lane 1 queues its events, the
 * calling thread delivers them): a callback may re-enter the library and may be thread-affine, as with the reference.  The
 * reference's counterpart is its executor's worker pool (parallelization/multiprocessing_3d.py:286-318: several
 * batches in flight).  While profiling brackets are on (fr3d_prof_enable(1)) a call runs on ONE lane: the event spans
 * of two lanes overlap and would not be kernel times. */
int fr3d_set_lanes(int lanes);

/* ---- the hot path --------------------------------------------------------------------- */

/* get_displacement (core/optical_flow_3d.py:319-542): multiscale coarse-to-fine solve.
 * fixed/moving: (Z,Y,X,C) fp32; uvw_init: (Z,Y,X,3) fp32 or NULL (= zeros);
 * weight: (Z,Y,X,C) fp32 or NULL (= 1/C); flow_out: (Z,Y,X,3) fp32. */
int fr3d_get_displacement(const fr3d_params *p, const float *fixed, const float *moving,
                          int Z, int Y, int X, int C, const float *uvw_init,
                          const float *weight, float *flow_out);
int fr3d_get_displacement_dev(const fr3d_params *p, const float *fixed, const float *moving,
                              int Z, int Y, int X, int C, const float *uvw_init,
                              const float *weight, float *flow_out);

/* VERIFICATION mode of fr3d_get_displacement (host pointers, one volume, a_smooth == 1): the same pyramid on the same
 * data path -- compact skewed solver layout, hyperplane launch schedule, resampler / prefilter / gather / tensor /
 * median stages -- with the solver evaluated exactly as core/level_solver_3d.py:356-377,472-540 writes it (fp64, expanded
 * quadratic form, per-channel accumulation order, true divisions, no FMA contraction) and the level flow kept in fp64
 * like the reference's `u = u + du`.  psi goes through the portable pow of flowreg3d_amd/csrc/portable_pow.h, the one the
 * `ppow` build of the CPU oracle uses: against that build the result is BIT-IDENTICAL, at any size
 * (tests/test_gpu_verify_mode.py).  About 5x slower than the fp64-storage mode; not a production mode.
 * flow_out: (Z,Y,X,3) float64. */
int fr3d_get_displacement_verify(const fr3d_params *p, const float *fixed, const float *moving, int Z, int Y, int X,
                                 int C, const float *uvw_init, const float *weight, double *flow_out);

/* Test hook: the cubic B-spline coefficients of a float32 volume as the gather reads them (pad-free prefilter,
 * coefficients -2 .. N+1 per axis): coef_out (Z+4,Y+4,X+4) float64 = SciPy's spline_filter of the volume padded by 12
 * replicated voxels, cropped to that range.  Every axis at least 41 voxels. */
int fr3d_spline_coefficients(const float *vol, int Z, int Y, int X, double *coef_out);

/* Test hook: the gradient-constancy motion tensor in float64 as the verification mode forms it (no rounding to the
 * solver's storage): J_out (10,Z,Y,X) float64, order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34. */
int fr3d_motion_tensor_f64(const float *f1, const float *f2, int Z, int Y, int X, double hz, double hy, double hx,
                           double *J_out);

/* Test hook: the verification mode's sweep alone (k_verify.hip) -- level_solver in the reference's arithmetic.
 * J: (C,10,Z,Y,X) float64 tensor entries in the order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34 (interior, no ghost
 * ring); weight (C,Z,Y,X) and uvw (3,Z,Y,X) float32; duvw_out (3,Z,Y,X) float64.  a_smooth == 1. */
int fr3d_level_solve_verify(const double *J, const float *weight, const float *uvw, int Z, int Y, int X, int C,
                            const double *alpha3, int iterations, int update_lag, const double *a_data, double hx,
                            double hy, double hz, double *duvw_out);

/* level_solver for ANY motion tensor (core/optical_flow_3d.py:262-316 takes whatever its caller built): the sweep in
 * the reference's own arithmetic on tensor entries -- fp64, the expanded quadratic form of psi_data, psi_smooth
 * re-evaluated every iteration when a_smooth != 1 (core/level_solver_3d.py:262-311,340-546) -- where
 * fr3d_level_solve needs the rank-3 square-root factors of the gradient-constancy tensor.  Slower (no frozen system,
 * one sweep per launch chain when a_smooth != 1); the Python mirror's level_solver takes it for tensors that are
 * not rank 3, and for u, v, w whose ghost ring is not the edge pad of the interior.  J: (C,10,Z,Y,X) float64, order
 * J11,J22,J33,J44,J12,J13,J23,J14,J24,J34 (interior); weight (C,Z,Y,X) fp32; uvw (3,Z,Y,X) float64 interior flow;
 * uvw_padded: NULL (ghosts = edge pad of the interior, what get_displacement passes) or the caller's padded arrays
 * (3,Z+2,Y+2,X+2) float64, whose ring then enters the surface voxels' stencil and psi_smooth exactly as in the
 * reference; duvw_out (3,Z,Y,X) float64. */
int fr3d_level_solve_tensor(const double *J, const float *weight, const double *uvw, const double *uvw_padded, int Z,
                            int Y, int X, int C, const double *alpha3, int iterations, int update_lag,
                            const double *a_data, double a_smooth, double hx, double hy, double hz, double *duvw_out);

/* Test hook: the verification mode's portable pow (flowreg3d_amd/csrc/portable_pow.h) evaluated on the device for n
 * host values -- tests compare it bit for bit with the same source compiled for the host. */
int fr3d_portable_pow(const double *x, const double *y, size_t n, double *out);

/* imregister_wrapper (core/optical_flow_3d.py:22-74): backward warp of `vol` by `flow`,
 * out-of-bounds voxels taken from `ref`.  order 3 = cubic B-spline with SciPy's prefilter,
 * order 1 = linear.  vol/ref (Z,Y,X,C) of vol_dtype; flow (Z,Y,X,3) of flow_dtype;
 * out (Z,Y,X,C) fp32. */
int fr3d_warp(const void *vol, int vol_dtype, const void *flow, int flow_dtype, const void *ref,
              int Z, int Y, int X, int C, int order, float *out);
int fr3d_warp_dev(const void *vol, int vol_dtype, const void *flow, int flow_dtype,
                  const void *ref, int Z, int Y, int X, int C, int order, float *out);

/* Per-volume body of the executors (motion_correction/parallelization/sequential_3d.py:148-175)
 * for T volumes against one fixed reference: flow = get_displacement(ref_proc, batch_proc[t],
 * uvw=w_init) ; registered = imregister(batch_raw[t], flow, ref_raw, order).
 * batch_proc/batch_raw: (T,Z,Y,X,C) fp32; ref_proc/ref_raw: (Z,Y,X,C) fp32;
 * flows_out: (T,Z,Y,X,3) fp32; registered_out: (T,Z,Y,X,C) fp32.
 * The fixed-reference pyramid is built once per call.  `progress` (nullable) is called with 1
 * after each volume (base_3d.py:46 progress_callback), on the calling thread, also when two engine lanes run. */
typedef void (*fr3d_progress_fn)(int volumes_done, void *user);
int fr3d_process_batch(const fr3d_params *p, const float *batch_proc, const float *batch_raw,
                       const float *ref_proc, const float *ref_raw, const float *w_init,
                       const float *weight, int T, int Z, int Y, int X, int C, int order,
                       float *flows_out, float *registered_out, fr3d_progress_fn progress,
                       void *user);
int fr3d_process_batch_dev(const fr3d_params *p, const float *batch_proc, const float *batch_raw,
                           const float *ref_proc, const float *ref_raw, const float *w_init,
                           const float *weight, int T, int Z, int Y, int X, int C, int order,
                           float *flows_out, float *registered_out, fr3d_progress_fn progress,
                           void *user);

/* The same with the raw series in its own element type, as the executors hand it over
 * (parallelization/sequential_3d.py:153-170 warps `batch[t]` itself): batch_raw and registered_out are
 * (T,Z,Y,X,C) of raw_dtype (FR3D_F32|F64|U8|U16|I16), ref_raw is (Z,Y,X,C) of ref_dtype (F32|F64; the
 * reference pipeline keeps it in float64).  The spline coefficients are computed from the raw values in
 * fp64 and the result is stored the way SciPy + the reference store it for that dtype: integer types
 * round half up / half away from zero and saturate (map_coordinates allocates its output in the input's
 * dtype), float64 holds the float32-rounded value, out-of-bounds voxels take ref_raw through float32. */
int fr3d_process_batch_raw(const fr3d_params *p, const float *batch_proc, const void *batch_raw,
                           int raw_dtype, const float *ref_proc, const void *ref_raw, int ref_dtype,
                           const float *w_init, const float *weight, int T, int Z, int Y, int X, int C,
                           int order, float *flows_out, void *registered_out,
                           fr3d_progress_fn progress, void *user);
int fr3d_process_batch_raw_dev(const fr3d_params *p, const float *batch_proc, const void *batch_raw,
                               int raw_dtype, const float *ref_proc, const void *ref_raw, int ref_dtype,
                               const float *w_init, const float *weight, int T, int Z, int Y, int X,
                               int C, int order, float *flows_out, void *registered_out,
                               fr3d_progress_fn progress, void *user);

/* Preprocessing in front of the flow path (SURVEY section 8 f-1;
 * motion_correction/compensate_recording_3D.py:229-254): per channel c
 *   out = gaussian_filter((frames - norm_min[c]) / norm_den[c], sigma, mode="reflect", truncate)
 * (util/image_processing_3D.py:12-162).  frames: (T,Z,Y,X,C) of `dtype`; sigma: (C,4) =
 * [sx,sy,sz,st] per channel (st filters across the T volumes of the batch, as the reference's 4-D
 * filter does; axes with sigma <= 1e-15 are skipped); out: (T,Z,Y,X,C) of out_dtype (F32|F64).
 * norm_min/norm_den are the caller's normalisation constants (min and max-min(+eps) of the
 * reference volume for "together", per channel for "separate"; 0 and 1 for no normalisation). */
int fr3d_preprocess(const void *frames, int dtype, int T, int Z, int Y, int X, int C,
                    const double *norm_min, const double *norm_den, const double *sigma,
                    double truncate, void *out, int out_dtype);
int fr3d_preprocess_dev(const void *frames, int dtype, int T, int Z, int Y, int X, int C,
                        const double *norm_min, const double *norm_den, const double *sigma,
                        double truncate, void *out, int out_dtype);

/* The same passes with any of scipy.ndimage's boundary modes (util/image_processing_3D.py:95-162 hands `mode` through to
 * scipy.ndimage.gaussian_filter; the pipeline itself only uses "reflect"): what lies beyond an edge of a b c d is
 *   FR3D_BOUNDARY_REFLECT  d c b a | a b c d | d c b a   ("reflect", "grid-mirror")
 *   FR3D_BOUNDARY_CONSTANT 0 0 0 0 | a b c d | 0 0 0 0   ("constant" / "grid-constant" with cval = 0, of the
 *                                                          normalised array)
 *   FR3D_BOUNDARY_NEAREST  a a a a | a b c d | d d d d   ("nearest")
 *   FR3D_BOUNDARY_MIRROR   d c b   | a b c d | c b a     ("mirror")
 *   FR3D_BOUNDARY_WRAP     a b c d | a b c d | a b c d   ("wrap", "grid-wrap")
 * fr3d_preprocess* is this entry with FR3D_BOUNDARY_REFLECT (the specialised kernels; the other modes take the
 * general pass). */
#define FR3D_BOUNDARY_REFLECT 0
#define FR3D_BOUNDARY_CONSTANT 1
#define FR3D_BOUNDARY_NEAREST 2
#define FR3D_BOUNDARY_MIRROR 3
#define FR3D_BOUNDARY_WRAP 4
int fr3d_gaussian_filter(const void *frames, int dtype, int T, int Z, int Y, int X, int C,
                         const double *norm_min, const double *norm_den, const double *sigma,
                         double truncate, int mode, void *out, int out_dtype);
int fr3d_gaussian_filter_dev(const void *frames, int dtype, int T, int Z, int Y, int X, int C,
                             const double *norm_min, const double *norm_den, const double *sigma,
                             double truncate, int mode, void *out, int out_dtype);

/* Reference update of the batch driver (BatchMotionCorrector._update_reference,
 * motion_correction/compensate_recording_3D.py:395-429): per channel, the last min(100,T) volumes of
 * batch_proc are warped by their flows (imregister_wrapper, fp32 result; out-of-bounds voxels from the
 * current reference_proc) and averaged in fp64 in stack order.  batch_proc: (T,Z,Y,X,C) of proc_dtype
 * (FR3D_F32|F64); flows: (T,Z,Y,X,3) fp32; ref_proc: (Z,Y,X,C) of ref_dtype; new_ref: (Z,Y,X,C) fp64 (left
 * untouched when T == 0, like the reference).  new_ref must not alias ref_proc. */
int fr3d_update_reference(const void *batch_proc, int proc_dtype, const float *flows, const void *ref_proc,
                          int ref_dtype, int T, int Z, int Y, int X, int C, int order, double *new_ref);
int fr3d_update_reference_dev(const void *batch_proc, int proc_dtype, const float *flows,
                              const void *ref_proc, int ref_dtype, int T, int Z, int Y, int X, int C,
                              int order, double *new_ref);

/* Per-volume displacement statistics of the batch driver (SURVEY section 8 f-2;
 * motion_correction/compensate_recording_3D.py:488-508).  flows: (T,Z,Y,X,3) fp32;
 * out (host): T x 6 doubles = mean|w|, max|w|, mean divergence (np.gradient, unit spacing),
 * mean u, mean v, mean w. */
int fr3d_flow_stats(const float *flows, int T, int Z, int Y, int X, double *out);
int fr3d_flow_stats_dev(const float *flows, int T, int Z, int Y, int X, double *out);

/* np.mean(stack, axis=0) of `count` float32 arrays of n elements, device pointers: float32 accumulation in
 * stack order, divided by float32(count) -- the batch driver's w_init updates
 * (compensate_recording_3D.py:342-393, 481-485) without taking the flows off the device. */
int fr3d_mean_stack_dev(const float *stack, int count, size_t n, float *out);

/* ---- kernel-level entry points (stage parity tests; host pointers) ---------------------- */

/* imresize_fused_gauss_cubic3D (util/resize_util_3D.py:114-156), one fp32 channel. */
int fr3d_resize3d(const float *src, int D, int H, int W, int od, int oh, int ow, float *dst);
/* the same with the function's other arguments: sigma_coeff (0.6 on the flow path) and per_axis (each axis'
 * Gaussian sigma from its own scale factor, :120-123) */
int fr3d_resize3d_ex(const float *src, int D, int H, int W, int od, int oh, int ow, double sigma_coeff,
                     int per_axis, float *dst);

/* get_motion_tensor_gc (core/optical_flow_3d.py:92-152) for one channel.  f1,f2 (Z,Y,X) fp32;
 * J: 10 x (Z,Y,X) fp32 interior values, order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34.
 * A (nullable): 12 x (Z,Y,X) fp32 square-root factors a_k = sqrt(reg_k)*(f_kx,f_ky,f_kz,f_kt),
 * k = x,y,z (index 4k+column), J = sum_k a_k a_k^T -- the form the solver evaluates psi from. */
int fr3d_motion_tensor(const float *f1, const float *f2, int Z, int Y, int X, double hz,
                       double hy, double hx, float *J, float *A);

/* level_solver -> compute_flow_3d (core/level_solver_3d.py:314-546).
 * A: (12,C,Z,Y,X) fp32 square-root factors of the motion tensor as fr3d_motion_tensor returns
 * them (J = sum_k a_k a_k^T; the solver rebuilds the tensor entries from them);
 * weight: (C,Z,Y,X) fp32; uvw: (3,Z,Y,X) fp32 interior flow (ghosts are the edge pad of
 * optical_flow_3d.py:88); duvw_out: (3,Z,Y,X) fp32 interior increments. */
int fr3d_level_solve(const float *A, const float *weight, const float *uvw, int Z, int Y, int X,
                     int C, const double *alpha3, int iterations, int update_lag,
                     const double *a_data, double a_smooth, double hx, double hy, double hz,
                     int solver_fp64, float *duvw_out);

/* scipy.ndimage.median_filter(size=5^3, mode="mirror") (core/optical_flow_3d.py:517-526). */
int fr3d_median5(const float *in, int Z, int Y, int X, float *out);

/* Level sizes of the pyramid (core/optical_flow_3d.py:77-85,389-408), coarse -> fine.
 * sizes: max_out x 3 ints (z,y,x).  Returns the number of solves (>= 1) or -1. */
int fr3d_schedule(int Z, int Y, int X, double eta, int levels, int min_level, int *sizes,
                  int max_out, int *min_level_eff);

/* Host-only self-check of the SOR launch schedule (k_sor.hip; no GPU needed): replays the kernel's index
 * arithmetic for a level of Z x Y x X voxels and `iterations` sweeps with tiles of 64 lanes x `tile_rows` rows x
 * `chain` consecutive iterations per workgroup (0, 0 = the shape the engine uses).  Every voxel update of
 * core/level_solver_3d.py:383-540 must be issued exactly once, by launch i + j + k + 2 t.
 * Returns the number of violations (0 = consistent), -1 on bad arguments; *n_updates = updates issued. */
long long fr3d_sor_schedule_check(int Z, int Y, int X, int iterations, int tile_rows, int chain, long long *n_updates);

/* ---- device memory helpers (so a host program needs no other GPU runtime) -------------- */
void *fr3d_dev_malloc(size_t bytes);
void fr3d_dev_free(void *p);
int fr3d_h2d(void *dst_dev, const void *src_host, size_t bytes);
int fr3d_d2h(void *dst_host, const void *src_dev, size_t bytes);
int fr3d_sync(void);

/* ---- measurement ----------------------------------------------------------------------- */

/* HIP-event timing of the kernels on the engine's own stream, accumulated since the last
 * reset.  Kernel ids: */
#define FR3D_K_SOR 0      /* SOR hyperplane sweep (K4+K6+K7 fused) */
#define FR3D_K_WARP 1     /* B-spline gather */
#define FR3D_K_PREFILTER 2
#define FR3D_K_TENSOR 3
#define FR3D_K_RESIZE 4
#define FR3D_K_MEDIAN 5
#define FR3D_K_OTHER 6
#define FR3D_K_PREPROC 7
#define FR3D_K_COUNT 8
typedef struct fr3d_kernel_stat {
    double ms;              /* summed event-to-event time */
    double algo_bytes;      /* algorithmic bytes moved (per-kernel definition) */
    long long launches;     /* kernel launches inside the timed brackets */
    long long units;        /* voxel updates / output voxels processed */
} fr3d_kernel_stat;
int fr3d_prof_enable(int on);   /* off by default: brackets cost two events per stage */
int fr3d_prof_reset(void);
int fr3d_prof_get(fr3d_kernel_stat *out /* FR3D_K_COUNT entries */);
/* Streaming rate this device sustains right now: `reps` launches of y += x over two arrays of
 * n_floats (12 B per element: two reads, one write), HIP-event timed on the engine stream.  The
 * practical ceiling next to the nominal 8 TB/s when reading roofline fractions. */
int fr3d_stream_probe(size_t n_floats, int reps, double *gbytes_per_s);
/* Same for a read-only stream (eight arrays of n_floats summed per thread, 32 B per element): the
 * ceiling for a read-heavy kernel such as the SOR sweep. */
int fr3d_read_probe(size_t n_floats, int reps, double *gbytes_per_s);
/* Which XCD (0..7, HW_REG_XCC_ID) each workgroup of a grid_x x grid_y launch of 128 threads ran on,
 * xcc_of_block[y * grid_x + x]: the placement the sweep's XCD-aware tile order assumes (workgroup ids with equal
 * blockIdx.x % 8 share an XCD) is an observation, not a HIP guarantee -- this lets a test and a user look. */
int fr3d_xcd_probe(int grid_x, int grid_y, int *xcc_of_block);

#ifdef __cplusplus
}
#endif
#endif
