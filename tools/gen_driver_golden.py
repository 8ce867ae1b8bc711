#!/usr/bin/env python3
"""Fixtures for the batch driver from the reference itself (build container only).

Runs the reference's own ``compensate_arr_3D`` (motion_correction/compensate_arr_3D.py:13-143, i.e.
``BatchMotionCorrector.run``, compensate_recording_3D.py:431-555) on small synthetic series and records

    (video, reference, options)  ->  (registered, w, mean_disp, max_disp, mean_div, mean_translation, final w_init)

as tests/golden/drv_*.npz (arrays + one JSON blob of the options; no source, no bytecode).

How the reference is made importable here: ``motion_correction/OF_options_3D.py:20`` imports ``tifffile`` and the IO
package imports ``h5py`` / ``hdf5storage`` at module level; none of them is installed and none is touched by the
in-memory array path, so EMPTY modules of those names are put into ``sys.modules`` next to the no-op ``numba.njit``
stand-in of tools/gen_golden.py (the reference's source then runs as plain Python under NumPy 2.2 / SciPy 1.15).
The sequential executor is selected by narrowing ``RuntimeContext``'s ``available_parallelization`` set to
{"sequential3d"} for the duration of the call (``_setup_executor`` auto-selects from it, compensate_recording_3D.py:79-93);
the multiprocessing executor would fork workers that time out under pure-Python kernels.

Cases (branches of the driver they pin):
  drv_t3_serial   T=3, one batch: n_init <= 4 -> serial ``_compute_flow_single`` bootstrap (:379-385); 4-D input with a
                  3-D reference (squeeze path of compensate_arr_3D.py:55-70,127-135)
  drv_t7_b5       T=7, buffer_size=5: n_init = 5 > 4 -> executor bootstrap (:365-378), w_init rolled into batch 2 (:481-485)
  drv_t7_b3       T=7, buffer_size=3: three batches, mean of each batch's flows carried forward
  drv_noinit      update_initialization_w=False (:470-473)
  drv_c2_u16      two channels with 1-D weights, uint16 series, output_typename="uint16" (compensate_arr_3D.py:112-125)
  drv_consistency the inputs of the reference's OWN cross-executor test (tests/motion_correction/test_parallelization.py:
                  152-198: seed 42, (8,6,12,12,2) uniform noise, reference = mean of the first two volumes,
                  OFOptions(quality_setting="fast", levels=2, iterations=5), everything else default); that test holds
                  its executors to rtol 1e-5 / atol 1e-6 of each other on `registered`

Usage:  python tools/gen_driver_golden.py [--only NAME]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_SRC = "/root/reference/src"


def series(T, shape, C, seed, amp=1.0, dtype=np.float32):
    """reference volume + T moving volumes: smooth texture, per-timepoint smooth displacement of ~amp voxels."""
    from gen_golden import moved, smooth_volume
    fixed = np.stack([smooth_volume(shape, seed + c) for c in range(C)], -1)
    vols = []
    for t in range(T):
        s = amp * (0.4 + 0.25 * t)
        shift = (0.9 * s, -0.6 * s, 0.35 * s)
        vols.append(np.stack([moved(fixed[..., c], shift, seed) for c in range(C)], -1))
    video = np.stack(vols)
    if np.issubdtype(dtype, np.integer):
        return (video * 3000 + 200).astype(dtype), (fixed * 3000 + 200).astype(np.float64)
    return (video * 100 + 10).astype(dtype), (fixed * 100 + 10).astype(np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    from gen_golden import _install_numba_stub
    _install_numba_stub()
    for name in ("tifffile", "h5py", "hdf5storage"):  # imported at module level by the IO layer, unused by arrays
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    import scipy
    from flowreg3d._runtime import RuntimeContext
    from flowreg3d.motion_correction import compensate_arr_3D as mod
    from flowreg3d.motion_correction import compensate_recording_3D as rec
    from flowreg3d.motion_correction.OF_options_3D import OFOptions

    # keep the corrector of the call so that its statistics and final w_init can be recorded
    made = []

    class Recording(rec.BatchMotionCorrector):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)

    mod.BatchMotionCorrector = Recording
    RuntimeContext._config["available_parallelization"] = {"sequential3d"}

    base = dict(alpha=(0.25, 0.25, 0.25), levels=3, min_level=0, eta=0.8, update_lag=3, iterations=9, a_smooth=1.0,
                a_data=0.45, verbose=True)
    cases = {
        "drv_t3_serial": dict(T=3, shape=(10, 14, 16), C=1, squeeze=True, dtype=np.float32,
                              opts=dict(base, weight=[1.0], sigma=[[1.0, 0.8, 0.6, 0.3]], buffer_size=10)),
        "drv_t7_b5": dict(T=7, shape=(9, 12, 14), C=1, squeeze=False, dtype=np.float32,
                          opts=dict(base, weight=[1.0], sigma=[[1.0, 1.0, 1.0, 0.1]], buffer_size=5)),
        "drv_t7_b3": dict(T=7, shape=(9, 12, 14), C=1, squeeze=False, dtype=np.float64,
                          opts=dict(base, weight=[1.0], sigma=[[0.8, 0.8, 0.8, 0.4]], buffer_size=3,
                                    interpolation_method="linear")),
        "drv_noinit": dict(T=4, shape=(8, 12, 12), C=1, squeeze=False, dtype=np.float32,
                           opts=dict(base, weight=[1.0], sigma=[[1.0, 1.0, 1.0, 0.1]], buffer_size=2,
                                     update_initialization_w=False)),
        "drv_c2_u16": dict(T=3, shape=(8, 12, 14), C=2, squeeze=False, dtype=np.uint16,
                           opts=dict(base, weight=[0.7, 0.3], sigma=[[1.0, 1.0, 1.0, 0.1], [0.8, 0.8, 0.8, 0.2]],
                                     buffer_size=2, output_typename="uint16", channel_normalization="separate")),
    }
    cases["drv_consistency"] = dict(own_test=True, opts=dict(quality_setting="fast", levels=2, iterations=5))
    for name, cs in cases.items():
        if args.only and args.only != name:
            continue
        if cs.get("own_test"):
            np.random.seed(42)  # as the reference's test does
            video_in = np.random.rand(8, 6, 12, 12, 2).astype(np.float32)
            ref_in = np.mean(video_in[:2], axis=0)
        else:
            video, ref = series(cs["T"], cs["shape"], cs["C"], seed=7 * len(name), dtype=cs["dtype"])
            if cs["squeeze"]:
                video_in, ref_in = video[..., 0], ref[..., 0]
            else:
                video_in, ref_in = video, ref
        opt = OFOptions(**cs["opts"])
        if cs.get("own_test"):
            # record what the defaults resolved to, so that the product's Options can be built from the fixture alone
            cs = dict(cs, opts=dict(alpha=[float(a) for a in np.atleast_1d(opt.alpha)], weight=[float(x) for x in np.atleast_1d(opt.weight)],
                                    levels=int(opt.levels), min_level=int(opt.min_level), quality_setting=str(getattr(opt.quality_setting, "value", opt.quality_setting)),
                                    eta=float(opt.eta), update_lag=int(opt.update_lag), iterations=int(opt.iterations),
                                    a_smooth=float(opt.a_smooth), a_data=float(opt.a_data), sigma=np.asarray(opt.sigma).tolist(),
                                    buffer_size=int(opt.buffer_size),
                                    output_typename=getattr(opt, "output_typename", "double"),
                                    channel_normalization=str(getattr(opt.channel_normalization, "value", opt.channel_normalization)),
                                    interpolation_method=str(getattr(opt.interpolation_method, "value", opt.interpolation_method)),
                                    update_initialization_w=bool(opt.update_initialization_w)))
        made.clear()
        seen = []
        t0 = time.time()
        reg, w = mod.compensate_arr_3D(video_in.copy(), ref_in.copy(), opt, progress_callback=lambda a, b: seen.append((a, b)))
        dt = time.time() - t0
        bc = made[-1]
        meta = dict(case=name, options={k: (list(v) if isinstance(v, tuple) else v) for k, v in cs["opts"].items()},
                    numpy=np.__version__, scipy=scipy.__version__, executor=type(bc.executor).__name__,
                    effective_min_level=int(opt.effective_min_level), seconds=dt,
                    generator="tools/gen_driver_golden.py: flowreg3d.motion_correction.compensate_arr_3D (reference source, "
                              "no-op numba.njit, empty tifffile/h5py/hdf5storage modules, sequential executor)")
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, video=video_in, reference=ref_in, registered=reg, w=w,
                            mean_disp=np.asarray(bc.mean_disp), max_disp=np.asarray(bc.max_disp),
                            mean_div=np.asarray(bc.mean_div), mean_translation=np.asarray(bc.mean_translation),
                            w_init_final=np.asarray(bc.w_init), progress=np.asarray(seen, dtype=np.int64).reshape(-1, 2),
                            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print(f"{name}: {dt:.1f} s, executor {meta['executor']}, registered {reg.dtype}{reg.shape}, w {w.dtype}{w.shape}, "
              f"mean |w| {np.round(bc.mean_disp, 3).tolist()}, {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
