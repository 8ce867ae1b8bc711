#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference itself (run in the build container only).

The reference (/root/reference/src/flowreg3d) is pure Python + Numba.  Numba is not installed
here, so a no-op ``numba.njit`` stand-in is put in ``sys.modules`` and the reference's own source
runs as plain Python under NumPy 2.2 / SciPy 1.15 (strict IEEE order; numba's fastmath
reassociation is not reproduced).  Each fixture stores inputs and the reference's outputs for one
stage of the hot path or for the whole get_displacement call.
Only data is written -- no reference source or bytecode is copied.

Usage:  python tools/gen_golden.py [--only NAME]
"""
import argparse
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_SRC = "/root/reference/src"


def _install_numba_stub():
    m = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    m.njit = njit
    sys.modules["numba"] = m


def smooth_volume(shape, seed, sigma=1.5):
    from scipy.ndimage import gaussian_filter
    rng = np.random.Generator(np.random.PCG64(seed))
    a = gaussian_filter(rng.random(shape, dtype=np.float32).astype(np.float64), sigma, mode="reflect")
    a = (a - a.min()) / (a.max() - a.min())
    return a.astype(np.float32)


def moved(fixed, shift, seed):
    """moving(x) = fixed(x - d(x)) with a smooth d around `shift` (dx,dy,dz)."""
    from scipy.ndimage import map_coordinates
    Z, Y, X = fixed.shape
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    dx = shift[0] + 0.3 * np.sin(2 * np.pi * yy / Y)
    dy = shift[1] + 0.3 * np.cos(2 * np.pi * xx / X)
    dz = shift[2] + 0.2 * np.sin(2 * np.pi * (xx + yy) / (X + Y))
    out = map_coordinates(fixed.astype(np.float64), [zz - dz, yy - dy, xx - dx], order=3, mode="nearest")
    return out.astype(np.float32)


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    _install_numba_stub()
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    from flowreg3d.core import optical_flow_3d as of
    from flowreg3d.core import level_solver_3d as ls
    from flowreg3d.util import resize_util_3D as rz
    import scipy
    from scipy.ndimage import median_filter, spline_filter, map_coordinates

    meta = dict(numpy=np.__version__, scipy=scipy.__version__)
    print("reference imported; numpy", np.__version__, "scipy", scipy.__version__)

    def want(n):
        return args.only is None or args.only == n

    # ---- K1: tables and 3-D resampling -----------------------------------------------------
    if want("k1_resize"):
        out = {}
        cases = [(64, 51, 0.6 / (51 / 64)), (64, 20, 0.6 / (20 / 64)), (41, 51, 0.0),
                 (20, 64, 0.0), (7, 5, 0.84), (512, 168, 0.6 / (168 / 512))]
        out["table_cases"] = np.array(cases, dtype=np.float64)
        for k, (a, b, s) in enumerate(cases):
            idx, wt = rz._precompute_fused_gauss_cubic(int(a), int(b), float(s))
            out[f"idx{k}"] = idx
            out[f"wt{k}"] = wt
        vol = smooth_volume((18, 22, 26), 7)
        out["vol"] = vol
        out["down"] = rz.imresize_fused_gauss_cubic3D(vol, (11, 15, 17))
        out["up"] = rz.imresize_fused_gauss_cubic3D(vol, (23, 28, 33))
        out["mixed"] = rz.imresize_fused_gauss_cubic3D(vol, (18, 30, 13))
        vol4 = np.stack([vol, smooth_volume((18, 22, 26), 8)], axis=-1).astype(np.float64)
        out["vol4"] = vol4
        out["down4"] = rz.imresize_fused_gauss_cubic3D(vol4, (12, 14, 20))
        save("k1_resize", **out)

    # ---- K1 options off the flow path: per_axis, sigma_coeff, integer images (util/resize_util_3D.py:114-156)
    if want("k1_resize_opts"):
        vol = smooth_volume((18, 22, 26), 7)
        out = dict(vol=vol)
        out["per_axis"] = rz.imresize_fused_gauss_cubic3D(vol, (11, 22, 30), per_axis=True)
        out["per_axis_s09"] = rz.imresize_fused_gauss_cubic3D(vol, (9, 15, 13), sigma_coeff=0.9, per_axis=True)
        out["s03"] = rz.imresize_fused_gauss_cubic3D(vol, (11, 15, 17), sigma_coeff=0.3)
        u16 = (vol * 60000.0 + 2000.0).astype(np.uint16)
        out["u16"] = u16
        out["u16_down"] = rz.imresize_fused_gauss_cubic3D(u16, (11, 15, 17))
        out["u16_up"] = rz.imresize_fused_gauss_cubic3D(u16, (23, 28, 33))
        i16 = (vol * 60000.0 - 30000.0).astype(np.int16)
        out["i16"] = i16
        out["i16_mixed"] = rz.imresize_fused_gauss_cubic3D(i16, (18, 30, 13), per_axis=True)
        save("k1_resize_opts", **out)

    # ---- K2: warp (cubic B-spline with prefilter / linear) ---------------------------------
    if want("k2_warp"):
        out = {}
        f2 = np.stack([smooth_volume((14, 18, 20), 11), smooth_volume((14, 18, 20), 12)], -1).astype(np.float64)
        f1 = np.stack([smooth_volume((14, 18, 20), 13), smooth_volume((14, 18, 20), 14)], -1).astype(np.float64)
        rng = np.random.Generator(np.random.PCG64(5))
        u = (rng.random((14, 18, 20)) - 0.5) * 6.0   # forces out-of-bounds voxels too
        v = (rng.random((14, 18, 20)) - 0.5) * 6.0
        w = (rng.random((14, 18, 20)) - 0.5) * 6.0
        out.update(f2=f2, f1=f1, u=u, v=v, w=w)
        out["cubic"] = of.imregister_wrapper(f2, u, v, w, f1, "cubic")
        out["linear"] = of.imregister_wrapper(f2, u, v, w, f1, "linear")
        out["cubic_c1"] = of.imregister_wrapper(f2[..., 0], u, v, w, f1[..., 0])
        # scipy pieces in isolation (third-party arithmetic on the path)
        out["spline_in"] = f2[..., 0]
        out["spline_coef"] = spline_filter(np.pad(f2[..., 0], 12, mode="edge"), 3, output=np.float64, mode="nearest")
        save("k2_warp", **out)

    # ---- K3: motion tensor -------------------------------------------------------------------
    if want("k3_tensor"):
        f1 = smooth_volume((12, 15, 17), 21).astype(np.float64)
        f2 = moved(f1.astype(np.float32), (0.7, -0.4, 0.3), 0).astype(np.float64)
        hz, hy, hx = 32 / 12, 64 / 15, 64 / 17
        J = of.get_motion_tensor_gc(f1, f2, hz, hy, hx)
        save("k3_tensor", f1=f1, f2=f2, h=np.array([hz, hy, hx]), J=np.stack(J, 0))

    # ---- K4-K7: level solver -----------------------------------------------------------------
    if want("k7_solver"):
        out = {}
        f1 = smooth_volume((9, 11, 13), 31).astype(np.float64)
        f2 = moved(f1.astype(np.float32), (0.5, -0.3, 0.2), 0).astype(np.float64)
        g1 = smooth_volume((9, 11, 13), 33).astype(np.float64)
        g2 = moved(g1.astype(np.float32), (0.5, -0.3, 0.2), 0).astype(np.float64)
        hz, hy, hx = 1.9, 2.3, 2.6
        Ja = of.get_motion_tensor_gc(f1, f2, hz, hy, hx)
        Jb = of.get_motion_tensor_gc(g1, g2, hz, hy, hx)
        P, M, N = Ja[0].shape
        rng = np.random.Generator(np.random.PCG64(9))
        u = of.add_boundary((rng.random((9, 11, 13)) - 0.5).astype(np.float32).astype(np.float64))
        v = of.add_boundary((rng.random((9, 11, 13)) - 0.5).astype(np.float32).astype(np.float64))
        w = of.add_boundary((rng.random((9, 11, 13)) - 0.5).astype(np.float32).astype(np.float64))
        out.update(u=u, v=v, w=w, h=np.array([hx, hy, hz]))
        # C = 1
        J1 = [np.ascontiguousarray(j[..., None]) for j in Ja]
        wt1 = np.pad(np.ones((9, 11, 13, 1)), ((1, 1), (1, 1), (1, 1), (0, 0)))
        out["J_c1"] = np.stack(J1, 0)
        out["wt_c1"] = wt1
        out["c1_a045_s1"] = ls.compute_flow_3d(*J1, wt1, u, v, w, 0.25, 0.3, 0.35, 12, 5,
                                               np.array([0.45]), 1.0, hx, hy, hz)
        out["c1_a1_s1"] = ls.compute_flow_3d(*J1, wt1, u, v, w, 0.25, 0.3, 0.35, 7, 3,
                                             np.array([1.0]), 1.0, hx, hy, hz)
        out["c1_a045_s05"] = ls.compute_flow_3d(*J1, wt1, u, v, w, 0.25, 0.3, 0.35, 6, 2,
                                                np.array([0.45]), 0.5, hx, hy, hz)
        # C = 2
        J2 = [np.ascontiguousarray(np.stack([a, b], -1)) for a, b in zip(Ja, Jb)]
        wt2 = np.pad(np.ones((9, 11, 13, 2)) * np.array([0.6, 0.4]), ((1, 1), (1, 1), (1, 1), (0, 0)))
        out["J_c2"] = np.stack(J2, 0)
        out["wt_c2"] = wt2
        out["c2_a045_s1"] = ls.compute_flow_3d(*J2, wt2, u, v, w, 0.25, 0.3, 0.35, 10, 5,
                                               np.array([0.45, 0.6]), 1.0, hx, hy, hz)
        save("k7_solver", **out)

    # ---- K8: median --------------------------------------------------------------------------
    if want("k8_median"):
        rng = np.random.Generator(np.random.PCG64(3))
        a = rng.standard_normal((9, 12, 7))
        b = rng.standard_normal((6, 6, 6))
        save("k8_median", a=a, a_med=median_filter(a, size=(5, 5, 5), mode="mirror"),
             b=b, b_med=median_filter(b, size=(5, 5, 5), mode="mirror"))

    # ---- schedule ----------------------------------------------------------------------------
    if want("schedule"):
        rows = []
        for (p, m, n, eta, levels) in [(32, 64, 64, 0.8, 2), (256, 256, 256, 0.8, 4), (512, 512, 512, 0.8, 5),
                                      (256, 512, 512, 0.8, 8), (256, 512, 512, 0.8, 100), (12, 16, 16, 0.8, 50),
                                      (20, 28, 28, 0.75, 50), (5, 40, 40, 0.8, 3), (30, 30, 30, 0.5, 10)]:
            rows.append([p, m, n, eta, levels, of.warpingDepth(eta, levels, p, m, n)])
        save("schedule", rows=np.array(rows, dtype=np.float64),
             rounds=np.array([[x, round(x)] for x in (0.5, 1.5, 2.5, 40.96, 51.2, 20.48, 10.5, 11.5)], dtype=np.float64))

    # ---- f-1: preprocessing (normalize + gaussian), reference util/image_processing_3D.py -------
    if want("f1_preproc"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("im3d", os.path.join(REF_SRC, "flowreg3d", "util",
                                                                           "image_processing_3D.py"))
        im3d = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(im3d)
        rng = np.random.Generator(np.random.PCG64(77))
        batch = (rng.random((4, 7, 9, 11, 2)) * 900 + 50)
        ref = (rng.random((7, 9, 11, 2)) * 1000)
        sigma = np.array([[1.0, 1.2, 0.8, 0.6], [0.5, 2.0, 0.7, 0.1]])
        out = dict(batch=batch, ref=ref, sigma=sigma, batch_u16=batch.astype(np.uint16))
        for cn in ("together", "separate"):
            n5 = im3d.normalize(batch, ref=ref, channel_normalization=cn)
            out[f"norm5_{cn}"] = n5
            out[f"filt5_{cn}"] = im3d.apply_gaussian_filter(n5, sigma, mode="reflect", truncate=4.0)
            n4 = im3d.normalize(ref, ref=None, channel_normalization=cn)
            out[f"filt4_{cn}"] = im3d.apply_gaussian_filter(n4, sigma, mode="reflect", truncate=4.0)
        nu = im3d.normalize(batch.astype(np.uint16), ref=ref, channel_normalization="together")
        out["filt5_u16"] = im3d.apply_gaussian_filter(nu, np.array([1.0, 1.0, 1.0, 0.1]), mode="reflect", truncate=4.0)
        save("f1_preproc", **out)

    # ---- the executor boundary: the reference's own SequentialExecutor3D.process_batch ------------
    # (motion_correction/parallelization/sequential_3d.py:37-175) with the reference's
    # get_displacement / imregister_wrapper injected, exactly as BatchMotionCorrector calls it
    # (compensate_recording_3D.py:301-340); the flow_params dict has the pipeline's keys.
    if want("ex_seq"):
        from flowreg3d.motion_correction.parallelization.sequential_3d import SequentialExecutor3D
        from scipy.ndimage import gaussian_filter
        out = {}

        def series(shape, C, T, seed):
            """raw series (T,Z,Y,X,C) float64 in camera-like counts + its fixed reference"""
            ref = np.stack([smooth_volume(shape, seed + c, sigma=2.0) for c in range(C)], -1)
            vols = []
            for t in range(T):
                sh = (0.9 * np.cos(0.9 * t), -0.6 + 0.2 * t, 0.4 * np.sin(1.3 * t + 0.5))
                vols.append(np.stack([moved(ref[..., c], sh, 0) for c in range(C)], -1))
            raw = np.stack(vols, 0).astype(np.float64) * 3000.0 + 200.0
            return raw, ref.astype(np.float64) * 3000.0 + 200.0

        def proc_of(a, lo, hi):
            # what the pipeline feeds as batch_proc: normalised to the reference's range, float64
            return (a - lo) / (hi - lo + 1e-8)

        ex = SequentialExecutor3D(n_workers=1)
        cases = {
            "c1_f32": dict(shape=(12, 18, 18), C=1, T=3, dtype=np.float32, method="cubic",
                           fp=dict(alpha=(0.25, 0.25, 0.25), weight=np.array([1.0]), levels=50, min_level=0, eta=0.8,
                                   update_lag=5, iterations=20, a_smooth=1.0, a_data=0.45)),
            "c2_u16": dict(shape=(12, 16, 20), C=2, T=2, dtype=np.uint16, method="cubic",
                           fp=dict(alpha=(0.3, 0.25, 0.2), weight=np.array([0.6, 0.4]), levels=50, min_level=0, eta=0.8,
                                   update_lag=5, iterations=15, a_smooth=1.0, a_data=0.45)),
            "c1_f64_lin": dict(shape=(10, 14, 16), C=1, T=2, dtype=np.float64, method="linear",
                               fp=dict(alpha=(0.25, 0.25, 0.25), weight=np.array([1.0]), levels=50, min_level=1, eta=0.8,
                                       update_lag=4, iterations=12, a_smooth=1.0, a_data=0.45)),
        }
        names = []
        for name, cs in cases.items():
            t0 = time.time()
            raw64, ref_raw = series(cs["shape"], cs["C"], cs["T"], 300 + len(names) * 10)
            batch = raw64.astype(cs["dtype"])            # integer dtypes truncate like a camera would deliver
            lo, hi = ref_raw.min(), ref_raw.max()
            batch_proc = proc_of(batch.astype(np.float64), lo, hi)
            ref_proc = proc_of(ref_raw, lo, hi)
            rng = np.random.Generator(np.random.PCG64(55))
            w_init = np.stack([gaussian_filter(0.4 * (rng.random(cs["shape"]) - 0.5), 2.0) for _ in range(3)], -1)
            w_init = w_init.astype(np.float32)
            calls = []
            reg, flows = ex.process_batch(batch, batch_proc, ref_raw, ref_proc, w_init, of.get_displacement,
                                          of.imregister_wrapper, interpolation_method=cs["method"],
                                          progress_callback=lambda n: calls.append(n), flow_params=dict(cs["fp"]))
            assert reg.dtype == batch.dtype and flows.dtype == np.float32 and sum(calls) == cs["T"]
            fp = cs["fp"]
            out[f"{name}_batch"] = batch
            out[f"{name}_batch_proc"] = batch_proc
            out[f"{name}_ref_raw"] = ref_raw
            out[f"{name}_ref_proc"] = ref_proc
            out[f"{name}_w_init"] = w_init
            out[f"{name}_registered"] = reg
            out[f"{name}_flows"] = flows
            out[f"{name}_weight"] = np.asarray(fp["weight"], np.float64)
            out[f"{name}_params"] = np.array([fp["alpha"][0], fp["alpha"][1], fp["alpha"][2], fp["update_lag"],
                                              fp["iterations"], fp["min_level"], fp["levels"], fp["eta"],
                                              fp["a_smooth"], fp["a_data"], 3 if cs["method"] == "cubic" else 1],
                                             dtype=np.float64)
            names.append(name)
            print(f"    ex_seq/{name}: {time.time() - t0:.1f}s, registered {reg.dtype}, mean flow {flows.mean(axis=(0, 1, 2, 3))}")
        out["cases"] = np.array(names)
        save("ex_seq", **out)

    # ---- f-4: update_reference arithmetic (compensate_recording_3D.py:395-429): per channel, warp the
    # last <= 100 batch_proc volumes by their flows (imregister_wrapper, fp32 result), mean over them in fp64
    if want("f4_update_ref"):
        from scipy.ndimage import gaussian_filter
        shape, C, T = (10, 14, 12), 2, 5
        rng = np.random.Generator(np.random.PCG64(91))
        bp = np.stack([np.stack([smooth_volume(shape, 500 + 7 * t + c, 1.5) for c in range(C)], -1) for t in range(T)],
                      0).astype(np.float64)
        ref_proc = np.stack([smooth_volume(shape, 490 + c, 1.5) for c in range(C)], -1).astype(np.float64)
        w = np.stack([np.stack([gaussian_filter(5.0 * (rng.random(shape) - 0.5), 1.5) for _ in range(3)], -1)
                      for t in range(T)], 0).astype(np.float32)
        res = {}
        for method in ("cubic", "linear"):
            new_ref = np.zeros_like(ref_proc)
            n_ref = min(100, T)
            start = T - n_ref
            for c in range(C):
                comp = np.zeros((n_ref,) + shape, dtype=np.float64)
                for t in range(n_ref):
                    comp[t] = of.imregister_wrapper(bp[start + t, ..., c], w[start + t, ..., 0], w[start + t, ..., 1],
                                                    w[start + t, ..., 2], ref_proc[..., c], interpolation_method=method)
                new_ref[..., c] = np.mean(comp, axis=0)
            res[method] = new_ref
        save("f4_update_ref", batch_proc=bp, ref_proc=ref_proc, w=w, new_ref_cubic=res["cubic"],
             new_ref_linear=res["linear"])

    # ---- end to end --------------------------------------------------------------------------
    def e2e(name, shape, C, shift, kw, uvw_amp=0.0, weight=None):
        t0 = time.time()
        chans_f, chans_m = [], []
        for c in range(C):
            f = smooth_volume(shape, 100 + c, sigma=2.0)
            chans_f.append(f)
            chans_m.append(moved(f, shift, 0))
        fixed = np.stack(chans_f, -1)
        moving = np.stack(chans_m, -1)
        if C == 1:
            fixed, moving = fixed[..., 0], moving[..., 0]
        uvw = None
        if uvw_amp:
            rng = np.random.Generator(np.random.PCG64(17))
            uvw = (uvw_amp * (rng.random(shape + (3,)) - 0.5)).astype(np.float64)
            from scipy.ndimage import gaussian_filter
            for d in range(3):
                uvw[..., d] = gaussian_filter(uvw[..., d], 2.0) + 0.5 * shift[d]
        flow = of.get_displacement(fixed, moving, uvw=None if uvw is None else uvw.copy(), weight=weight, **kw)
        arrs = dict(fixed=fixed, moving=moving, flow=flow,
                    params=np.array([kw["alpha"][0], kw["alpha"][1], kw["alpha"][2], kw["update_lag"],
                                     kw["iterations"], kw["min_level"], kw["levels"], kw["eta"],
                                     kw["a_smooth"], kw["a_data"]], dtype=np.float64))
        if uvw is not None:
            arrs["uvw"] = uvw
        if weight is not None:
            arrs["weight"] = np.asarray(weight, dtype=np.float64)
        save(name, **arrs)
        print(f"    {name}: {time.time() - t0:.1f}s  mean flow {flow.mean(axis=(0, 1, 2))}")

    base = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=20, min_level=0, levels=50,
                eta=0.8, a_smooth=1.0, a_data=0.45)
    if want("e2e_small"):
        e2e("e2e_small", (12, 16, 16), 1, (0.8, -0.5, 0.3), dict(base, iterations=8, update_lag=3))
    if want("e2e_c2"):
        e2e("e2e_c2", (14, 20, 18), 2, (0.9, -0.6, 0.4), dict(base, iterations=10, alpha=(0.3, 0.25, 0.2)),
            uvw_amp=0.5, weight=np.array([0.7, 0.3]))
    if want("e2e_minlevel"):
        e2e("e2e_minlevel", (16, 24, 24), 1, (1.2, -0.7, 0.5), dict(base, iterations=10, min_level=1))
    if want("e2e_asmooth"):
        e2e("e2e_asmooth", (12, 16, 16), 1, (0.8, -0.5, 0.3), dict(base, iterations=6, a_smooth=0.5, alpha=(2, 2, 2)))
    if want("e2e_cfg5like"):
        # reduced BASELINE config 5: two channels (weights 0.5/0.5), expansion/contraction + rotations,
        # synthetic pair from the build's own generator (flowreg3d_amd/synthetic.py)
        sys.path.insert(0, ROOT)
        from flowreg3d_amd.synthetic import make_pair
        fixed, moving, _ = make_pair((24, 48, 48), seed=1234, channels=2, motion="expansion", scale=1.0)
        kw = dict(base, iterations=60, levels=4, weight=np.array([0.5, 0.5]))
        t0 = time.time()
        flow = of.get_displacement(fixed, moving, **kw)
        save("e2e_cfg5like", fixed=fixed, moving=moving, flow=flow, weight=kw["weight"],
             params=np.array([0.25, 0.25, 0.25, 5, 60, 0, 4, 0.8, 1.0, 0.45]))
        print(f"    e2e_cfg5like: {time.time() - t0:.1f}s")
    if want("e2e_cfg1"):
        # BASELINE config 1: 32x64x64 (Z,Y,X), levels=2 -> 3 solves; iterations=20 keeps the
        # pure-Python run affordable
        e2e("e2e_cfg1", (32, 64, 64), 1, (1.7, -1.1, 0.6), dict(base, levels=2))


if __name__ == "__main__":
    main()
