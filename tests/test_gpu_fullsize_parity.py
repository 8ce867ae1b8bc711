"""-m gpu: flow parity AT FULL SIZE for BASELINE configurations 2, 3 and 5 against the CPU oracle.

The oracle (oracle/fr3d_oracle.c, pinned to the reference by tests/golden/*.npz) needs 4-50 minutes
and up to 40 GB per volume at these sizes, so it was run ONCE per case in the build container on the
deterministic synthetic inputs of flowreg3d_amd.synthetic.fullsize_case(); a strided lattice
(every 8th voxel per axis) and one central 32^3 block of its flow field are committed as
tests/golden/fullsize_<case>.npz together with the SHA-256 of the inputs (tools/gen_fullsize_golden.py).
Here the same inputs are regenerated, their checksum verified, the HIP path runs in its DEFAULT solver mode
(FR3D_SOLVER_AUTO, what the Python mirror, the executor and bench.py pass) at the full 100 iterations, and the flow
is compared on the sample.

Cases
  cfg2_recipe, cfg3_recipe   the inputs bench.py times: SURVEY section 8d's recipe (blurred PCG64 noise + 8 blobs,
                             translation (1.7,-1.1,0.6) + 1.5 degree rotation about z, cubic backward warp)
  cfg2_recipe_s135           the same at 1.35x the motion: the largest time point of bench.py's synthetic series
  cfg2, cfg3                 the O(N) stand-in inputs of round 2 (periodic texture, pure translation)
  cfg2_asmooth05             config 2's stand-in volume with get_displacement's own default a_smooth = 0.5 (psi_smooth path)
  cfg5                       256x512x512, two channels, 13-level pyramid
  cfg5_levels8               the survey's own config-5 schedule (levels=8: 9 solves)

AUTO resolves to packed 42-bit solver storage for single-channel volumes above 2^22 voxels (256^3 and 512^3 here; fp32
storage with fp64 update arithmetic below) and to fp64 storage for several channels (config 5).  The fp32-storage mode
-- the one SURVEY 8d's 76 B / update figure is defined on, timed by bench.py beside the default -- is measured too:
8.6e-5 at 256^3 on the recipe inputs (inside the bound, by 10 %), 1.5e-4 at 512^3 (outside).

Tolerance: mean end-point error < 1e-4 voxels (BASELINE.json north_star) on the lattice, its interior and the block, for
EVERY configuration.  Config 5 (two channels) meets it since the level tail is evaluated like the reference's (fp64
median, one rounding of u + du: k_median.hip k_median5_refine): 2e-11 with fp64 storage -- it measured 2.8e-4 while the
increments were rounded to fp32 before the median, a double rounding of the level flow that the two-channel iteration
amplifies.  Packed storage on config 5 measures 2.5e-4 (pinned below): why AUTO takes fp64
storage for several channels.
Every run appends its measured figures to gpurun_out/parity_fullsize.json (merged back by gpurun; the committed copy is
profiles/parity_fullsize.json, which bench.py quotes).
"""
import hashlib
import json
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

TOL_MEAN = 1e-4
SOLVER_MODE_NAMES = {0: "fp32 storage, fp32 arithmetic", 1: "fp32 storage, fp64 arithmetic", 2: "fp64 storage",
                     3: "packed 42-bit storage, fp64 arithmetic"}


def _digest(fixed, moving):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(fixed).tobytes())
    h.update(np.ascontiguousarray(moving).tobytes())
    return h.hexdigest()


def _load(case):
    path = os.path.join(GOLDEN, f"fullsize_{case}.npz")
    if not os.path.exists(path):
        pytest.skip(f"no committed oracle sample for {case}")
    g = np.load(path)
    meta = json.loads(bytes(g["meta"]).decode())
    return g, meta


def _epe(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return float(d.mean()), float(d.max())


def _auto_mode(shape, channels, a_smooth):
    nvox = int(np.prod(shape[:3]))
    if a_smooth != 1.0:
        return 2 if (channels >= 2 or nvox > (1 << 25)) else 1
    return 2 if channels >= 2 else (3 if nvox > (1 << 22) else 1)


def _record(entry):
    """append one measurement to gpurun_out/parity_fullsize.json (keyed by case/mode)"""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_fullsize.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        rec = {}
    rec[f"{entry['case']}/mode{entry['solver_mode']}"] = entry
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)


def _measure(case, solver_fp64=None):
    import flowreg3d_amd as fr
    from flowreg3d_amd.synthetic import fullsize_case
    g, meta = _load(case)
    fixed, moving, gt, kw = fullsize_case(case, warp=fr.imregister_wrapper)
    assert list(fixed.shape[:3]) == meta["shape_zyx"]
    if "_recipe" in case:
        # the moving volume comes out of the engine's own cubic warp here and out of the CPU oracle's in the fixture: the
        # fixed volume by checksum, the moving volume on the lattice to the last float32 bits
        assert hashlib.sha256(np.ascontiguousarray(fixed).tobytes()).hexdigest() == meta["fixed_sha256"]
        st_ = meta["stride"]
        dm = np.abs(moving[::st_, ::st_, ::st_].astype(np.float64) - g["moving_lattice"].astype(np.float64))
        assert dm.max() <= 2.5e-7 and (dm > 0).mean() < 0.05, (dm.max(), (dm > 0).mean())
    else:
        assert _digest(fixed, moving) == meta["inputs_sha256"], "synthetic inputs differ from the ones the oracle ran on"
    assert kw["iterations"] == 100 == meta["params"]["iterations"] and kw["levels"] == meta["params"]["levels"]
    t0 = time.perf_counter()
    flow = fr.get_displacement(fixed, moving, solver_fp64=solver_fp64, **kw)  # full iterations
    dt = time.perf_counter() - t0
    assert flow.shape == tuple(meta["shape_zyx"]) + (3,) and np.isfinite(flow).all()
    st, bl = meta["stride"], meta["block"]
    z0, y0, x0 = meta["block_origin_zyx"]
    lat_mean, lat_max = _epe(flow[::st, ::st, ::st], g["lattice"])
    blk_mean, blk_max = _epe(flow[z0:z0 + bl, y0:y0 + bl, x0:x0 + bl], g["block"])
    # the lattice includes the volume faces; its interior separately (crop one lattice step)
    int_mean, _ = _epe(flow[::st, ::st, ::st][1:-1, 1:-1, 1:-1], g["lattice"][1:-1, 1:-1, 1:-1])
    gpu_gt, _ = _epe(flow[::st, ::st, ::st][1:-1, 1:-1, 1:-1], g["gt_lattice"][1:-1, 1:-1, 1:-1])
    cpu_gt, _ = _epe(g["lattice"][1:-1, 1:-1, 1:-1], g["gt_lattice"][1:-1, 1:-1, 1:-1])
    channels = 1 if fixed.ndim == 3 else fixed.shape[3]
    mode = solver_fp64 if solver_fp64 is not None else _auto_mode(fixed.shape, channels, kw["a_smooth"])
    entry = {"case": case, "solver_mode": int(mode), "solver": SOLVER_MODE_NAMES[int(mode)], "auto": solver_fp64 is None,
             "shape_zyx": meta["shape_zyx"], "channels": channels, "levels": kw["levels"], "a_smooth": kw["a_smooth"],
             "lattice_mean_epe": lat_mean, "lattice_max_epe": lat_max, "interior_lattice_mean_epe": int_mean,
             "block_mean_epe": blk_mean, "block_max_epe": blk_max, "gpu_vs_ground_truth": gpu_gt,
             "cpu_vs_ground_truth": cpu_gt, "oracle_seconds_1core": meta.get("oracle_seconds_1core"),
             "gpu_call_seconds_incl_host_copies": dt, "inputs_sha256": meta["inputs_sha256"][:16]}
    _record(entry)
    msg = (f"{case} [{entry['solver']}]: EPE vs oracle lattice mean {lat_mean:.3e} max {lat_max:.3e}, interior lattice mean "
           f"{int_mean:.3e}, central block mean {blk_mean:.3e} max {blk_max:.3e}; vs ground truth GPU {gpu_gt:.5f} CPU {cpu_gt:.5f}")
    print(msg)
    return entry, msg


@pytest.mark.parametrize("case", ["cfg2_recipe", "cfg2_recipe_s135", "cfg2", "cfg2_asmooth05", "cfg3_recipe", "cfg3", "cfg5",
                                  "cfg5_levels8", "thr_160x176x176", "thr_200"])
def test_fullsize_flow_matches_oracle_sample(hip, case):
    e, msg = _measure(case)
    tol = TOL_MEAN
    assert e["lattice_mean_epe"] < tol and e["block_mean_epe"] < tol and e["interior_lattice_mean_epe"] < tol, msg
    assert e["lattice_max_epe"] < 0.25 and e["block_max_epe"] < 0.05, msg
    # the GPU solves the same problem as the CPU path: same error against the synthetic ground truth
    assert abs(e["gpu_vs_ground_truth"] - e["cpu_vs_ground_truth"]) < 1e-3 * max(1.0, e["cpu_vs_ground_truth"]), msg


@pytest.mark.parametrize("case,mode,lo,hi", [("cfg3", 1, 5e-5, 2e-4), ("cfg3", 2, 0.0, 1e-7), ("cfg2_recipe", 1, 3e-5, 1e-4),
                                              ("cfg2_recipe_s135", 1, 3e-5, 1.5e-4), ("cfg2_recipe", 2, 0.0, 1e-7),
                                              ("cfg5", 3, 1e-4, 3.5e-4), ("cfg5_levels8", 3, 3e-5, 2e-4)])
def test_other_storage_modes_are_measured_and_stated(hip, case, mode, lo, hi):
    """The storage modes AUTO does not pick, timed by bench.py beside the packed mode.  fp32 storage (solver_fp64=1, the
    mode SURVEY 8d's 76 B / update figure is defined on): 8.6e-5 at 256^3 on the recipe inputs (lattice mean; the central
    block measures 1.03e-4) and 1.5e-4 at 512^3 -- ABOVE the 1e-4 bound; increments, frozen system and factors each cost
    about 1e-4 at that size when held in fp32 (profiles/r02/numerics_512_rounding_groups.md).  fp64 storage: 2e-9 ... 4e-9
    (what is left is the fp32 rounding of the output).  Packed storage on the two-channel config 5: 2.5e-4, outside the
    bound (AUTO takes fp64 storage there).  The test pins the measured levels."""
    e, msg = _measure(case, solver_fp64=mode)
    assert lo <= e["lattice_mean_epe"] < hi, msg
