"""-m gpu: flow parity AT FULL SIZE for BASELINE configurations 2, 3 and 5 against the CPU oracle, and for config 2's
volume with get_displacement's own default a_smooth = 0.5 (the psi_smooth solver path; measured 1.6e-5).

The oracle (oracle/fr3d_oracle.c, pinned to the reference by tests/golden/*.npz) needs 4-30 minutes
and up to 25 GB per volume at these sizes, so it was run ONCE in the build container on the
deterministic synthetic inputs of flowreg3d_amd.synthetic.fullsize_case(); a strided lattice
(every 8th voxel per axis) and one central 32^3 block of its flow field are committed as
tests/golden/fullsize_<cfg>.npz together with the SHA-256 of the inputs
(tools/gen_fullsize_golden.py).  Here the same inputs are regenerated, their checksum verified,
the HIP path runs in its DEFAULT solver mode (FR3D_SOLVER_AUTO, what the Python mirror, the executor and
bench.py pass: fp32 solver storage with fp64 update arithmetic for one channel up to 2^25 voxels -- config 2;
fp64 storage for larger volumes -- config 3 -- and for several channels -- config 5) at the full 100
iterations, and the flow is compared on the sample.  The fp32-storage mode at 512^3, which bench.py also
times, is measured by its own test below: 1.5e-4, above the bound, which is why AUTO leaves it at that size.

Tolerance: mean end-point error < 1e-4 voxels (BASELINE.json north_star), on the lattice and on the
block, for configs 2 and 3; for config 5 the reference's own reproducibility at that size (6.0e-4 between two
builds of the CPU path, see CFG5_CPU_REPRODUCIBILITY below; the GPU measures 2.8e-4).  The maxima are reported in
the assertion message and bounded loosely (single voxels next to flat regions are ill-conditioned in the
reference iteration itself).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL_MEAN = 1e-4
# Config 5 (two channels, update_lag 5) is ill-conditioned at this level in the reference iteration itself: the same
# CPU source rebuilt with FMA contraction (the reassociation numba's fastmath=True allows the reference) differs from
# the committed sample by 5.98e-4 mean / 2.2e-2 max at full size (profiles/r02/cfg5_oracle_reproducibility.json).  The
# bound for the GPU path there is the CPU path's own reproducibility, not 1e-4; measured: 2.8e-4.
CFG5_CPU_REPRODUCIBILITY = 5.98e-4


def _digest(fixed, moving):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(fixed).tobytes())
    h.update(np.ascontiguousarray(moving).tobytes())
    return h.hexdigest()


def _load(case):
    path = os.path.join(GOLDEN, f"fullsize_{case}.npz")
    g = np.load(path)
    meta = json.loads(bytes(g["meta"]).decode())
    return g, meta


def _epe(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return float(d.mean()), float(d.max())


@pytest.mark.parametrize("case", ["cfg2", "cfg2_asmooth05", "cfg3", "cfg5"])
def test_fullsize_flow_matches_oracle_sample(hip, case):
    import flowreg3d_amd as fr
    from flowreg3d_amd.synthetic import fullsize_case
    g, meta = _load(case)
    fixed, moving, gt, kw = fullsize_case(case)
    assert list(fixed.shape[:3]) == meta["shape_zyx"]
    assert _digest(fixed, moving) == meta["inputs_sha256"], "synthetic inputs differ from the ones the oracle ran on"
    assert kw["iterations"] == 100 == meta["params"]["iterations"]
    flow = fr.get_displacement(fixed, moving, **kw)  # default solver mode, full iterations
    assert flow.shape == tuple(meta["shape_zyx"]) + (3,) and np.isfinite(flow).all()
    st, bl = meta["stride"], meta["block"]
    z0, y0, x0 = meta["block_origin_zyx"]
    lat_mean, lat_max = _epe(flow[::st, ::st, ::st], g["lattice"])
    blk_mean, blk_max = _epe(flow[z0:z0 + bl, y0:y0 + bl, x0:x0 + bl], g["block"])
    # the lattice includes the volume faces; its interior separately (crop one lattice step)
    int_mean, _ = _epe(flow[::st, ::st, ::st][1:-1, 1:-1, 1:-1], g["lattice"][1:-1, 1:-1, 1:-1])
    msg = (f"{case}: EPE vs oracle lattice mean {lat_mean:.3e} max {lat_max:.3e}, interior lattice mean "
           f"{int_mean:.3e}, central block mean {blk_mean:.3e} max {blk_max:.3e}")
    print(msg)
    tol = CFG5_CPU_REPRODUCIBILITY if case == "cfg5" else TOL_MEAN
    assert lat_mean < tol and blk_mean < tol and int_mean < tol, msg
    assert lat_max < 0.25 and blk_max < 0.05, msg
    # the GPU solves the same problem as the CPU path: same error against the synthetic ground truth
    gpu_gt, _ = _epe(flow[::st, ::st, ::st][1:-1, 1:-1, 1:-1], g["gt_lattice"][1:-1, 1:-1, 1:-1])
    cpu_gt, _ = _epe(g["lattice"][1:-1, 1:-1, 1:-1], g["gt_lattice"][1:-1, 1:-1, 1:-1])
    assert abs(gpu_gt - cpu_gt) < 1e-3 * max(1.0, cpu_gt), (gpu_gt, cpu_gt)


def test_cfg3_fp32_storage_is_measured_and_stated(hip):
    """512^3 with fp32 solver storage (solver_fp64=1, the mode the 76 B / update roofline figure is defined on and
    bench.py's `cfg3_fp32_storage` leg times): mean EPE against the CPU sample 1.5e-4 -- ABOVE the 1e-4 bound, stated
    as such in DESIGN.md and in the bench line; increments, frozen system and factors each cost about 1e-4 at this size
    when held in fp32 (profiles/r02/numerics_512_rounding_groups.md).  The test pins the measured level (< 2e-4)."""
    import flowreg3d_amd as fr
    from flowreg3d_amd.synthetic import fullsize_case
    g, meta = _load("cfg3")
    fixed, moving, gt, kw = fullsize_case("cfg3")
    assert _digest(fixed, moving) == meta["inputs_sha256"]
    flow = fr.get_displacement(fixed, moving, solver_fp64=1, **kw)
    st = meta["stride"]
    lat_mean, lat_max = _epe(flow[::st, ::st, ::st], g["lattice"])
    print(f"cfg3, fp32 solver storage: EPE vs oracle lattice mean {lat_mean:.3e} max {lat_max:.3e}")
    assert 5e-5 < lat_mean < 2e-4, lat_mean
