"""CPU: the committed full-size oracle samples (tests/golden/fullsize_*.npz) are well-formed and the
synthetic generator still produces the inputs they were computed from (checked on the 256^3 case;
the GPU tests verify the checksum of every case before comparing)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = [c for c in ("cfg2", "cfg2_asmooth05", "cfg3", "cfg5", "thr_160x176x176", "thr_200")
         if os.path.exists(os.path.join(GOLDEN, f"fullsize_{c}.npz"))]


@pytest.mark.parametrize("case", CASES)
def test_fixture_is_well_formed(case):
    g = np.load(os.path.join(GOLDEN, f"fullsize_{case}.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    Z, Y, X = meta["shape_zyx"]
    st, bl = meta["stride"], meta["block"]
    assert g["lattice"].shape == (-(-Z // st), -(-Y // st), -(-X // st), 3)
    assert g["gt_lattice"].shape == g["lattice"].shape
    assert g["block"].shape == (bl, bl, bl, 3) and g["block"].dtype == np.float64
    assert np.isfinite(g["lattice"]).all() and np.isfinite(g["block"]).all()
    assert meta["params"]["iterations"] == 100 and len(meta["inputs_sha256"]) == 64
    # the oracle solved the synthetic motion: sub-voxel error against the ground truth in the interior (config 5:
    # 0.58, of which 0.57 is the difference between the generating field and the backward map)
    assert meta["epe_oracle_vs_gt_mean_interior8"] < 1.0


def test_all_cases_are_committed():
    # (thr_*: volumes just above the 2^22-voxel switch of FR3D_SOLVER_AUTO from fp32 to packed solver storage)
    assert CASES == ["cfg2", "cfg2_asmooth05", "cfg3", "cfg5", "thr_160x176x176", "thr_200"]
    meta = json.loads(bytes(np.load(os.path.join(GOLDEN, "fullsize_cfg2_asmooth05.npz"))["meta"]).decode())
    assert meta["params"]["a_smooth"] == 0.5 and meta["params"]["levels"] == 4


def test_cfg2_inputs_reproduce():
    from flowreg3d_amd.synthetic import fullsize_case
    meta = json.loads(bytes(np.load(os.path.join(GOLDEN, "fullsize_cfg2.npz"))["meta"]).decode())
    fixed, moving, _, kw = fullsize_case("cfg2")
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(fixed).tobytes())
    h.update(np.ascontiguousarray(moving).tobytes())
    assert h.hexdigest() == meta["inputs_sha256"]
    assert kw["levels"] == meta["params"]["levels"] == 4
