"""CPU: the batch driver pinned to the REFERENCE.  tests/golden/drv_*.npz hold the outputs of the
reference's own compensate_arr_3D (tools/gen_driver_golden.py); here the driver loop restated on the CPU oracle
(tests/driver_cases.py) must reproduce them -- registered series, flows, the four per-volume statistics and the final
w_init -- at the reference's cross-executor tolerance (rtol 1e-5 / atol 1e-6 on `registered`,
tests/motion_correction/test_parallelization.py:192-198).  That makes the loop logic (bootstrap branch, rolling w_init,
update_initialization_w, typed output, squeeze rules) part of the pinned oracle; the -m gpu test then holds the product's
pipeline.compensate_arr_3D to the same fixtures."""
import numpy as np
import pytest

from driver_cases import CASES, check_against_reference, load_case, oracle_driver


@pytest.mark.parametrize("name", CASES)
def test_oracle_driver_reproduces_reference_compensate_arr(oracle, name):
    g, meta, opt = load_case(name)
    reg, w, stats, w_init = oracle_driver(oracle, g["video"], g["reference"], opt)
    check_against_reference(g, reg, w, stats, w_init, 5e-6, name + " (oracle loop)")


def test_fixture_branches_are_the_intended_ones():
    """the fixtures really exercise the branches they are named after"""
    g, meta, opt = load_case("drv_t3_serial")
    assert g["video"].ndim == 4 and g["reference"].ndim == 3 and g["registered"].ndim == 4  # squeeze path
    assert g["video"].shape[0] <= 4                                                        # serial bootstrap (:379-385)
    g, meta, opt = load_case("drv_t7_b5")
    assert min(22, opt.buffer_size) > 4 and g["video"].shape[0] > opt.buffer_size           # executor bootstrap + 2 batches
    g, meta, opt = load_case("drv_t7_b3")
    assert g["video"].shape[0] == 7 and opt.buffer_size == 3 and opt.interpolation_method == "linear"
    g, meta, opt = load_case("drv_noinit")
    assert opt.update_initialization_w is False
    g, meta, opt = load_case("drv_c2_u16")
    assert g["video"].dtype == np.uint16 and g["registered"].dtype == np.uint16 and g["video"].shape[-1] == 2
    assert list(g["progress"][-1]) == [3, 3]
