"""CPU: the portable pow (flowreg3d_amd/csrc/portable_pow.h: plain arithmetic, same source for gcc and hipcc) that the
verification mode and the `ppow` oracle build use in the psi nonlinearities -- accuracy against libm over the ranges the
solver feeds it, and the `ppow` oracle against the default oracle on a reference golden."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from conftest import ROOT, golden, params_of

SRC = r"""
#include "%s/flowreg3d_amd/csrc/portable_pow.h"
void ppow_many(const double *x, const double *y, int n, double *out) { for (int i = 0; i < n; i++) out[i] = fr3d_ppow(x[i], y[i]); }
"""


def _build():
    d = tempfile.mkdtemp(prefix="ppow_")
    c = os.path.join(d, "ppow.c")
    so = os.path.join(d, "ppow.so")
    with open(c, "w") as fh:
        fh.write(SRC % ROOT)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, c])
    lib = C.CDLL(so)
    lib.ppow_many.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_int, C.POINTER(C.c_double)]
    return lib


def test_portable_pow_is_within_a_few_ulp_of_libm():
    lib = _build()
    rng = np.random.default_rng(0)
    n = 400000
    # psi arguments: val + 1e-6 with val from 0 to large residuals; exponents a - 1 for a in (0, 1]
    x = np.concatenate([10.0 ** rng.uniform(-6, 8, n), 1e-6 + rng.random(n) * 1e-5, 1.0 + rng.normal(0, 1e-3, n) ** 2])
    y = np.concatenate([rng.uniform(-1.0, 0.0, n), np.full(n, -0.55), rng.choice([-0.55, -0.5, -0.9, -0.1], n)])
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    lib.ppow_many(x.ctypes.data_as(dp), y.ctypes.data_as(dp), len(x), out.ctypes.data_as(dp))
    want = np.power(x, y)
    ulp = np.abs(out - want) / np.spacing(want)
    print(f"portable pow vs libm: max {ulp.max():.2f} ulp, mean {ulp.mean():.3f} ulp, exact {np.mean(out == want):.3f}")
    assert ulp.max() <= 4.0 and ulp.mean() < 0.6
    # exact cases and monotone scaling
    one = np.array([1.0, 2.0, 4.0, 0.25]); yy = np.array([-0.55, -1.0, -0.5, -0.5]); o = np.empty(4)
    lib.ppow_many(one.ctypes.data_as(dp), yy.ctypes.data_as(dp), 4, o.ctypes.data_as(dp))
    assert o[0] == 1.0 and o[1] == 0.5 and o[2] == 0.5 and o[3] == 2.0


def test_ppow_oracle_build_tracks_the_default_oracle(oracle):
    """same restatement, psi through the portable pow: flows within 1e-6 of the default build on a reference golden
    (the difference is the last bits of psi through the lagged-nonlinearity iteration)"""
    g = golden("e2e_small")
    kw = params_of(g)
    a = oracle.get_displacement(g["fixed"], g["moving"], **kw)
    try:
        oracle.use_build("ppow")
        b = oracle.get_displacement(g["fixed"], g["moving"], **kw)
    finally:
        oracle.use_build("")
    d = np.linalg.norm(a - b, axis=-1)
    print(f"ppow oracle vs default oracle: mean {d.mean():.2e} max {d.max():.2e}")
    assert d.mean() < 1e-6 and d.max() < 1e-3
    assert not np.array_equal(a, b) or d.max() == 0.0
