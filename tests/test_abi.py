"""CPU: the C-ABI shared library builds, loads and exports every symbol include/*.h declares; the
ctypes mirror agrees with the C struct layouts.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "flowreg3d_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from flowreg3d_amd import _lib
    return _lib


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fr3d_[a-z0-9_]+)\s*\(", txt)) - {"fr3d_progress_fn"})


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 20
    handle = lib.load()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"
        assert n in lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert sorted(lib.SIGNATURES) == names


def test_struct_layouts_match_ctypes(lib, tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "flowreg3d_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fr3d_params),'
                   'offsetof(fr3d_params,update_lag),offsetof(fr3d_params,eta),offsetof(fr3d_params,a_data),'
                   'offsetof(fr3d_params,solver_fp64),sizeof(fr3d_kernel_stat),offsetof(fr3d_kernel_stat,launches));'
                   'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    P, K = lib.Params, lib.KernelStat
    assert vals == [C.sizeof(P), P.update_lag.offset, P.eta.offset, P.a_data.offset, P.solver_fp64.offset,
                    C.sizeof(K), K.launches.offset]


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "flowreg3d_hip.h"\nint main(void){return FR3D_K_COUNT == 8 ? 0 : 1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(tmp_path / "t")])


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """Without a GPU the product path must raise, never fall back to the CPU."""
    import numpy as np
    import flowreg3d_amd
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    z = np.zeros((8, 8, 8), np.float32)
    with pytest.raises(RuntimeError):
        flowreg3d_amd.get_displacement(z, z)
    with pytest.raises(RuntimeError):
        flowreg3d_amd.imregister_wrapper(z, z, z, z, z)
    from flowreg3d_amd.executor import HipExecutor3D
    assert HipExecutor3D.register() is False  # declines -> reference pipeline falls back


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flowreg3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"
