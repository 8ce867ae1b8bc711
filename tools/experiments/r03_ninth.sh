#!/bin/bash
set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1; tail -12 $O/pytest_all.log
python - > $O/cfg5_mode3.txt 2>&1 <<'PY'
import sys, time; sys.path.insert(0, "tests")
import test_gpu_fullsize_parity as t
from flowreg3d_amd import _lib
lib = _lib.init(0)
for case in ("cfg5", "cfg5_levels8"):
    for m in (3, 2):
        lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
        e, msg = t._measure(case, solver_fp64=m)
        s = _lib.prof_get()["sor"]
        print(f"   {case} mode {m}: sor {s['ms']:.0f} ms, frac(own basis) {s['algo_bytes']/s['ms']/8e9:.3f}")
PY
cut -c1-260 $O/cfg5_mode3.txt
