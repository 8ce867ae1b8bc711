#!/bin/bash
set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
python - > $O/cfg2_recipe_modes.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tests")
import test_gpu_fullsize_parity as t
from flowreg3d_amd import _lib
_lib.init(0)
for m in (3, 2, 0):
    e, msg = t._measure("cfg2_recipe", solver_fp64=m)
PY
cat $O/cfg2_recipe_modes.txt | cut -c1-250
L=flowreg3d_amd/lib
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 256 8 1 $L/libflowreg3d_hip.so > $O/ab_256_m3.jsonl
FR3D_PROBE_MODE=1 python tools/experiments/lib_ab_probe.py 256 8 1 $L/libflowreg3d_hip.so > $O/ab_256_m1.jsonl
FR3D_PROBE_MODE=0 python tools/experiments/lib_ab_probe.py 256 8 1 $L/libflowreg3d_hip.so > $O/ab_256_m0.jsonl
cat $O/ab_256_*.jsonl | cut -c1-220
