#!/bin/bash
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/host_path.txt
import json, bench
from flowreg3d_amd import _lib
_lib.init(0)
for n in (8, 16):
    print(json.dumps(bench.host_path("cfg2", n, None)))
PY
python -m pytest tests/test_gpu_executor.py tests/test_gpu_e2e.py -q -m gpu 2>&1 | tail -3
