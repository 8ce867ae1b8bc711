#!/bin/bash
set -o pipefail
O=gpurun_out/r03n; mkdir -p $O
L=flowreg3d_amd/lib
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shfl.so $L/libflowreg3d_hip_shflu.so > $O/ab_512_m3.jsonl
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shfl.so $L/libflowreg3d_hip_shflu.so > $O/ab_256_m3.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shflu.so $L/libflowreg3d_hip_shflp.so > $O/ab_512_m2.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03n/*.jsonl")):
    for l in open(f):
        j=json.loads(l); print(j["edge"], j["mode"], j["lib"][:40], j.get("sor_ms_per_vol"), j.get("frac"), j.get("flow_sha"), j.get("error","")[:100])
PY
