#!/bin/bash
# round 3: clean A/B after the SorEntry scalar-load fix: r02 | new (2x1) | chains | phase-ordered kernel
set -e -o pipefail
O=gpurun_out/r03e; mkdir -p $O
L=flowreg3d_amd/lib
E=$L/libflowreg3d_hip_exp.so; P=$L/libflowreg3d_hip_phased.so
python tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $E@FR3D_SOR_SHAPE=2x2 $E@FR3D_SOR_SHAPE=2x4 $E@FR3D_SOR_SHAPE=4x1 $P $P@FR3D_SOR_SHAPE=2x4 > $O/ab_256_m1.jsonl
echo 256 done
python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $E@FR3D_SOR_SHAPE=2x2 $E@FR3D_SOR_SHAPE=2x4 $P $P@FR3D_SOR_SHAPE=2x4 > $O/ab_512_m1.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $E@FR3D_SOR_SHAPE=2x4 $P $P@FR3D_SOR_SHAPE=2x4 > $O/ab_512_m2.jsonl
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $E@FR3D_SOR_SHAPE=2x2 $E@FR3D_SOR_SHAPE=2x4 $P $P@FR3D_SOR_SHAPE=2x4 > $O/ab_512_m3.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e/*.jsonl")):
    for l in open(f):
        j=json.loads(l); print(j["edge"], j["mode"], j["lib"][:60], j.get("sor_ms_per_vol"), j.get("frac"), j.get("flow_sha"), j.get("error","")[:80])
PY
