#!/bin/bash
set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
export FR3D_LIB=$PWD/flowreg3d_amd/lib/libflowreg3d_hip_exp.so
FR3D_PROBE_MODE=3 python tools/experiments/sor_env_probe.py 256 8 FR3D_SOR_SHAPE 2x1,2x2,4x1,4x2,1x2,1x4,2x3 3 > $O/shape_256_m3.jsonl
FR3D_PROBE_MODE=3 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x2,4x1,4x2,1x2,1x4,2x3 2 > $O/shape_512_m3.jsonl
python - <<'PY'
import json,collections
for f in ("shape_256_m3","shape_512_m3"):
    d=collections.defaultdict(list)
    for l in open(f"gpurun_out/r03h/{f}.jsonl"):
        j=json.loads(l)
        if "frac" in j: d[j["FR3D_SOR_SHAPE"]].append((j["sor_ms_per_vol"],j["frac"]))
        elif not j.get("bit_identical_to_first",True): print("NOT IDENTICAL", j)
    print(f, {k:v for k,v in d.items()})
PY
