#!/bin/bash
# psi_smooth solver path (a_smooth = 0.5): fused P-stage/sweep workgroups against the split form, same box
#   tools/experiments/smooth_ab.sh <tag> ["forms"]   split = two kernels per step (shipped), one = both stages in one kernel
#   (rounds 2-3), fused / paired = P-stage and sweep tiles sharing a plane; all but split need the experiment build
tag=${1:-smooth}; out=gpurun_out/$tag; mkdir -p "$out"
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fuzz.py tests/test_gpu_executor.py -x -q -m gpu > "$out/tests.log" 2>&1 || { tail -20 "$out/tests.log"; exit 1; }
tail -1 "$out/tests.log"
for form in ${2:-split one paired split one}; do
  FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so FR3D_SMOOTH=$form timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --a-smooth 0.5 --steps 8 --condition 10 \
     > "$out/bench_$form.json" 2> "$out/err_$form.log" || { tail -5 "$out/err_$form.log"; exit 1; }
  python - "$out/bench_$form.json" $form <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],"value",round(d["value"],3),"one_lane",round(d["one_lane"]["value"],3),"sor ms/volume",round(d["roofline"]["sor_ms_per_step"],1),"frac",round(d["roofline"]["frac"],4))
PY
done
