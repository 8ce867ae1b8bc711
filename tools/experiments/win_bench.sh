#!/bin/bash
# window sweep against plane sweep on the GPU box: the window tests, then bench.py (no extras) per solver mode
#   tools/experiments/win_bench.sh <tag> [modes="2 1 3"] [sweeps="window planes"]
tag=${1:-win}; modes=${2:-"2 1 3"}; sweeps=${3:-"window"}
out=gpurun_out/$tag; mkdir -p "$out"
timeout -k 10 300 python -m pytest tests/test_gpu_sor_window.py -x -q -m gpu > "$out/wintest.log" 2>&1 || { tail -20 "$out/wintest.log"; exit 1; }
tail -1 "$out/wintest.log"
for sw in $sweeps; do for m in $modes; do
  FR3D_SWEEP=$sw timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --solver-fp64 $m --steps 4 \
     > "$out/bench_${sw}_m$m.json" 2> "$out/err_${sw}_m$m.log" || { tail -5 "$out/err_${sw}_m$m.log"; exit 1; }
  python - "$out/bench_${sw}_m$m.json" $sw $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],"mode",sys.argv[3],"value",d["value"],"sor ms/step",d["roofline"].get("sor_ms_per_step"))
PY
done; done
