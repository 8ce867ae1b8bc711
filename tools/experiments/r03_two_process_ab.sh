#!/bin/bash
# 8 volumes of cfg2 on ONE GPU three ways, alternated: one process / one lane; one process / two lanes (FR3D_LANES=2);
# two processes (gloo ranks sharing the device, 4 volumes each).  usage (GPU box, repo root): tools/experiments/r03_two_process_ab.sh
out=gpurun_out/r03_lanes; mkdir -p $out; : > $out/two_process_ab.txt
val() { python3 -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$2 value %.3f volumes/s' % d['value'])" >> $out/two_process_ab.txt; }
for rep in 0 1; do
  FR3D_LANES=1 timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --condition 15 > $out/_a.json 2>/dev/null || exit 1; val $out/_a.json "rep $rep one process, one lane: "
  FR3D_LANES=2 timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --condition 15 > $out/_b.json 2>/dev/null || exit 1; val $out/_b.json "rep $rep one process, two lanes:"
  FR3D_DIST_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2954$rep bench.py --gpus 2 --steps 4 --warmup 1 --condition 15 > $out/_c.json 2>/dev/null || exit 1; val $out/_c.json "rep $rep two processes x 4:      "
done
cat $out/two_process_ab.txt
