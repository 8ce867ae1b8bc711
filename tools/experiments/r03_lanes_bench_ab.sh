#!/bin/bash
# bench.py's headline region with one engine lane and with two (FR3D_LANES), alternated on one box
# usage (GPU box, repo root): tools/experiments/r03_lanes_bench_ab.sh
out=gpurun_out/r03_lanes; mkdir -p $out; : > $out/bench_ab.txt
for rep in 0 1; do for lanes in 1 2; do
  FR3D_LANES=$lanes timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --condition 20 > $out/_b.json 2> $out/_b.err || { tail -3 $out/_b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$out/_b.json'));print('lanes $lanes rep $rep value %.3f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $out/bench_ab.txt
done; done
cat $out/bench_ab.txt
