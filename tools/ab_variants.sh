#!/bin/bash
# runs bench.py (no CPU baseline) for a list of "ENV=VAL,ENV=VAL" settings; prints value + sor roofline
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  envs=$(echo "$cfg" | tr ',' ' ')
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1)
  echo "$cfg :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.3f  ms/step %.1f  sor achieved %.0f GB/s" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"]))' 2>&1 | tail -1)"
done
