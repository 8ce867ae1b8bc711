#!/bin/bash
# samples rocm-smi clocks/power while bench.py runs: tools/clock_watch.sh [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --no-cpu-baseline "$@" > gpurun_out/cw_bench.json 2> gpurun_out/cw_bench.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|fclk|mclk|Power|Temperature \(Sensor (junction|memory)" | sed -E 's/\s+/ /g' | tr '\n' ';'
  echo
  sleep 2
done
wait $pid
python -c "import json; d=json.loads(open('gpurun_out/cw_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'])"
