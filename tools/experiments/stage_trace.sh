#!/bin/bash
# Per-kernel durations of the non-solver stages (rocprofv3 kernel trace of one cfg2 bench batch), grouped by
# kernel and grid: run on the GPU box from the repo root; writes gpurun_out/r02/kt_now.csv and prints the table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/_kt -- python3 bench.py --workload ${1:-cfg2} --steps 8 --warmup 0 --no-cpu-baseline --no-extras --condition 2 > gpurun_out/r02/kt.log 2>&1 || { tail -5 gpurun_out/r02/kt.log; exit 1; }
cp $(find gpurun_out/r02/_kt -name "*kernel_trace.csv" | head -1) gpurun_out/r02/kt_now.csv
rm -rf gpurun_out/r02/_kt
python3 tools/experiments/stage_trace.py gpurun_out/r02/kt_now.csv
