#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_measure.sh <tag> [workload] [pmc:0|1]
# runs the parity tests' fast subset, the bench, and (optionally) the two PMC passes; writes
# gpurun_out/<tag>_*.  Never combines --pmc with tracing domains other than --kernel-trace.
tag=$1; wl=${2:-cfg2}; pmc=${3:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py --workload $wl --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1 || { tail -5 gpurun_out/${tag}_bench.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*\|"kernel_ms_per_step": {[^}]*}' gpurun_out/${tag}_bench.log
if [ "$pmc" = "1" ]; then
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/_pf -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 &&
  timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/_pw -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/_pf > gpurun_out/${tag}_pmc_fetch.txt
  python3 tools/pmc_summary.py gpurun_out/_pw > gpurun_out/${tag}_pmc_write.txt
  grep -h -E "sor|tensor|axpy" gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt | cut -c1-160
  rm -rf gpurun_out/_pf gpurun_out/_pw
fi
