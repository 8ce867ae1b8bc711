#!/bin/bash
# quick PMC pass for the SOR kernels: tools/pmc_quick.sh "<counters>" [bench args...]   (env selects kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ctr=$1; shift
out=gpurun_out/pmcq; rm -rf $out; mkdir -p $out
timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 0 --batch 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
python3 tools/pmc_summary.py $out k_sor | cut -c1-160
python3 tools/pmc_summary.py $out k_axpy | cut -c1-160
rm -rf $out
