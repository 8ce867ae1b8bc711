#!/bin/bash
# psi_smooth solver (a_smooth = 0.5): what a step's four kinds of workgroups cost, and the path's real HBM traffic.
# usage (GPU box, repo root): tools/experiments/r03_smooth_decompose.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_smooth; mkdir -p $out
export FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so FR3D_PROBE_ASMOOTH=0.5 FR3D_PROBE_MODE=1
timeout -k 10 400 python3 tools/experiments/sor_env_probe.py 256 8 FR3D_SM_DBG 0,1,2,4,8,5,10,16 2 > $out/decompose_256.jsonl 2> $out/decompose_256.err || { tail -3 $out/decompose_256.err; exit 1; }
unset FR3D_LIB FR3D_PROBE_ASMOOTH FR3D_PROBE_MODE
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/_p -- python3 bench.py --workload cfg2 --steps 1 --warmup 0 --batch 1 --a-smooth 0.5 --lanes 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/_p > $out/pmc_${c}_cfg2_asmooth05.txt; rm -rf $out/_p
done
grep -h "smooth\|axpy" $out/pmc_*.txt | cut -c1-160
grep -v bit_identical $out/decompose_256.jsonl
