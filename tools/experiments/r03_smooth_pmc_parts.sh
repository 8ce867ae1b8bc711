#!/bin/bash
# psi_smooth solver: HBM traffic of the parts of a step (experiment build, FR3D_SM_DBG leaves kinds of workgroups out):
#   12 = interior tiles only (P + sweep), 14 = P tiles only, 13 = sweep tiles only, 3 = surface workgroups only
# usage (GPU box, repo root): tools/experiments/r03_smooth_pmc_parts.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_smooth; mkdir -p $out
export FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so
for dbg in 12 14 13 3; do
  export FR3D_SM_DBG=$dbg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/_p -- python3 bench.py --workload cfg2 --steps 1 --warmup 0 --batch 1 --a-smooth 0.5 --lanes 1 --no-cpu-baseline --no-extras > $out/pmc_parts.log 2>&1 || { tail -5 $out/pmc_parts.log; exit 1; }
    python3 tools/pmc_summary.py $out/_p | grep smooth | sed "s/^/dbg=$dbg /" >> $out/pmc_parts.txt; rm -rf $out/_p
  done
done
cut -c1-170 $out/pmc_parts.txt
