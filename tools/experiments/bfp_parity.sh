#!/bin/bash
# numerics what-if (experiment build): block floating-point rounding of the frozen system (M: 6 values + b: 3 values, one
# exponent per group) and of the increments (3 values, one exponent) on top of the packed 42-bit mode; parity against the
# committed full-size CPU samples.   tools/experiments/bfp_parity.sh <tag> "<cases>" "<dbg values>"
tag=${1:-bfp}; cases=${2:-"cfg2_recipe"}; dbgs=${3:-"0"}
out=gpurun_out/$tag; mkdir -p $out
for c in $cases; do for d in $dbgs; do
  echo "case $c FR3D_SOR_DBG=$d (bits $((d>>8)), flags $((d&255)))" | tee -a $out/log.txt
  FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so FR3D_SOR_DBG=$d timeout -k 10 500 python tools/experiments/mode_parity_probe.py $c 3 2>&1 | tail -2 | tee -a $out/log.txt
done; done
