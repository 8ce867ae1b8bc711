#!/bin/bash
# what bounds the sweep: the same loads and stores without the relaxation arithmetic (16) / without the pow (32)
set -o pipefail
O=gpurun_out/r03l; mkdir -p $O
export FR3D_LIB=$PWD/flowreg3d_amd/lib/libflowreg3d_hip_exp.so
for m in 1 3; do
  FR3D_PROBE_MODE=$m python tools/experiments/sor_env_probe.py 256 8 FR3D_SOR_DBG 0,16,32,48 2 > $O/skel_256_m$m.jsonl
  FR3D_PROBE_MODE=$m python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_DBG 0,16,32,48 1 > $O/skel_512_m$m.jsonl
done
grep -h sor_ms $O/skel_*.jsonl | cut -c1-170
