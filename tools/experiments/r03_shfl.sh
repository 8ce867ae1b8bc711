#!/bin/bash
set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
L=flowreg3d_amd/lib
python tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shfl.so > $O/ab_256_m1.jsonl
python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shfl.so > $O/ab_512_m1.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_shfl.so > $O/ab_512_m2.jsonl
cat $O/*.jsonl | cut -c1-200
