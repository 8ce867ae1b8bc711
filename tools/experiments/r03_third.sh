#!/bin/bash
# round 3: same-box A/B of the round-2 library against the fused, phase-ordered sweep kernel
set -e -o pipefail
O=gpurun_out/r03c; mkdir -p $O
L=flowreg3d_amd/lib
python tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=2x2 $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=1x1 > $O/ab_256_m1.jsonl
echo 256 done
python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=2x2 > $O/ab_512_m1.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=2x2 > $O/ab_512_m2.jsonl
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=2x2 > $O/ab_512_m3.jsonl
cat $O/*.jsonl | cut -c1-230
