#!/bin/bash
# a_smooth = 1 sweep: workgroup ids that share an XCD take one contiguous eighth of a launch's tile list
# (variant build: make variant VARIANT=xcd VFLAGS=-DFR3D_SOR_XCD=1) against the shipped order, interleaved on one box
# usage (GPU box, repo root): tools/experiments/r03_sor_xcd_ab.sh
out=gpurun_out/r03_xcd; mkdir -p $out
L=flowreg3d_amd/lib
FR3D_PROBE_MODE=1 timeout -k 10 400 python3 tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip_exp.so $L/libflowreg3d_hip_xcd.so > $out/ab_256_m1.jsonl || exit 1
FR3D_PROBE_MODE=3 timeout -k 10 400 python3 tools/experiments/lib_ab_probe.py 256 8 2 $L/libflowreg3d_hip_exp.so $L/libflowreg3d_hip_xcd.so > $out/ab_256_m3.jsonl || exit 1
FR3D_PROBE_MODE=3 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 512 4 2 $L/libflowreg3d_hip_exp.so $L/libflowreg3d_hip_xcd.so > $out/ab_512_m3.jsonl || exit 1
cat $out/ab_*.jsonl | cut -c1-250
