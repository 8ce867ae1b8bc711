#!/bin/bash
# a_smooth = 1 sweep: within every run of 8 G tiles an XCD takes G consecutive ones (experiment build, FR3D_SOR_XCD_G;
# 0 = blockIdx order), alternated on one box.  usage (GPU box, repo root): tools/experiments/r03_sor_xcd_group_ab.sh
out=gpurun_out/r03_xcd; mkdir -p $out
E=flowreg3d_amd/lib/libflowreg3d_hip_exp.so
FR3D_PROBE_MODE=3 timeout -k 10 900 python3 tools/experiments/lib_ab_probe.py 512 4 2 $E@FR3D_SOR_XCD_G=0 $E@FR3D_SOR_XCD_G=8 $E@FR3D_SOR_XCD_G=16 $E@FR3D_SOR_XCD_G=32 $E@FR3D_SOR_XCD_G=64 > $out/g_512_m3.jsonl || exit 1
FR3D_PROBE_MODE=3 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 256 8 2 $E@FR3D_SOR_XCD_G=0 $E@FR3D_SOR_XCD_G=8 $E@FR3D_SOR_XCD_G=16 $E@FR3D_SOR_XCD_G=32 $E@FR3D_SOR_XCD_G=64 > $out/g_256_m3.jsonl || exit 1
FR3D_PROBE_MODE=1 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 256 8 2 $E@FR3D_SOR_XCD_G=0 $E@FR3D_SOR_XCD_G=16 $E@FR3D_SOR_XCD_G=32 > $out/g_256_m1.jsonl || exit 1
FR3D_PROBE_MODE=1 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 512 4 1 $E@FR3D_SOR_XCD_G=0 $E@FR3D_SOR_XCD_G=16 > $out/g_512_m1.jsonl || exit 1
FR3D_PROBE_MODE=2 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 512 4 1 $E@FR3D_SOR_XCD_G=0 $E@FR3D_SOR_XCD_G=16 > $out/g_512_m2.jsonl || exit 1
echo finished
