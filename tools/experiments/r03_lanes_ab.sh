#!/bin/bash
# one engine lane against two (FR3D_LANES=2: lock-step batches of half the size alternate between two streams), the
# shipped library, alternated on one box.  usage (GPU box, repo root): tools/experiments/r03_lanes_ab.sh
out=gpurun_out/r03_lanes; mkdir -p $out
L=flowreg3d_amd/lib/libflowreg3d_hip.so
FR3D_PROBE_MODE=3 timeout -k 10 400 python3 tools/experiments/lib_ab_probe.py 256 8 2 $L $L@FR3D_LANES=2 > $out/ab_256_m3.jsonl || exit 1
FR3D_PROBE_MODE=1 timeout -k 10 400 python3 tools/experiments/lib_ab_probe.py 256 8 2 $L $L@FR3D_LANES=2 > $out/ab_256_m1.jsonl || exit 1
FR3D_PROBE_MODE=3 timeout -k 10 600 python3 tools/experiments/lib_ab_probe.py 512 4 2 $L $L@FR3D_LANES=2 > $out/ab_512_m3.jsonl || exit 1
cat $out/ab_*.jsonl | cut -c1-260
