#!/bin/bash
# round 3: restored r02 kernel body inside the chain-schedule / packed-storage framework: A/B + parity + GPU suite
set -e -o pipefail
O=gpurun_out/r03d; mkdir -p $O
L=flowreg3d_amd/lib
python tools/experiments/lib_ab_probe.py 256 8 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so > $O/ab_256_m1.jsonl
python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so > $O/ab_512_m1.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip_r02.so $L/libflowreg3d_hip.so > $O/ab_512_m2.jsonl
FR3D_PROBE_MODE=3 python tools/experiments/lib_ab_probe.py 512 4 1 $L/libflowreg3d_hip.so $L/libflowreg3d_hip_exp.so@FR3D_SOR_SHAPE=4x1 > $O/ab_512_m3.jsonl
cat $O/*.jsonl | cut -c1-200
python tools/experiments/mode_parity_probe.py cfg3 3 > $O/parity_cfg3.jsonl
cat $O/parity_cfg3.jsonl
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize_parity.py > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
