#!/bin/bash
# round 3, first GPU call: correctness of the split / chained sweep and the packed storage, then tile-shape sweeps
set -e -o pipefail
O=gpurun_out/r03a; mkdir -p $O
export FR3D_LIB=$PWD/flowreg3d_amd/lib/libflowreg3d_hip_exp.so
python tools/experiments/sor_env_probe.py 256 8 FR3D_SOR_SHAPE 2x1,2x2,2x4,4x2,4x4,1x4,1x8,8x2,4x1 3 > $O/shape_256_m1.jsonl
echo 256 done
FR3D_PROBE_MODE=3 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x4,4x2,2x2 2 > $O/shape_512_m3.jsonl
echo 512 m3 done
FR3D_PROBE_MODE=2 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x4 2 > $O/shape_512_m2.jsonl
echo 512 m2 done
FR3D_PROBE_MODE=1 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x4 2 > $O/shape_512_m1.jsonl
echo 512 m1 done
unset FR3D_LIB
python tools/experiments/mode_parity_probe.py cfg3 3,2 > $O/parity_cfg3.jsonl
cat $O/parity_cfg3.jsonl
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_stages.py -q -m gpu > $O/pytest.log 2>&1 || true
tail -3 $O/pytest.log
