#!/bin/bash
set -o pipefail
O=gpurun_out/r03p; mkdir -p $O
FR3D_LIB=$PWD/flowreg3d_amd/lib/libflowreg3d_hip_exp.so python tools/experiments/cfg5_decompose.py 0.5 2>&1 | tee $O/cfg5_decompose_half_exact_tail.jsonl
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_stages.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_verify_mode.py tests/test_gpu_reference_executor.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -15
python -m pytest tests/test_gpu_fullsize_parity.py -q -m gpu 2>&1 | tail -15
