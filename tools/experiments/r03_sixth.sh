#!/bin/bash
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_gpu_verify_mode.py -q -m gpu -x > $O/pytest_verify.log 2>&1; tail -15 $O/pytest_verify.log
python -m pytest tests -q -m gpu --deselect tests/test_gpu_verify_mode.py > $O/pytest_all.log 2>&1; tail -8 $O/pytest_all.log
grep -h "EPE vs oracle" $O/pytest_all.log | head
