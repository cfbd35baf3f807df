#!/bin/bash
# GPU round: constrained beam search tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_cbs.py tests/test_cabi.py -q -m gpu -s > gpurun_out/tests_cbs_r04.log 2>&1
echo "tests rc=$?" >> gpurun_out/tests_cbs_r04.log
