#!/bin/bash
# in-kernel split sums of the weight-gradient GEMM (tickets + spin: under timeout), then the training A/B and the input side with capped host threads
timeout 240 python -m pytest tests/test_hip_train_ops.py -q -x -k "gemm_tn" 2>&1 | tail -4 > gpurun_out/r05_t21_tests.log
if ! grep -q passed gpurun_out/r05_t21_tests.log; then echo "gemm_tn tests did not pass: stopping" >> gpurun_out/r05_t21_tests.log; exit 1; fi
for rep in 1 2; do for f in 0 1; do
  VITCAP_TRAIN_TN_SUM=$f timeout 300 python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train tn-sum=$f', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t21.log
done; done
timeout 300 python -m pytest tests/test_hip_train_e2e.py -q -x 2>&1 | tail -3 >> gpurun_out/r05_t21_tests.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8,10,8 python tools/input_side_bench.py 16384 gpurun_out/r05_input_side_capped.json 2>&1 | grep -E "num_workers|cgroup|steady" >> gpurun_out/r05_t21.log
