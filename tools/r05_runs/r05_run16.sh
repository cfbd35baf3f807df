#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_ops.py tests/test_hip_train_e2e.py -m gpu -q -x -k "extras or fused_bias or test_losses or gradients_per_tensor or parameter_update or graph_step or batch64" 2>&1 | tail -3 > gpurun_out/r05_t16.log
for i in 1 2; do
python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['frac'])" >> gpurun_out/r05_t16.log
done
