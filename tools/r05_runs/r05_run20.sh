#!/bin/bash
# attention backward on the LDS-DMA ring: parity tests, then both forms per launch and inside the training step; loader timers
python -m pytest tests/test_hip_train_ops.py tests/test_hip_ops.py -q -x -k "attn" 2>&1 | tail -4 > gpurun_out/r05_t20_tests.log
for f in 0 1 0 1; do VITCAP_ATTN_BWD_DMA=$f python tools/attn_bwd_bench.py >> gpurun_out/r05_t20.log 2>&1; done
for rep in 1 2; do for f in 0 1; do
  VITCAP_ATTN_BWD_DMA=$f python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train attn-bwd-dma=$f', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t20.log
done; done
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8,10 python tools/input_side_bench.py 16384 gpurun_out/r05_input_side_timers.json 2>&1 | grep -E "num_workers|cgroup" >> gpurun_out/r05_t20.log
