#!/bin/bash
# hardware-queue aliasing?  the same sequence of predict() calls in one process with 4 (default) / 8 / 16 HIP hardware queues, and the headline bench
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> gpurun_out/r05_t30.log
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t30.log
  GPU_MAX_HW_QUEUES=$q INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,10,8,10 python tools/input_side_bench.py 24576 2>&1 | grep -E "ceiling|num_workers" | cut -c1-200 >> gpurun_out/r05_t30.log
done
