#!/bin/bash
# input side: where the consumer thread's time goes, and the interpreter's thread switch interval (GIL hand-over) as an A/B
for sw in 0 1 0 1; do
  echo "switch-interval-short=$sw" >> gpurun_out/r05_t22.log
  INPUT_SIDE_SWITCH=$sw INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8 python tools/input_side_bench.py 16384 2>&1 | grep -E "num_workers|steady" >> gpurun_out/r05_t22.log
done
