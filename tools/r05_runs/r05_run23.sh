#!/bin/bash
# input side: deeper buffering against decode jitter (batches decoded ahead in the pool x batches queued on the device)
for cfg in "3 4" "6 12" "3 4" "10 24" "6 12" "10 24"; do
  set -- $cfg
  echo "ahead=$1 prefetch=$2" >> gpurun_out/r05_t23.log
  VITCAP_LOADER_AHEAD=$1 VITCAP_LOADER_PREFETCH=$2 INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8 python tools/input_side_bench.py 24576 2>&1 | grep -E "num_workers|steady" | cut -c1-420 >> gpurun_out/r05_t23.log
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t23.log
