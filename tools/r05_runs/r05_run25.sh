#!/bin/bash
# input side: workers x buffering, 24 576 images, one box; the resident bench before and after
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t25.log
for rep in 1 2; do for cfg in "8 3 4" "8 10 24" "10 10 24" "12 12 24" "10 6 16"; do
  set -- $cfg
  echo "workers=$1 ahead=$2 prefetch=$3" >> gpurun_out/r05_t25.log
  VITCAP_LOADER_AHEAD=$2 VITCAP_LOADER_PREFETCH=$3 INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=$1 python tools/input_side_bench.py 24576 2>&1 | grep -E "num_workers|steady" | cut -c1-420 >> gpurun_out/r05_t25.log
done; done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t25.log
