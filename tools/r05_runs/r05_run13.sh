#!/bin/bash
mkdir -p gpurun_out
for v in "VITCAP_LOADER_PREFETCH=4" "VITCAP_LOADER_CHUNK=4" "VITCAP_LOADER_CHUNK=16 VITCAP_LOADER_PREFETCH=4" "VITCAP_LOADER_PIN=0"; do
  echo "$v" >> gpurun_out/r05_input_side8.log
  env $v INPUT_SIDE_WORKERS=8 OMP_NUM_THREADS=4 python tools/input_side_bench.py 24576 2>&1 | grep -E "num_workers" >> gpurun_out/r05_input_side8.log
done
