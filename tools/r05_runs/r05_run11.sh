#!/bin/bash
mkdir -p gpurun_out
INPUT_SIDE_WORKERS=8,10,12 OMP_NUM_THREADS=4 python tools/input_side_bench.py 24576 gpurun_out/r05_input_side_long.json 2>&1 | grep -E "num_workers|cgroup" > gpurun_out/r05_input_side6.log
