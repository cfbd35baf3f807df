#!/bin/bash
mkdir -p gpurun_out
INPUT_SIDE_WORKERS=6,8,12 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 gpurun_out/r05_input_side_cpu.json 2>&1 | grep -E "num_workers|cgroup" > gpurun_out/r05_input_side5.log
