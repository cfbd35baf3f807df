#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench resident', d['value'], d['ms_per_step'])" > gpurun_out/r05_input_side7.log
INPUT_SIDE_WORKERS=8,9 OMP_NUM_THREADS=4 python tools/input_side_bench.py 24576 gpurun_out/r05_input_side_long2.json 2>&1 | grep -E "num_workers|cgroup" >> gpurun_out/r05_input_side7.log
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench resident', d['value'], d['ms_per_step'])" >> gpurun_out/r05_input_side7.log
