#!/bin/bash
mkdir -p gpurun_out
INPUT_SIDE_WORKERS=6,8,10,12 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 gpurun_out/r05_input_side_blocking.json 2>&1 | grep -E "num_workers" > gpurun_out/r05_input_side3.log
echo "pinned workers (VITCAP_LOADER_CPUS=8:2)" >> gpurun_out/r05_input_side3.log
VITCAP_LOADER_CPUS=8:2 INPUT_SIDE_WORKERS=8,10 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 2>&1 | grep -E "num_workers" >> gpurun_out/r05_input_side3.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])" >> gpurun_out/r05_input_side3.log
python bench.py --mode train --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['frac'], d['roofline'] and d['roofline'].get('note'))" >> gpurun_out/r05_input_side3.log
