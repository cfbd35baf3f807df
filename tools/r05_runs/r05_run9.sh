#!/bin/bash
mkdir -p gpurun_out
INPUT_SIDE_WORKERS=6,8,10 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 gpurun_out/r05_input_side_depth2.json 2>&1 | grep -E "num_workers" > gpurun_out/r05_input_side4.log
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench resident', d['value'], d['ms_per_step'])" >> gpurun_out/r05_input_side4.log
python -m pytest tests/test_hip_e2e.py -m gpu -q -x -k "predict_honours or image_tsv or pipeline" 2>&1 | tail -2 >> gpurun_out/r05_input_side4.log
