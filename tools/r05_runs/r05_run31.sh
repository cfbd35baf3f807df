#!/bin/bash
# process-wide role streams: the sequence of predict() calls that alternated between 3 750 and 1 935 images/s, default hardware-queue count; e2e tests
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t31.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,10,8,10,8,10 python tools/input_side_bench.py 24576 gpurun_out/r05_input_side_role_streams.json 2>&1 | grep -E "ceiling|num_workers" | cut -c1-330 >> gpurun_out/r05_t31.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t31.log
python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t31.log
python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t31.log
timeout 1500 python -m pytest tests/test_hip_e2e.py -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r05_t31_tests.log
