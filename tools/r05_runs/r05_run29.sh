#!/bin/bash
# the loader thread on its own stream: pipeline tests, then the configuration that showed the 22 ms-per-batch mode (ceiling arms first)
timeout 900 python -m pytest tests/test_hip_e2e.py tests/test_hip_pipeline.py -q -x -m gpu -k "predict or pipeline or loader or tsv" 2>&1 | tail -3 > gpurun_out/r05_t29_tests.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t29.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,10,8,10 python tools/input_side_bench.py 24576 gpurun_out/r05_input_side_own_stream.json 2>&1 | grep -E "ceiling|num_workers" | cut -c1-400 >> gpurun_out/r05_t29.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t29.log
