#!/bin/bash
# (1) the 4-wave extras test through tile_hint 42; (2) input side: what the predict loop sustains when JPEG decoding costs nothing
python -m pytest tests/test_hip_train_ops.py -q -x -k "extras or colsum" 2>&1 | tail -3 > gpurun_out/r05_t19_tests.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,8 python tools/input_side_bench.py 16384 gpurun_out/r05_input_side_ceiling.json > gpurun_out/r05_t19.log 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r05_t19.log
