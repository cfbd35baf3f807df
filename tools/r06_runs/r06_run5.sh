#!/bin/bash
# round 6, call 5: device JPEG back half -- bit-exactness tests, the TSV pipeline tests, and the input side with / without it (interleaved)
python -m pytest tests/test_hip_jpeg.py tests/test_hip_image_transform.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r06_t5_tests.log
python -m pytest tests/test_hip_e2e.py tests/test_pipeline_surface.py -m gpu -q -x -k "tsv or pipeline or predict" 2>&1 | tail -5 >> gpurun_out/r06_t5_tests.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,8,8,8,6,6 INPUT_SIDE_DEVICE_JPEG=1,0,1,0,1,0 python tools/input_side_bench.py 24576 gpurun_out/r06_input_side.json > gpurun_out/r06_input_side.log 2>&1
