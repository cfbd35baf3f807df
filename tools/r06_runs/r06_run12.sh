#!/bin/bash
# round 6, call 12: jpeg_backhalf with merged host -> device copies (one per worker task) and one output allocation: tests + input side A/B
python -m pytest tests/test_hip_jpeg.py tests/test_hip_image_transform.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r06_t12_tests.log
python -m pytest tests/test_hip_e2e.py tests/test_pipeline_surface.py tests/test_hip_train_e2e.py -m gpu -q -x -k "tsv or pipeline or predict" 2>&1 | tail -3 >> gpurun_out/r06_t12_tests.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8,8,8,8,6,6 INPUT_SIDE_DEVICE_JPEG=1,0,1,0,1,0 python tools/input_side_bench.py 24576 gpurun_out/r06_input_side_b.json > gpurun_out/r06_input_side_b.log 2>&1
