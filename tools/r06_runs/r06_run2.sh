#!/bin/bash
# round 6, call 2: new trainer tests (NaN watch, graph + exchange), which limiter holds the clock, the bench line with the whole-caption cpu_baseline
python -m pytest tests/test_hip_train_e2e.py -m gpu -q -x -k "nan_watch or rccl_exchange or graph_step" 2>&1 | tail -8 > gpurun_out/r06_t2_tests.log
python tools/throttle_probe.py 5 > gpurun_out/r06_throttle_probe.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
