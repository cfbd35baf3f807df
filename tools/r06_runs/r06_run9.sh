#!/bin/bash
# round 6, call 9: the whole GPU suite + smoke + the driver's bench call on the final tree
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
python bench.py --steps 30 --warmup 3 --mode train > gpurun_out/r06_final_bench_train.json 2>> gpurun_out/r06_final_bench.err
python tools/exact_rate.py 2>&1 | grep EXACT > gpurun_out/r06_final_exact_rate.txt
