#!/bin/bash
# round 5, third GPU call: changed tests, power during the bench, 4-wave forms inside the pipeline with 2 encoder parts, input side
mkdir -p gpurun_out
python -m pytest tests/test_hip_e2e.py tests/test_hip_cbs.py tests/test_hip_train_e2e.py -m gpu -q -s -x -k "predict_honours or cbs_captions or trajectory or beam5_batch256 or train_step_batch64 or scst_step_at_config or image_tsv or trains_on_tsv" 2>&1 | grep -E "vs oracle|oracle|case|passed|failed|Error|error|assert" | tail -40 > gpurun_out/r05_t3.log
python tools/power_during.py 25 -- python bench.py --steps 300 --warmup 5 --no-cpu-baseline --isolated 0 > gpurun_out/r05_power_bench.txt 2>/dev/null
python tools/power_during.py 25 -- python bench.py --steps 40 --warmup 3 --batch 512 --no-cpu-baseline --isolated 0 > gpurun_out/r05_power_bench_b512.txt 2>/dev/null
for i in 1 2; do for v in "-1,2" "1,2" "0,2"; do
  echo "B=64 pipeline, VITCAP_GEMM_4W=$v" >> gpurun_out/r05_forms64.txt
  VITCAP_GEMM_4W=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r05_forms64.txt
done; done
for n in 2304 3072 768; do
  echo "B=64 pipeline, 4-wave one-tile for N=$n only" >> gpurun_out/r05_forms64.txt
  VITCAP_GEMM_4W_TILES_N=$n,1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r05_forms64.txt
done
OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 gpurun_out/r05_input_side.json > gpurun_out/r05_input_side.log 2>&1
