#!/bin/bash
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_form"][:40])'
for i in 1 2; do for v in 0 1; do
  echo "B=64 pipeline, VITCAP_GEMM_4W_MIX=$v" >> gpurun_out/r05_mix.txt
  VITCAP_GEMM_4W_MIX=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_mix.txt
done; done
for v in 0 1; do
  echo "beam 5 x 256, VITCAP_GEMM_4W_MIX=$v" >> gpurun_out/r05_mix.txt
  VITCAP_GEMM_4W_MIX=$v python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_mix.txt
  echo "B=512, VITCAP_GEMM_4W_MIX=$v" >> gpurun_out/r05_mix.txt
  VITCAP_GEMM_4W_MIX=$v python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_mix.txt
done
for v in base ares wres awres pf pfall pf4; do for m in 295424 36928; do
  echo "variant $v"; tools/probes/_bin/g4w_probe_$v $m 2 | cut -c1-250
done; done > gpurun_out/r05_g4w_probe2.txt 2>&1
python tools/encode_only_bench.py 64 60 2>/dev/null | grep "B=" > gpurun_out/r05_encode_only.txt
python tools/encode_only_bench.py 512 10 2>/dev/null | grep "B=" >> gpurun_out/r05_encode_only.txt
python tools/power_during.py 9 -- python bench.py --steps 600 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | grep -a power_during > gpurun_out/r05_power_bench.txt
python tools/power_during.py 12 -- python bench.py --steps 80 --warmup 3 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | grep -a power_during > gpurun_out/r05_power_bench_b512.txt
INPUT_SIDE_WORKERS=6,8,10 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 gpurun_out/r05_input_side_pinned.json 2>&1 | grep -E "num_workers|page-locked|Register" > gpurun_out/r05_input_side2.log
VITCAP_LOADER_PIN=0 INPUT_SIDE_WORKERS=6 OMP_NUM_THREADS=4 python tools/input_side_bench.py 6144 2>&1 | grep -E "num_workers" >> gpurun_out/r05_input_side2.log
