#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do for v in 0 1; do
VITCAP_GEMM_4W_EXTRAS=$v python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train extras-on-4w=$v', d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['frac'])" >> gpurun_out/r05_t17.log
done; done
