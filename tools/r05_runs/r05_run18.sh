#!/bin/bash
# which training extra costs on the 4-wave kernel: bit 0 zout, bit 1 aux, bit 2 colsum
for rep in 1 2; do
for m in 0 1 2 4 3 7; do
  VITCAP_GEMM_4W_EXTRAS=$m python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train extras-mask=$m', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t18.log
done; done
