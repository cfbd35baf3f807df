#!/bin/bash
# round 6, call 3: the shipped A-panel prefetch (persistent 4-wave bf16-output GEMMs from 64k rows on) at B = 512, interleaved A/B on one box
O=gpurun_out/r06_pf_ab.txt
: > $O
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -3 >> $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'W', e.get('board_power_w_median'), 'MHz', e.get('sclk_mhz_median'), 'J/step', e.get('joules_per_step'))"; }
for rep in 1 2; do
  python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline default(8-wave)" >> $O
  VITCAP_GEMM_4W_MIX_BIG=1 VITCAP_GEMM_4W_PF=0 python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline bf16->persistent4w noPF" >> $O
  VITCAP_GEMM_4W_MIX_BIG=1 python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline bf16->persistent4w PF" >> $O
  VITCAP_GEMM_4W=2 python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline all persistent4w PF" >> $O
  VITCAP_GEMM_4W_PF=0 python bench.py --steps 10 --warmup 2 --batch 512 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B512 one-stream noPF" >> $O
  python bench.py --steps 10 --warmup 2 --batch 512 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B512 one-stream PF" >> $O
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "B64 pipeline" >> $O
