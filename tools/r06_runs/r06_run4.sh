#!/bin/bash
# round 6, call 4: decode-step GEMM forms at M = 1024 / 2560 rows (B = 512 greedy, beam 5 x 256); beam 5 x 256 with the bf16 GEMMs persistent from 64k rows on
O=gpurun_out/r06_run4.txt
: > $O
python tools/decode_gemm_forms.py 1024 2560 >> $O 2>&1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'dec', d.get('decode_phase_ms_per_batch'), 'W', e.get('board_power_w_median'), 'J/step', e.get('joules_per_step'))"; }
for rep in 1 2; do
  python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline --isolated 0 2>/dev/null | line "beam5x256 default" >> $O
  VITCAP_GEMM_4W_MIX_BIG=1 python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline --isolated 0 2>/dev/null | line "beam5x256 MIX_BIG" >> $O
done
