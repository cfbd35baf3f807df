#!/bin/bash
# round 6, call 8: deferred stores in the persistent 4-wave bf16 GEMM -- parity, stamps, then the pipeline
O=gpurun_out/r06_run8.txt
: > $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -4 >> $O
for m in 36928 295424; do
  echo "== DEFER=0 M=$m" >> $O; VITCAP_GEMM_4W_DEFER=0 VITCAP_GEMM_4W_PF=0 tools/probes/_bin/g4w_probe_defer $m 2 | grep -E "qkv|fc1" | cut -c1-250 >> $O
  echo "== DEFER=1 M=$m" >> $O; tools/probes/_bin/g4w_probe_defer $m 2 | grep -E "qkv|fc1" | cut -c1-250 >> $O
done
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'W', e.get('board_power_w_median'), 'J/step', e.get('joules_per_step'))"; }
for rep in 1 2; do
  VITCAP_GEMM_4W_DEFER=0 python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline DEFER=0 (PF)" >> $O
  python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 pipeline DEFER=1" >> $O
  VITCAP_GEMM_4W_DEFER=0 python bench.py --steps 10 --warmup 2 --batch 512 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B512 one-stream DEFER=0 (PF)" >> $O
  python bench.py --steps 10 --warmup 2 --batch 512 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B512 one-stream DEFER=1" >> $O
  VITCAP_GEMM_4W_DEFER=0 python bench.py --steps 30 --warmup 3 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B64 one-stream DEFER=0" >> $O
  python bench.py --steps 30 --warmup 3 --pipeline 0 --no-cpu-baseline 2>/dev/null | line "B64 one-stream DEFER=1" >> $O
  VITCAP_GEMM_4W_DEFER=0 python bench.py --steps 20 --warmup 3 --mode train 2>/dev/null | line "train DEFER=0" >> $O
  python bench.py --steps 20 --warmup 3 --mode train 2>/dev/null | line "train DEFER=1" >> $O
done
