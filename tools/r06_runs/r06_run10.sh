#!/bin/bash
# round 6, call 10: two-rank graph-mode DP step (gloo, one GPU); hardware-queue count against the headline pipeline (5 streams on 4 default queues)
O=gpurun_out/r06_run10.txt
: > $O
timeout 1200 python -m pytest tests/test_hip_train_e2e.py -m gpu -q -x -k "two_process_data_parallel" 2>&1 | tail -3 >> $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'J/step', e.get('joules_per_step'))"; }
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | line "B64 default queues" >> $O
  GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | line "B64 GPU_MAX_HW_QUEUES=8" >> $O
  GPU_MAX_HW_QUEUES=6 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | line "B64 GPU_MAX_HW_QUEUES=6" >> $O
done
