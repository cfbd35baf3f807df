#!/bin/bash
# round 6, call 11: dense attention with buffer-descriptor LDS-DMA + max3 tree against round 5's kernel (same arithmetic: identical bits)
O=gpurun_out/r06_run11.txt
: > $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attn" 2>&1 | tail -3 >> $O
R=$PWD
for rep in 1 2; do
  (cd $R/_ab/r05 && python $R/tools/attn_time.py 64 2>/dev/null | sed 's/^/r05  /' >> $R/$O)
  python tools/attn_time.py 64 2>/dev/null | sed 's/^/HEAD /' >> $O
done
(cd $R/_ab/r05 && python $R/tools/attn_time.py 32 2>/dev/null | sed 's/^/r05  /' >> $R/$O)
python tools/attn_time.py 32 2>/dev/null | sed 's/^/HEAD /' >> $O
timeout 1500 python -m pytest tests/test_hip_e2e.py -m gpu -q -x 2>&1 | tail -3 >> $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'J/step', e.get('joules_per_step'))"; }
for rep in 1 2 3; do
  (cd $R/_ab/r05 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "B64 r05" >> $R/$O)
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "B64 HEAD" >> $O
done
