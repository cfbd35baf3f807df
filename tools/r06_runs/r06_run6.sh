#!/bin/bash
# round 6, call 6: (a) the jpeg tests after the colour-kernel change + the back half against its HBM roofline; (b) HEAD against round 5's tree
# (_ab/r05 = git archive of 8ba16cf, built here) on ONE box, interleaved: the training step and the headline
O=gpurun_out/r06_run6.txt
: > $O
python -m pytest tests/test_hip_jpeg.py -m gpu -q -x 2>&1 | tail -3 >> $O
python tools/jpeg_bench.py 64 >> $O 2>&1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d.get('extra') or {}
print('$1', d['value'], d['ms_per_step'], 'J/step', e.get('joules_per_step'))"; }
R=$PWD
for rep in 1 2; do
  (cd $R/_ab/r05 && python bench.py --steps 30 --warmup 3 --mode train 2>/dev/null | line "train r05" >> $R/$O)
  python bench.py --steps 30 --warmup 3 --mode train 2>/dev/null | line "train HEAD" >> $O
  (cd $R/_ab/r05 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "B64 r05" >> $R/$O)
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "B64 HEAD" >> $O
done
(cd $R/_ab/r05 && python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 r05" >> $R/$O)
python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 HEAD" >> $O
(cd $R/_ab/r05 && python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 r05" >> $R/$O)
python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | line "B512 HEAD" >> $O
