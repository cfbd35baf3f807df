#!/bin/bash
# headline: encoder parts 2 / 3 / 4 at B = 64 (two reps, interleaved), one box
for rep in 1 2; do for p in 2 3 4; do
  VITCAP_ENCODE_SPLIT=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('parts=$p', d['value'], d['ms_per_step'])" >> gpurun_out/r05_t33.log
done; done
