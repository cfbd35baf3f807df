#!/bin/bash
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for sp in 2 1 4 3 2; do
  echo "B=512 VITCAP_ENCODE_SPLIT=$sp" >> gpurun_out/r05_b512_split.txt
  VITCAP_ENCODE_SPLIT=$sp python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_b512_split.txt
done
python -m pytest tests/test_hip_train_e2e.py -m gpu -q -x -k "rccl_exchange" 2>&1 | tail -2 >> gpurun_out/r05_b512_split.txt
