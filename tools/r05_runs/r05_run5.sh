#!/bin/bash
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
python -m pytest tests/test_hip_e2e.py tests/test_hip_ops.py -m gpu -q -x -k "golden or batch64 or pipeline_equals or gemm or layernorm or attn_dense or ragged" 2>&1 | tail -4 > gpurun_out/r05_t5.log
for i in 1 2 3; do for v in 0 1; do
  echo "B=64 pipeline, VITCAP_ZIGZAG=$v" >> gpurun_out/r05_zigzag.txt
  VITCAP_ZIGZAG=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_zigzag.txt
done; done
for v in 0 1; do
  echo "B=64 one stream, VITCAP_ZIGZAG=$v" >> gpurun_out/r05_zigzag.txt
  VITCAP_ZIGZAG=$v python bench.py --steps 60 --warmup 5 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "$J" >> gpurun_out/r05_zigzag.txt
  echo "B=512, VITCAP_ZIGZAG=$v" >> gpurun_out/r05_zigzag.txt
  VITCAP_ZIGZAG=$v python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_zigzag.txt
  echo "beam 5 x 256, VITCAP_ZIGZAG=$v" >> gpurun_out/r05_zigzag.txt
  VITCAP_ZIGZAG=$v python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "$J" >> gpurun_out/r05_zigzag.txt
  echo "B=512 one stream, VITCAP_ZIGZAG=$v" >> gpurun_out/r05_zigzag.txt
  VITCAP_ZIGZAG=$v python bench.py --steps 6 --warmup 2 --batch 512 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "$J" >> gpurun_out/r05_zigzag.txt
done
