#!/bin/bash
# round 5, first GPU call: quick parity subset, encode_parts A/B, beam-5 regression A/B (round-3 library in _ab/r03), power probe
mkdir -p gpurun_out
python -m pytest tests/test_hip_e2e.py -m gpu -q -x -k "predict_honours or pipeline_equals or batch64" 2>&1 | tail -5 > gpurun_out/r05_t1.log
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "4wave or gemm" 2>&1 | tail -5 >> gpurun_out/r05_t1.log
for i in 1 2; do for sp in 1 2; do
  echo "split $sp" >> gpurun_out/r05_split.txt
  VITCAP_ENCODE_SPLIT=$sp python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'busy', r['frac_busy'])" >> gpurun_out/r05_split.txt
done; done
for i in 1 2; do
  echo "r03 lib" >> gpurun_out/r05_beam.txt
  python _ab/r03/bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_phase_ms_per_batch'))" >> gpurun_out/r05_beam.txt
  echo "HEAD lib" >> gpurun_out/r05_beam.txt
  python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_phase_ms_per_batch'))" >> gpurun_out/r05_beam.txt
done
python tools/power_probe.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_power_probe.txt
for m in 73856 147712; do tools/probes/_bin/g4w_probe_base $m 2 | cut -c1-250; done > gpurun_out/r05_g4w_probe_msweep.txt 2>&1
