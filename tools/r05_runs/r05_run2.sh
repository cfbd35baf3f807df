#!/bin/bash
# round 5, second GPU call: beam-5 regression cause (8-wave kernel forced), host CPU allotment, measured deltas of the tests to tighten
mkdir -p gpurun_out
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; grep -c processor /proc/cpuinfo; cat /proc/loadavg; df -h /dev/shm | tail -1 ) > gpurun_out/r05_host.txt 2>&1
for i in 1 2; do
  echo "HEAD lib, 8-wave forced under tile_hint 5" >> gpurun_out/r05_beam2.txt
  VITCAP_GEMM_4W=-1,2 python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_phase_ms_per_batch'))" >> gpurun_out/r05_beam2.txt
  echo "HEAD lib default" >> gpurun_out/r05_beam2.txt
  python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_phase_ms_per_batch'))" >> gpurun_out/r05_beam2.txt
  echo "HEAD lib, 8-wave forced, split 1" >> gpurun_out/r05_beam2.txt
  VITCAP_ENCODE_SPLIT=1 VITCAP_GEMM_4W=-1,2 python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_phase_ms_per_batch'))" >> gpurun_out/r05_beam2.txt
done
for i in 1 2; do for v in "-1,2" "2,2"; do
  echo "B=512 greedy, VITCAP_GEMM_4W=$v" >> gpurun_out/r05_b512_forms.txt
  VITCAP_GEMM_4W=$v python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r05_b512_forms.txt
done; done
python -m pytest tests/test_hip_cbs.py tests/test_hip_train_e2e.py -m gpu -q -s -k "cbs_captions or trajectory or test_losses or gradients_per_tensor" 2>&1 | grep -E "case|trajectory|loss hip|e-0|passed|failed" > gpurun_out/r05_deltas.txt
