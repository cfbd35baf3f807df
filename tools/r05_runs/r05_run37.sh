#!/bin/bash
# decode attention with the next sweep's loads requested ahead (88 registers: no longer resident next to two GEMM waves) against the shipped kernel
cp vitcap_amd/libvitcap_hip.so /tmp/lib_new.so
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py -q -x -m gpu -k "decode or greedy or golden or async or batch64" 2>&1 | tail -3 > gpurun_out/r05_t37_tests.log
for rep in 1 2 3; do for v in old new; do
  if [ $v = new ]; then cp /tmp/lib_new.so vitcap_amd/libvitcap_hip.so; else cp tools/probes/_bin/libvitcap_dec_old.so vitcap_amd/libvitcap_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B64 $v', d['value'], d['ms_per_step'])" >> gpurun_out/r05_t37.log
done; done
for v in old new old new; do
  if [ $v = new ]; then cp /tmp/lib_new.so vitcap_amd/libvitcap_hip.so; else cp tools/probes/_bin/libvitcap_dec_old.so vitcap_amd/libvitcap_hip.so; fi
  python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B512 $v', d['value'], d['ms_per_step'])" >> gpurun_out/r05_t37.log
  python bench.py --steps 50 --warmup 3 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('seq $v', d['value'], d['ms_per_step'])" >> gpurun_out/r05_t37.log
  python tools/encode_only_bench.py 64 300 dec 2>&1 | grep "^B=" | sed "s/^/$v /" >> gpurun_out/r05_t37.log
done
cp /tmp/lib_new.so vitcap_amd/libvitcap_hip.so
