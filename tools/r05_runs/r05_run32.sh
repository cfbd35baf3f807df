#!/bin/bash
# dQ kernel at three waves per SIMD (v6: S^T and dP^T chains one after the other, tail mode as a template parameter, 168 registers) against
# the shipped library (main) and the same source at two waves (v7); parity tests on v6 first
cp vitcap_amd/libvitcap_hip.so /tmp/lib_main.so
cp tools/probes/_bin/libvitcap_bwd_v6.so vitcap_amd/libvitcap_hip.so
timeout 600 python -m pytest tests/test_hip_train_ops.py tests/test_hip_ops.py -q -x -k "attn" 2>&1 | tail -3 > gpurun_out/r05_t32_tests.log
for rep in 1 2; do for v in main v6 v7; do
  if [ $v = main ]; then cp /tmp/lib_main.so vitcap_amd/libvitcap_hip.so; else cp tools/probes/_bin/libvitcap_bwd_$v.so vitcap_amd/libvitcap_hip.so; fi
  echo "== $v" >> gpurun_out/r05_t32.log
  python tools/attn_bwd_bench.py 2>&1 | grep -E "^(encoder|decoder|enc B)" >> gpurun_out/r05_t32.log
  python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train $v', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t32.log
done; done
cp /tmp/lib_main.so vitcap_amd/libvitcap_hip.so
