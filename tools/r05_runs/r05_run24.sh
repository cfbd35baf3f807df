#!/bin/bash
# attention backward variants, one box: v1 = LDS-DMA pair as first measured, v2 = dkv reads L / D with the fragments (shipped library), v3 = v2 + s_setprio
cp vitcap_amd/libvitcap_hip.so /tmp/lib_v2.so
for rep in 1 2; do for v in v1 v2 v3; do
  if [ $v = v2 ]; then cp /tmp/lib_v2.so vitcap_amd/libvitcap_hip.so; else cp tools/probes/_bin/libvitcap_bwd_$v.so vitcap_amd/libvitcap_hip.so; fi
  echo "== $v" >> gpurun_out/r05_t24.log
  python tools/attn_bwd_bench.py 2>&1 | grep -E "^(encoder|decoder|enc B)" >> gpurun_out/r05_t24.log
  python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train $v', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t24.log
done; done
cp /tmp/lib_v2.so vitcap_amd/libvitcap_hip.so
