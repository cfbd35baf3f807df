#!/bin/bash
# attention backward: prefetch variants (v4 = second half's fragments requested early, v5 = v4 + s_setprio) against the shipped form, one box;
# then the loader's new defaults (10 batches decoded ahead, 24 queued) and the training kernel profile
cp vitcap_amd/libvitcap_hip.so /tmp/lib_main.so
for rep in 1 2; do for v in main v4 v5; do
  if [ $v = main ]; then cp /tmp/lib_main.so vitcap_amd/libvitcap_hip.so; else cp tools/probes/_bin/libvitcap_bwd_$v.so vitcap_amd/libvitcap_hip.so; fi
  echo "== $v" >> gpurun_out/r05_t26.log
  python tools/attn_bwd_bench.py 2>&1 | grep -E "^(encoder|decoder|enc B)" >> gpurun_out/r05_t26.log
  python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('train $v', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r05_t26.log
done; done
cp /tmp/lib_main.so vitcap_amd/libvitcap_hip.so
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t26.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8,10,8,10 python tools/input_side_bench.py 24576 2>&1 | grep -E "num_workers|steady" | cut -c1-420 >> gpurun_out/r05_t26.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t26.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t26 -o train -- python3 $R/bench.py --steps 3 --warmup 1 --mode train > $R/gpurun_out/prof_t26.log 2>&1
cd $R
DB=$(find gpurun_out/prof_t26 -name "train_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py (train)" > gpurun_out/r05_t26_train_kernels.md 2>&1 || true
rm -rf gpurun_out/prof_t26
