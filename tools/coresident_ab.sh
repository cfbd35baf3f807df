#!/bin/bash
# usage (GPU box): bash tools/coresident_ab.sh  -- tools/coresident_probe.py with the decode attention at 60 registers (fits next to two
# 224-register GEMM waves per SIMD) and at 80 (does not), then the pipelined bench in both builds.  Restores the default build.
R=$GRAFT_REPO_ROOT
cd $R
for V in "-DVC_ATTN_DECODE_FAT=1" ""; do
  touch vitcap_amd/csrc/attn.hip
  make -C vitcap_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" | head -5
  echo "=== build [$V]"
  for g in qkv fc1 fc2 proj; do python tools/coresident_probe.py $g 2>&1 | grep together; done
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
