#!/bin/bash
# usage (GPU box): bash tools/exact_rate_ab.sh "<EXTRA flags variant 1>" ...  -> exact-match rates + end-to-end img/s per attention build
R=$GRAFT_REPO_ROOT
cd $R
for V in "$@" ""; do
  touch vitcap_amd/csrc/attn.hip
  make -C vitcap_amd/csrc EXTRA="$V" 2>&1 | grep -E "error|spill" | head -5
  echo "=== attention build [$V]"
  python tools/exact_rate.py 2>&1 | grep EXACT
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('   %.1f img/s' % d['value'])"
done
