#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/misc_ab_r04.txt
: > $out
run() { # label, env...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --isolated 0 $EXTRA 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "frac", r.get("frac"))')" >> $out
}
run "B64 pipe eager" X=1
EXTRA="--graph 1"
run "B64 pipe graph-replayed decode" X=1
EXTRA=""
run "B64 pipe encode split 2" VITCAP_ENCODE_SPLIT=2
run "B64 pipe decode prio 0" VITCAP_DECODE_PRIORITY=0
run "B64 pipe eager" X=1
cat $out
