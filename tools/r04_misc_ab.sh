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
run "B64 pipe default" X=1
run "B64 pipe qkv->4w form1" VITCAP_GEMM_4W_TILES_N=2304
run "B64 pipe qkv->4w form2" VITCAP_GEMM_4W_TILES_N=2304,2
run "B64 pipe fc1->4w form1" VITCAP_GEMM_4W_TILES_N=3072
run "B64 pipe fc1->4w form2" VITCAP_GEMM_4W_TILES_N=3072,2
run "B64 pipe res->4w form1" VITCAP_GEMM_4W_TILES_N=768
run "B64 pipe default" X=1
cat $out
