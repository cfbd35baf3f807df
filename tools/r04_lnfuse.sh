#!/bin/bash
# GPU round: GEMMs that normalise their own rows (last-arriver LayerNorm) -- parity + A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fused_layernorm" -x > gpurun_out/tests_lnfuse_r04.log 2>&1
echo "ops rc=$?" >> gpurun_out/tests_lnfuse_r04.log
out=gpurun_out/lnfuse_r04.txt
: > $out
run() { # label, env...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --isolated 0 $EXTRA 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "frac", r.get("frac"), "avg_ms", r.get("avg_launch_ms"))')" >> $out
}
run "B64 pipe fused(plain loads)" X=1
run "B64 pipe unfused" VITCAP_GEMM_LN_FUSE=0
run "B64 pipe publish only (no LN pass: wrong results)" VITCAP_GEMM_LN_DEBUG=1
run "B64 pipe fused(plain loads)" X=1
cat $out
