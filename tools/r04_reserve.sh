#!/bin/bash
# experiment: persistent 4-wave GEMMs inside the 2-slot pipeline, tight grids / R CUs left to the decode chain
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/reserve_r04.txt
: > $out
run() { # label, env...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --isolated 0 $EXTRA 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $out
}
run "B64 default(8-wave)" X=1
run "B64 form2 tight" VITCAP_GEMM_4W=2,2
run "B64 form2 tight reserve=8" VITCAP_GEMM_4W=2,2 VITCAP_GEMM_4W_RESERVE=8
run "B64 form2 tight reserve=16" VITCAP_GEMM_4W=2,2 VITCAP_GEMM_4W_RESERVE=16
run "B64 form2 tight reserve=32" VITCAP_GEMM_4W=2,2 VITCAP_GEMM_4W_RESERVE=32
run "B64 form2 loose reserve=32" VITCAP_GEMM_4W=2,2 VITCAP_GEMM_4W_RESERVE=32 VITCAP_GEMM_4W_TIGHT=0
run "B64 form2 tight MI=8" VITCAP_GEMM_4W=2,2 VITCAP_GEMM4W_MI=8
run "B64 form2 tight MI=7" VITCAP_GEMM_4W=2,2 VITCAP_GEMM4W_MI=7
run "B64 default(8-wave) again" X=1
EXTRA="--pipeline 0"
run "B64 one-stream tight" X=1
run "B64 one-stream loose" VITCAP_GEMM_4W_TIGHT=0
EXTRA="--batch 512"
run "B512 tight" X=1
run "B512 loose" VITCAP_GEMM_4W_TIGHT=0
cat $out
