#!/bin/bash
# energy balance of the 2-slot pipeline: board power of each phase on its own (about 8 s each), B = 64
for ph in "enc 600" "dec 1600" "pipe 500"; do
  set -- $ph
  echo "== phase $1" >> gpurun_out/r05_t34.log
  python tools/power_during.py 9 -- python tools/encode_only_bench.py 64 $2 $1 2>&1 | grep -E "^B=|power_during" >> gpurun_out/r05_t34.log
done
