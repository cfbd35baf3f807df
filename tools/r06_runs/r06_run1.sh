#!/bin/bash
# round 6, call 1: (a) the persistent 4-wave GEMM at MI = 4 (128 x 256 workgroup tiles) against the shipped MI = 7/8: what a
# half-height tile's main loop costs per flop (the deferred-epilogue form would run at the loop's rate); (b) matrix pipe power per MFMA shape
O=gpurun_out/r06_run1.txt
: > $O
for m in 36928 295424; do
  echo "== default MI, M=$m" >> $O; tools/probes/_bin/g4w_probe_mi4 $m 2 | cut -c1-260 >> $O
  echo "== MI=4, M=$m" >> $O; VITCAP_GEMM4W_MI=4 tools/probes/_bin/g4w_probe_mi4 $m 2 | cut -c1-260 >> $O
done
python tools/mfma_power.py 4 >> $O 2>&1
