#!/bin/bash
# marginal cost of a decode step inside the 2-slot pipeline: max_length 20 / 11 / 6 / 3 (19 / 10 / 5 / 2 steps), two reps
for rep in 1 2; do for ml in 20 11 6 3; do
  python tools/encode_only_bench.py 64 200 pipe $ml 2>&1 | grep "^B=" >> gpurun_out/r05_t35.log
done; done
python tools/encode_only_bench.py 64 200 enc 2>&1 | grep "^B=" >> gpurun_out/r05_t35.log
