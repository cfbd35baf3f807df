#!/bin/bash
mkdir -p gpurun_out
for v in base pf pfall; do for m in 295424 36928; do
  echo "variant $v"; tools/probes/_bin/g4w_probe_$v $m 2 | cut -c1-250
done; done > gpurun_out/r05_g4w_probe3.txt 2>&1
python -m pytest tests/test_hip_train_e2e.py -m gpu -q -x -s -k "graph_step or trains_on_tsv or checkpoint or resume" 2>&1 | grep -E "graph vs|dropout on|masked tokens|passed|failed|Error|error|assert" | tail -30 > gpurun_out/r05_t7.log
