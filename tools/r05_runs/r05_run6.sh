#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_e2e.py -m gpu -q -x -s -k "graph_step or test_losses or parameter_update or two_process_data or rccl_exchange" 2>&1 | grep -E "graph vs|dropout on|masked tokens|passed|failed|Error|error|assert" | tail -30 > gpurun_out/r05_t6.log
for g in 0 1; do
  python bench.py --mode train --steps 30 --warmup 3 --train-graph $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train graph', '$g', d['value'], d['ms_per_step'], 'host ms/step', d['host_issue_ms_per_step'], d['launch'][:60])" >> gpurun_out/r05_train_graph.txt
done
VITCAP_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --mode train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train graph + one-rank RCCL exchange', d['value'], d['ms_per_step'], 'host ms/step', d['host_issue_ms_per_step'], d['launch'][:80])" >> gpurun_out/r05_train_graph.txt
