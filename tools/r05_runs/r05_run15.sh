#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_ops.py tests/test_hip_train_e2e.py -m gpu -q -x -k "extras or fused_bias or gemm_with_training or test_losses or gradients_per_tensor or parameter_update or graph_step" 2>&1 | tail -4 > gpurun_out/r05_t15.log
for i in 1 2; do
python bench.py --mode train --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['frac'])" >> gpurun_out/r05_t15.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05x -o train -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --mode train > $GRAFT_REPO_ROOT/gpurun_out/prof_r05x.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_r05x -name "train_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py (train)" > gpurun_out/prof_r05x_train.md 2>&1 || true
rm -rf gpurun_out/prof_r05x
