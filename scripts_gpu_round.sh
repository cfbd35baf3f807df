#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof summary.  Usage: scripts_gpu_round.sh [tag]
TAG=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/tests_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python bench.py --steps 20 --warmup 3 --pipeline 0 --no-cpu-baseline > gpurun_out/bench_seq_$TAG.json 2>> gpurun_out/bench_$TAG.err
python tools/decode_bench.py 64 > gpurun_out/decode_bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o seq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --pipeline 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "bench.py --pipeline 0 --steps 3 (one stream, B=64)" > gpurun_out/prof_${TAG}_summary.md 2>&1 || true
