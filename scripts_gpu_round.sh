#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof summary.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > gpurun_out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err
python bench.py --steps 10 --warmup 2 --graph 1 --no-cpu-baseline > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
