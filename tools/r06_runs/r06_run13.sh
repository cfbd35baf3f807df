#!/bin/bash
# round 6, call 13: kernel profile of BASELINE configs[2] (beam 5 x 256, pipeline, graph-replayed decode)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_beam -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline --isolated 0 --single-region --power 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_beam.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_beam -name "b_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py --beams 5 --batch 256 --graph 1 (pipeline)" > gpurun_out/r06_beam5_kernel_stats.md 2>&1 || true
rm -rf gpurun_out/prof_beam
