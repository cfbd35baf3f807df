#!/bin/bash
# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py; writes gpurun_out/prof_<tag>/ and a markdown summary.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_$TAG/p_results.db "bench.py --steps 3 --warmup 1 $* (4 passes profiled)" > gpurun_out/prof_$TAG.md
head -30 gpurun_out/prof_$TAG.md
