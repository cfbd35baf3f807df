#!/bin/bash
# usage (GPU box): bash tools/prof_script.sh <tag> <python script> [args]  -> per-kernel rocprofv3 stats of any script
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/"$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_$TAG/p_results.db "$*" > gpurun_out/prof_$TAG.md
head -40 gpurun_out/prof_$TAG.md
