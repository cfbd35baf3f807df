# bench.py's kernel-bound event timing against rocprofv3 --kernel-trace of the same command
R=$GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/agree_bench.json 2> gpurun_out/agree.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_agree -o pipe -- python3 $R/bench.py --steps 20 --warmup 3 --isolated 0 --no-cpu-baseline > $R/gpurun_out/agree_prof_bench.json 2>> $R/gpurun_out/agree.err
cd $R
DB=$(find gpurun_out/prof_agree -name "pipe_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py (pipelined)" > gpurun_out/agree_prof.md 2>&1
rm -rf gpurun_out/prof_agree
