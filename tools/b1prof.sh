R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b1 -o b1 -- python3 $R/bench.py --steps 20 --warmup 3 --batch 1 --pipeline 0 --graph 0 --no-cpu-baseline > $R/gpurun_out/prof_b1.log 2>&1
cd $R
DB=$(find gpurun_out/prof_b1 -name "b1_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py B=1" > gpurun_out/prof_b1.md
