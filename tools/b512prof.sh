R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b512 -o b512 -- python3 $R/bench.py --steps 6 --warmup 2 --batch 512 --isolated 0 --no-cpu-baseline > $R/gpurun_out/prof_b512.log 2>&1
cd $R
DB=$(find gpurun_out/prof_b512 -name "b512_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py --batch 512 (2-slot pipeline)" > gpurun_out/prof_r03_b512.md 2>&1
rm -rf gpurun_out/prof_b512
tail -1 gpurun_out/prof_b512.log | cut -c1-300
head -45 gpurun_out/prof_r03_b512.md | cut -c1-160
