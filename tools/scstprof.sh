R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_scst -o scst -- python3 $R/bench.py --steps 4 --warmup 1 --mode scst > $R/gpurun_out/prof_scst.log 2>&1
cd $R
DB=$(find gpurun_out/prof_scst -name "scst_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py --mode scst (16 images x 5 samples)" > gpurun_out/prof_scst.md
rm -rf gpurun_out/prof_scst
