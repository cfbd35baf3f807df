R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_beam -o beam -- python3 $R/bench.py --steps 3 --warmup 1 --beams 5 --batch 256 --pipeline 0 --no-cpu-baseline > $R/gpurun_out/prof_beam.log 2>&1
cd $R
DB=$(find gpurun_out/prof_beam -name "beam_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py --beams 5 --batch 256 --pipeline 0 (one stream)" > gpurun_out/prof_beam.md
rm -rf gpurun_out/prof_beam
