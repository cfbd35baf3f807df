#!/bin/bash
# round 6, call 7: JPEG back half, second form of the kernels (coalesced loads through LDS, 24-bit multiplies, shared chroma loads)
O=gpurun_out/r06_run7.txt
: > $O
python -m pytest tests/test_hip_jpeg.py tests/test_hip_image_transform.py -m gpu -q -x 2>&1 | tail -3 >> $O
python tools/jpeg_bench.py 64 >> $O 2>&1
python tools/jpeg_bench.py 512 >> $O 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_jpeg -o j -- python3 $GRAFT_REPO_ROOT/tools/jpeg_bench.py 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_jpeg.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_jpeg -name "j_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "tools/jpeg_bench.py 64" > gpurun_out/r06_jpeg_kernel_stats.md 2>&1 || true
rm -rf gpurun_out/prof_jpeg
