#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, the bench lines of BASELINE configs[1..4], rocprof summaries, PMC traffic.
# Usage: bash scripts_gpu_round.sh [tag] [notests]
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/tests_$TAG.log
python -m pytest tests/test_hip_e2e.py -m gpu -q -s -k population 2>&1 | grep "population" > gpurun_out/population_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
fi
# HBM traffic of the kernels (PMC passes) FIRST: bench.py reads the dominant kernel's bytes per launch from profiles/<tag>_hbm_traffic_pmc*.json
bash tools/pmc_traffic.sh $TAG > gpurun_out/traffic_$TAG.log 2>&1
bash tools/pmc_traffic.sh ${TAG}_b512 "--batch 512" > gpurun_out/traffic_${TAG}_b512.log 2>&1
cp gpurun_out/traffic_$TAG.json profiles/${TAG}_hbm_traffic_pmc.json
cp gpurun_out/traffic_${TAG}_b512.json profiles/${TAG}_hbm_traffic_pmc_b512.json
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/bench_long_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 50 --warmup 3 --pipeline 0 --no-cpu-baseline > gpurun_out/bench_seq_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 20 --warmup 2 --beams 5 --batch 256 --graph 1 --no-cpu-baseline > gpurun_out/bench_beam5_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 30 --warmup 3 --mode train > gpurun_out/bench_train_$TAG.json 2>> gpurun_out/bench_$TAG.err
# the launcher + RCCL path on the one GPU of this box: a process group of ONE rank, the gradient exchange forced (VITCAP_DP_FORCE)
VITCAP_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --mode train > gpurun_out/bench_train_rccl1_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 30 --warmup 3 --mode scst > gpurun_out/bench_scst_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 10 --warmup 2 --batch 512 --no-cpu-baseline > gpurun_out/bench_b512_$TAG.json 2>> gpurun_out/bench_$TAG.err
python bench.py --steps 200 --warmup 5 --batch 1 --pipeline 0 --graph 1 --no-cpu-baseline > gpurun_out/bench_b1_$TAG.json 2>> gpurun_out/bench_$TAG.err
(python tools/library_yardstick.py; python tools/library_yardstick.py 295424) 2>&1 | grep -v amdgpu.ids > gpurun_out/library_yardstick_$TAG.txt
python bench.py --steps 30 --warmup 3 --mode train --train-graph 0 > gpurun_out/bench_train_eager_$TAG.json 2>> gpurun_out/bench_$TAG.err
python tools/exact_rate.py 2>&1 | grep EXACT > gpurun_out/exact_rate_$TAG.txt
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,8,6,6,8,8 INPUT_SIDE_DEVICE_JPEG=1,0,1,0,1,0 python tools/input_side_bench.py 24576 gpurun_out/input_side_$TAG.json > gpurun_out/input_side_$TAG.log 2>&1
bash tools/pmc_hot.sh $TAG > gpurun_out/pmc_hot_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o pipe -- python3 $R/bench.py --steps 20 --warmup 3 --isolated 0 --no-cpu-baseline --single-region --power 0 > $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o seq -- python3 $R/bench.py --steps 20 --warmup 3 --pipeline 0 --no-cpu-baseline --single-region --power 0 >> $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o train -- python3 $R/bench.py --steps 3 --warmup 1 --mode train >> $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b512 -- python3 $R/bench.py --steps 6 --warmup 2 --batch 512 --isolated 0 --no-cpu-baseline --single-region --power 0 >> $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
for k in pipe seq train b512; do
DB=$(find gpurun_out/prof_$TAG -name "${k}_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "bench.py ($k)" > gpurun_out/prof_${TAG}_${k}.md 2>&1 || true
done
# raw traces are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_$TAG gpurun_out/traffic_${TAG}_FETCH_SIZE gpurun_out/traffic_${TAG}_WRITE_SIZE gpurun_out/traffic_${TAG}_b512_FETCH_SIZE gpurun_out/traffic_${TAG}_b512_WRITE_SIZE
du -sh gpurun_out | tail -1
