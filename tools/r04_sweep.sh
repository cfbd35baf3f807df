#!/bin/bash
# batch-size sweep (pipelined + one-stream latency) and a kernel profile of a constrained-beam-search run
cd "$(dirname "$0")/.." || exit 1
R=$PWD
mkdir -p gpurun_out
out=gpurun_out/batch_sweep_r04.txt
: > $out
for b in 1 8 16 32 64 128 256 512; do
  s=40; [ $b -ge 128 ] && s=16; [ $b -ge 512 ] && s=8
  python bench.py --batch $b --steps $s --warmup 3 --no-cpu-baseline --isolated 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d pipelined %.1f img/s %.3f ms/batch' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))" >> $out
done
for b in 1 8 64; do
  python bench.py --batch $b --steps 40 --warmup 3 --no-cpu-baseline --pipeline 0 --isolated 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d one-stream %.1f img/s %.3f ms (latency)' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))" >> $out
done
cat $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cbs -o cbs -- python3 $R/tools/cbs_bench.py 16 5 > $R/gpurun_out/prof_cbs.log 2>&1
cd $R
DB=$(find gpurun_out/prof_cbs -name "cbs_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "tools/cbs_bench.py 16 5 (constrained beam search: 16 images x 8 states x 5 beams, then plain beam 5)" > gpurun_out/prof_r04_cbs.md 2>&1 || true
rm -rf gpurun_out/prof_cbs
