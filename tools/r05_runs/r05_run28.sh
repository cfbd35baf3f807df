#!/bin/bash
# input side: the slow mode seen in 2 of 4 runs of the round script (copies 22 ms per batch): with and without the ceiling arms / decode sweeps in front
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t28.log
echo "--- as in the round script (decode sweeps + ceiling arms first)" >> gpurun_out/r05_t28.log
INPUT_SIDE_CEILING=1 INPUT_SIDE_WORKERS=8,10,8,10 python tools/input_side_bench.py 24576 2>&1 | grep -E "ceiling|num_workers|device alloc" | cut -c1-400 >> gpurun_out/r05_t28.log
echo "--- loader runs only" >> gpurun_out/r05_t28.log
INPUT_SIDE_SKIP_DECODE=1 INPUT_SIDE_WORKERS=8,10,8,10 python tools/input_side_bench.py 24576 gpurun_out/r05_input_side_final.json 2>&1 | grep -E "num_workers|device alloc" | cut -c1-400 >> gpurun_out/r05_t28.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/r05_t28.log
