timeout 600 python -m pytest tests/test_hip_ops.py -q -k "4wave or tile_heights" -x 2>&1 | tail -15 > gpurun_out/t4w.log
timeout 600 python tools/gemm_bench.py 5,40 36928,295424 > gpurun_out/gemm_bench_4w.txt 2>&1
