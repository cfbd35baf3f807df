#!/bin/bash
python tools/gemm_extras_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_t27.log
python tools/gemm_extras_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_t27.log
