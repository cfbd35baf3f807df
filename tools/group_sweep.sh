for g in 0 2 3 4 6; do echo "GROUP_N=$g"; VITCAP_GEMM_GROUP_N=$g python tools/gemm_shapes_bench.py 2>&1 | grep "us:"; done
VITCAP_GEMM_GROUP_N=4 python tools/gemm_trace.py 36928 12 2>&1 | grep -A3 "^fc1 \|^qkv "
VITCAP_GEMM_GROUP_N=4 python -m pytest tests/test_hip_ops.py -q -m gpu 2>&1 | tail -2
