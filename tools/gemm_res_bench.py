"""Residual-epilogue GEMM shapes: persistent (hint 12) vs non-persistent (hint 5) at B=64 and B=512 rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
for M in (36928, 295424):
    for name, N, K, act, of, res in (('proj', 768, 768, L.ACT_NONE, 1, True), ('fc2', 768, 3072, L.ACT_NONE, 1, True),
                                     ('qkv', 2304, 768, L.ACT_NONE, 0, False), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False)):
        row = []
        for h in (5, 12):
            ms, tf = bench(M, N, K, act, of, res, h)
            row.append('hint%d %.3f ms %.0f TF' % (h, ms, tf))
        print('M=%d %-5s N=%d K=%d : %s' % (M, name, N, K, ' | '.join(row)))
