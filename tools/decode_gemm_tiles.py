"""Decode-step GEMM shapes (M = 2 rows per sequence) under each tile hint, next to the auto choice (GPU box).
    python tools/decode_gemm_tiles.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
Ms = [int(x) for x in sys.argv[1:]] or [1024]
for M in Ms:
    for name, N, K, act, f32, res in (('qkv', 2304, 768, L.ACT_NONE, 0, False), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False),
                                      ('out', 768, 768, L.ACT_NONE, 1, False), ('fc2', 768, 3072, L.ACT_NONE, 1, False)):
        row = []
        for h in (0, 1, 2, 3):
            us = min(bench(M, N, K, act, f32, res, h, iters=50)[0] for _ in range(2)) * 1e3
            row.append('hint%d %.1f' % (h, us))
        print('M=%d %-4s N=%4d K=%4d us: %s' % (M, name, N, K, ' | '.join(row)), flush=True)
