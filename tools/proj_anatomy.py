"""What the fp32-residual epilogue costs: the N = 768 GEMMs with bf16 / fp32 output, with and without the residual (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
for name, N, K in (('proj', 768, 768), ('fc2', 768, 3072)):
    row = []
    for label, f32, res in (('bf16 out', 0, False), ('fp32 out', 1, False), ('fp32 out + fp32 residual', 1, True)):
        for hint in (5, 33):
            us = min(bench(M, N, K, L.ACT_NONE, f32, res, hint, iters=40)[0] for _ in range(2)) * 1e3
            row.append('%s hint%d %.1f' % (label, hint, us))
    print('M=%d %s: %s' % (M, name, ' | '.join(row)), flush=True)
