import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
for M, N, K, act, f32, res in ((36928, 2304, 768, L.ACT_NONE, 0, False), (36928, 3072, 768, L.ACT_GELU_ERF, 0, False),
                               (36928, 768, 768, L.ACT_NONE, 1, True), (36928, 768, 3072, L.ACT_NONE, 1, True)):
    for h, name in ((5, 'full'), (10, 'MFMA+barriers'), (11, 'DMA+barriers'), (16, 'no stores'), (17, 'no epilogue'), (0, 'default dispatch')):
        ms, tf = bench(M, N, K, act, f32, res, h, iters=30)
        print('M=%d N=%d K=%d act=%d f32=%d res=%d %-16s %.1f us (%.0f TF-equiv)' % (M, N, K, act, f32, res, name, ms * 1e3, tf), flush=True)
