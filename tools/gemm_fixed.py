import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
for N in (2304, 768):
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        for h in (5, 2):
            for of in (0, 1):
                ms, tf = bench(36928, N, K, L.ACT_NONE, of, False, h, iters=20)
                print('N=%d K=%d hint=%d out=%s: %.1f us (%.0f TF)' % (N, K, h, 'f32' if of else 'bf16', ms * 1e3, tf))
