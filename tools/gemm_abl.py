import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
for M, N, K in ((36928, 2304, 768), (295424, 2304, 768), (36928, 768, 3072)):
    for h, name in ((5, 'full'), (7, 'noDMA'), (8, 'noLDSread'), (9, 'noMFMA'), (10, 'MFMA+barriers'), (11, 'DMA+barriers'),
                    (16, 'no stores'), (17, 'no epilogue')):
        ms, tf = bench(M, N, K, L.ACT_NONE, 0, False, h, iters=20)
        print('M=%d N=%d K=%d %-14s %.3f ms (%.0f TF-equiv)' % (M, N, K, name, ms, tf))
