"""Decode-shape GEMM with weights that are cold in L2 / MALL (rotating through many copies), for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L
ncopy = int(sys.argv[1]) if len(sys.argv) > 1 else 120
x = ((torch.rand(128, 768, device='cuda') * 2 - 1)).to(torch.bfloat16)
ws = [((torch.rand(2304, 768, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
bias = torch.rand(2304, device='cuda')
for hint in (1, 13, 14):
    for rep in range(2):
        for w in ws:
            ops.gemm_bias_act(x, w, bias, tile_hint=hint)
torch.cuda.synchronize()
