"""One weight-gradient shape, a few launches (for PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops
M, N, K, S = 36928, 2304, 768, 9
y = (torch.rand(M, N, device='cuda') - 0.5).to(torch.bfloat16)
x = (torch.rand(M, K, device='cuda') - 0.5).to(torch.bfloat16)
slabs = torch.empty(S, N, K, device='cuda')
for _ in range(4):
    ops.gemm_tn(y, x, S, slabs)
torch.cuda.synchronize()
