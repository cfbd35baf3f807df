"""Run one GEMM shape a few times (for rocprofv3 --pmc).  args: hint M N K act out_f32 res iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops
hint, M, N, K, act, of, res, iters = [int(x) for x in sys.argv[1:9]]
a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
bias = torch.rand(N, device='cuda')
r = torch.rand(M, N, device='cuda') if res else None
out = torch.empty(M, N, device='cuda', dtype=torch.float32 if of else torch.bfloat16)
for _ in range(iters):
    ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=hint)
torch.cuda.synchronize()
