"""LayerNorm backward at the encoder shape under VITCAP_LNBWD_WAVES (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops
M = 36928
x = torch.randn(M, 768, device='cuda'); g = torch.rand(768, device='cuda') + 0.5
dres = torch.randn(M, 768, device='cuda')
for name, dy in (('dy bf16', torch.randn(M, 768, device='cuda').to(torch.bfloat16)), ('dy fp32', torch.randn(M, 768, device='cuda'))):
    dg = torch.zeros(768, device='cuda'); db = torch.zeros(768, device='cuda')
    for _ in range(3):
        ops.layernorm_bwd(x, dy, g, 1e-6, dg, db, dres=dres)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.layernorm_bwd(x, dy, g, 1e-6, dg, db, dres=dres)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = M * 768 * (4 + dy.element_size() + 4 + 4 + 2)
    print('WAVES=%s %s: %.1f us, %.2f TB/s' % (os.environ.get('VITCAP_LNBWD_WAVES', '16'), name, us, byt / us / 1e6), flush=True)
