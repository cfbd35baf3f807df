"""Run attn_dense a few times (for rocprofv3).  args: B S iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops
B, S, iters = [int(x) for x in sys.argv[1:4]]
qkv = ((torch.rand(B * S, 2304, device='cuda') * 2 - 1) * 2).to(torch.bfloat16)
for _ in range(iters):
    o = ops.attn_dense(qkv, B, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    o = ops.attn_dense(qkv, B, S)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('attn_dense B=%d S=%d: %.1f us, %.0f TF' % (B, S, ms * 1e3, 4.0 * B * 12 * S * S * 64 / ms / 1e9))
