"""Attention backward (vitcap_attn_dense_bwd: dQ kernel + dK/dV kernel) per launch at the training shapes, and a checksum of the result
so that two builds can be compared bit for bit (round 5 compared the LDS-DMA pair with the first pair this way:
profiles/r05_train_attn_bwd_ab.txt):  python tools/attn_bwd_bench.py
  encoder   B = 64, S = 577, no dropout          decoder   B = 64, S = 578 + 40 caption rows, causal_from = 578, dropout 0.1"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitcap_amd import ops


def run(name, B, S, iters=30, **kw):
    g = torch.Generator().manual_seed(7)
    qkv = (torch.randn(B * S, 2304, generator=g) * 1.2).to(torch.bfloat16).cuda()
    dout = torch.randn(B * S, 768, generator=g).to(torch.bfloat16).cuda()
    out, lse = ops.attn_dense_train(qkv, B, S, **kw)
    d = ops.attn_dense_bwd(qkv, out, dout, lse, B, S, **kw)
    torch.cuda.synchronize()
    digest = hashlib.sha256(d.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    dq = torch.empty_like(qkv)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for i in range(iters):
        ev[i].record()
        ops.attn_dense_bwd(qkv, out, dout, lse, B, S, dqkv=dq, **kw)
    ev[iters].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(5, iters))
    flops = 7 * 2.0 * B * 12 * S * S * 64
    print('%-8s %.1f us per backward (median; min %.1f)  %.0f TFLOP/s over the 7 matmuls  sha %s' % (
        name, ts[len(ts) // 2], ts[0], flops / ts[len(ts) // 2] / 1e6, digest), flush=True)


if __name__ == '__main__':
    run('encoder', 64, 577)
    run('decoder', 64, 618, p_drop=0.1, drop_seed=1234, causal_from=578)
    run('enc B=8', 8, 577)
