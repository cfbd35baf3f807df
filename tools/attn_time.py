"""Dense attention alone: us per launch at the encoder's shape (B images x 12 heads, S = 577) and the prefill's (S = 578), back-to-back launches.
    python tools/attn_time.py [B]        (run from a tree's root: imports that tree's vitcap_amd)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from vitcap_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for S in (577, 578):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.5).to(torch.bfloat16)
    for _ in range(5):
        o = ops.attn_dense(qkv, B, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(200):
            o = ops.attn_dense(qkv, B, S)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    import hashlib
    h = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    print('B=%d S=%d: %.1f us per launch = %.0f TF/s; sha256 of the output %s' % (B, S, best, 4.0 * B * 12 * S * S * 64 / best * 1e-6, h))
