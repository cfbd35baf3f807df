"""Yardstick, not product: what the libraries in this image (hipBLASLt behind torch.nn.functional.linear, the flash attention
behind scaled_dot_product_attention) reach on the hot shapes, next to this repo's kernels measured the same way (back-to-back
launches on one stream, torch events).  Nothing under vitcap_amd/ calls these libraries.  Run on the GPU box:
    python tools/library_yardstick.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vitcap_amd import _lib as L
from vitcap_amd import ops


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3       # us


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
    dev = 'cuda'
    print('torch', torch.__version__, torch.cuda.get_device_name(0))
    print('--- GEMMs, M = %d (us per launch, TFLOP/s)' % M)
    shapes = [('qkv  bias -> bf16', 2304, 768, L.ACT_NONE, False), ('proj bias+res -> f32', 768, 768, L.ACT_NONE, True),
              ('fc1  bias+gelu -> bf16', 3072, 768, L.ACT_GELU_ERF, False), ('fc2  bias+res -> f32', 768, 3072, L.ACT_NONE, True)]
    for name, N, K, act, res in shapes:
        a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)
        bias = torch.rand(N, device=dev)
        bias16 = bias.to(torch.bfloat16)
        r = torch.rand(M, N, device=dev) if res else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else torch.bfloat16)
        gf = 2.0 * M * N * K / 1e9
        rows = []
        for label, hint in (('ours 8-wave 256-row tiles', 32), ('ours 8-wave planned mix', 33), ('ours 4-wave one tile', 41), ('ours 4-wave persistent', 42)):
            us = timed(lambda: ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=hint))
            rows.append('%s %.1f us %.0f TF' % (label, us, gf / us * 1e3))
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timed(lambda: torch.mm(a, w.t(), out=o16))
        rows.append('lib mm only (bf16 out) %.1f us %.0f TF' % (us, gf / us * 1e3))
        us = timed(lambda: torch.addmm(bias16, a, w.t(), out=o16))
        rows.append('lib addmm bias (bf16 out) %.1f us %.0f TF' % (us, gf / us * 1e3))
        if act == L.ACT_GELU_ERF:
            us = timed(lambda: F.gelu(F.linear(a, w, bias16)))
            rows.append('lib linear + gelu kernel %.1f us' % us)
        if res:
            us = timed(lambda: torch.add(r, F.linear(a, w, bias16), out=out))
            rows.append('lib linear + fp32 residual add kernel %.1f us' % us)
        print('%-24s N=%4d K=%4d | ' % (name, N, K) + ' | '.join(rows))
    print('--- dense attention, 12 heads x 64 (us per launch, TFLOP/s of 4*B*12*S*S*64)')
    for B, S in ((M // 577 if M % 577 == 0 else 64, 577), (64, 578)):
        qkv = (torch.randn(B * S, 2304, device=dev) * 0.5).to(torch.bfloat16)
        gf = 4.0 * B * 12 * S * S * 64 / 1e9
        us = timed(lambda: ops.attn_dense(qkv, B, S))
        q, k, v = (t.reshape(B, S, 12, 64).transpose(1, 2) for t in qkv.split(768, dim=1))
        qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
        row = ['ours (reads the packed qkv rows) %.1f us %.0f TF' % (us, gf / us * 1e3)]
        for label, args in (('lib sdpa, strided views of qkv', (q, k, v)), ('lib sdpa, contiguous (B,H,S,d)', (qc, kc, vc))):
            try:
                us = timed(lambda: F.scaled_dot_product_attention(*args))
                row.append('%s %.1f us %.0f TF' % (label, us, gf / us * 1e3))
            except Exception as e:          # a yardstick that cannot run is reported, not fatal
                row.append('%s failed: %s' % (label, str(e)[:80]))
        print('B=%d S=%d | ' % (B, S) + ' | '.join(row))


if __name__ == '__main__':
    main()
