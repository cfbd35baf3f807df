"""Residual GEMM + LayerNorm: one launch (the kernel normalises its own rows) against GEMM launch + LayerNorm launch, back to back on
an otherwise idle GPU.  python tools/lnfuse_bench.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
    for K in (768, 3072):
        a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(768, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
        bias = torch.rand(768, device='cuda')
        g, b = torch.rand(768, device='cuda') + 0.5, torch.rand(768, device='cuda')
        x = torch.rand(M, 768, device='cuda')
        out = torch.empty_like(x)
        h = torch.empty(M, 768, device='cuda', dtype=torch.bfloat16)
        cnt = torch.zeros(M // 128 + 8, dtype=torch.int32, device='cuda')
        t_g = timed(lambda: ops.gemm_bias_act(a, w, bias, residual=x, out=out, tile_hint=5))
        t_sep = timed(lambda: ops.gemm_layernorm(a, w, bias, x, g, b, 1e-6, out=out, ln_out=h, tile_hint=5))
        t_fused = timed(lambda: ops.gemm_layernorm(a, w, bias, x, g, b, 1e-6, out=out, ln_out=h, counters=cnt, tile_hint=5))
        print('M=%d K=%d | GEMM %.1f us | GEMM + LayerNorm launches %.1f us | one launch %.1f us' % (M, K, t_g, t_sep, t_fused))


if __name__ == '__main__':
    main()
