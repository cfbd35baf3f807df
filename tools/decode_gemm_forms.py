"""Decode-step GEMM shapes at M = 2 rows per sequence: the resident whole-K form (tile_hint 20 / 21 / 22) the engine uses against the
4-stage ring forms (hints 1 / 13 / 14 / 15) now that their counted LDS-DMA waits work (docs/LAB_r01_r04.md 4.2 i).  GPU box.
    python tools/decode_gemm_forms.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L


def bench(M, N, K, act, f32, hint, iters=200):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.rand(N, device='cuda') if K == 768 else None
    slabs = K // 768 if hint in (20, 21, 22) and K > 768 else 1
    out = torch.empty((slabs, M, N) if slabs > 1 else (M, N), device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
    fn = lambda: ops.gemm_bias_act(a, w, bias if slabs == 1 else None, act=act if slabs == 1 else L.ACT_NONE, out=out, tile_hint=hint)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in [int(x) for x in sys.argv[1:]] or [128]:
    for name, N, K, act, f32 in (('qkv', 2304, 768, L.ACT_NONE, 0), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0), ('ao', 768, 768, L.ACT_NONE, 1),
                                 ('fc2', 768, 3072, L.ACT_NONE, 1)):
        row = []
        for h in (20, 21, 22, 1, 13, 14, 15, 2, 3, 0):
            try:
                row.append('h%d %.1f' % (h, bench(M, N, K, act, f32, h)))
            except Exception as e:
                row.append('h%d n/a' % h)
        print('M=%d %-3s N=%4d K=%4d us: %s' % (M, name, N, K, ' | '.join(row)), flush=True)
