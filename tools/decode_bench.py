"""GPU micro-benchmark of the decode-step kernels at B=64 (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = 2 * B
def rb(*s, sc=1.0): return ((torch.rand(*s, device='cuda') * 2 - 1) * sc).to(torch.bfloat16)
x = rb(R, 768); w_qkv = rb(2304, 768, sc=0.05); w_fc1 = rb(3072, 768, sc=0.05); w_ao = rb(768, 768, sc=0.05)
w_fc2 = rb(768, 3072, sc=0.05); h = rb(R, 3072); bias = torch.rand(3072, device='cuda')
for name, a, w, act in (('qkv', x, w_qkv, L.ACT_NONE), ('fc1', x, w_fc1, L.ACT_GELU_ERF)):
    for hint in (1, 13, 14, 15, 4):
        us = timeit(lambda: ops.gemm_bias_act(a, w, bias[:w.shape[0]], act=act, tile_hint=hint))
        print('%s hint %d: %.1f us' % (name, hint, us))
    for sk in (2, 3, 6):
        us = timeit(lambda: ops.gemm_bias_act(a, w, None, split_k=sk))
        print('%s split %d (partials only): %.1f us' % (name, sk, us))
for name, a, w, sks in (('ao', x, w_ao, (1, 2, 3, 6)), ('fc2', h, w_fc2, (4, 6, 8, 12, 24))):
    for sk in sks:
        if sk == 1:
            us = timeit(lambda: ops.gemm_bias_act(a, w, bias[:768], out_dtype=torch.float32, tile_hint=4))
        else:
            us = timeit(lambda: ops.gemm_bias_act(a, w, None, split_k=sk))
        print('%s split %d: %.1f us' % (name, sk, us))
g = torch.ones(768, device='cuda'); bt = torch.zeros(768, device='cuda'); res = torch.rand(R, 768, device='cuda')
for S in (6, 12):
    parts = torch.rand(S, R, 768, device='cuda')
    us = timeit(lambda: ops.sum_layernorm(parts, bias[:768], res, g, bt, 1e-12))
    print('sum_layernorm S=%d: %.1f us' % (S, us))
xf = torch.rand(R, 768, device='cuda')
print('layernorm R rows: %.1f us' % timeit(lambda: ops.layernorm(xf, g, bt, 1e-12, True, True)))
vis = rb(B * 578, 2304); step = rb(R, 2304); cache = rb(B, 20, 2, 768)
for t in (1, 10, 19):
    us = timeit(lambda: ops.attn_decode_step(step, vis, cache, B, 578, t))
    mb = B * 12 * (578 + t + 1) * 256 / 1e6
    print('attn_decode t=%d: %.1f us  (%.0f MB -> %.2f TB/s)' % (t, us, mb, mb / us))
wl = rb(30592, 768, sc=0.05); hb = rb(B, 768); bl = torch.rand(30592, device='cuda')
for hint in (1, 4):
    us = timeit(lambda: ops.gemm_bias_act(hb, wl, bl, out_dtype=torch.float32, tile_hint=hint))
    print('lm head hint %d: %.1f us (%.2f TB/s weights)' % (hint, us, 47.0 / us))
logits = torch.randn(B, 30592, device='cuda')
st = ops.greedy_init(B)
print('greedy_step: %.1f us' % timeit(lambda: ops.greedy_step(logits, st, 5)))
