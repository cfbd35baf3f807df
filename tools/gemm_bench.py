"""GPU micro-benchmark of vitcap_gemm_bias_act on the hot-path shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L

def bench(M, N, K, act, out_f32, res, hint, iters=30):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.rand(N, device='cuda')
    r = torch.rand(M, N, device='cuda') if res else None
    out = torch.empty(M, N, device='cuda', dtype=torch.float32 if out_f32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=hint)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=hint)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

def check(M, N, K, hint):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.rand(N, device='cuda')
    got = ops.gemm_bias_act(a, w, bias, out_dtype=torch.float32, tile_hint=hint)
    want = a.float() @ w.float().t() + bias
    return float((got - want).abs().max())

if __name__ == '__main__':
    hints = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [2, 3]
    Ms = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [36928]
    for h in hints:
        print('hint', h, 'max err vs torch (M=1000,N=768,K=768):', check(1000, 768, 768, h), check(4099, 2304, 3072, h))
    shapes = [('qkv', 2304, 768, L.ACT_NONE, 0, False), ('proj', 768, 768, L.ACT_NONE, 1, True),
              ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False), ('fc2', 768, 3072, L.ACT_NONE, 1, True),
              ('fc1-nogelu', 3072, 768, L.ACT_NONE, 0, False)]
    for M in Ms:
        for name, N, K, act, of, res in shapes:
            row = []
            for h in hints:
                ms, tf = bench(M, N, K, act, of, res, h)
                row.append('hint%d %.3f ms %.0f TF' % (h, ms, tf))
            print('M=%d %-10s N=%d K=%d : %s' % (M, name, N, K, ' | '.join(row)))
