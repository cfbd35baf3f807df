"""Weight-gradient GEMM: gemm_tn (LDS transpose reads, 256x256 tiles) vs transposes + NT split-K (128x128 tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    y = (torch.rand(M, N, device='cuda') - 0.5).to(torch.bfloat16)
    x = (torch.rand(M, K, device='cuda') - 0.5).to(torch.bfloat16)
    tiles = (N // 256) * (K // 256)
    for S in sorted(set([max(1, 256 // tiles), max(1, 512 // tiles), max(1, 768 // tiles)])):
        slabs = torch.empty(S, N, K, device='cuda')
        out = torch.empty(N, K, device='cuda')
        t = timeit(lambda: (ops.gemm_tn(y, x, S, slabs), ops.reduce_slabs(slabs, out)))
        print('M=%d N=%d K=%d  gemm_tn S=%2d + reduce: %7.1f us  %5.0f TF' % (M, N, K, S, t, 2.0 * M * N * K / t / 1e6))
    bias = torch.zeros(N, device='cuda')
    def old():
        yT = ops.transpose_colsum(y, bias)
        xT = ops.transpose_colsum(x)
        t128 = ((N + 127) // 128) * ((K + 127) // 128)
        S = max(1, min(32, yT.shape[1] // 64, (640 + t128 - 1) // t128))
        sl = ops.gemm_ex(yT, xT, split_k=S)
        ops.reduce_slabs(sl, out)
    out = torch.empty(N, K, device='cuda')
    t = timeit(old)
    print('M=%d N=%d K=%d  transposes + NT split-K:      %7.1f us  %5.0f TF' % (M, N, K, t, 2.0 * M * N * K / t / 1e6))
    t = timeit(lambda: ops.colsum_bf16(y, bias))
    print('   colsum %.1f us' % t)
