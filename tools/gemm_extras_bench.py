"""The two training GEMMs whose epilogues carry extras, per launch: fc1 forward (erf-GELU + zout = gelu') and the fc2 input gradient
(aux multiply + column sums), M = 36 928, N = 3072, K = 768, on the 8-wave kernel (tile_hint 5) and on the persistent 4-wave kernel
(tile_hint 42), next to the plain bf16-output launch of the same shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitcap_amd import ops


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for i in range(iters):
        ev[i].record()
        fn()
    ev[iters].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    return ts[len(ts) // 2]


M, N, K = 36928, 3072, 768
g = torch.Generator().manual_seed(3)
x = (torch.rand(M, K, generator=g) - 0.5).to(torch.bfloat16).cuda()
w = ((torch.rand(N, K, generator=g) - 0.5) * 0.1).to(torch.bfloat16).cuda()
bias = torch.zeros(N).cuda()
z = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
aux = (torch.rand(M, N, generator=g) + 0.5).to(torch.bfloat16).cuda()
cs = torch.zeros(N).cuda()
fl = 2.0 * M * N * K
for name, hint in (('8-wave', 5), ('4-wave persistent', 42)):
    t0 = timeit(lambda: ops.gemm_ex(x, w, bias=bias, act=1, tile_hint=hint))
    t1 = timeit(lambda: ops.gemm_ex(x, w, bias=bias, act=1, zout=z, tile_hint=hint))
    t2 = timeit(lambda: ops.gemm_ex(x, w, aux=aux, colsum=cs, tile_hint=hint))
    t3 = timeit(lambda: ops.gemm_ex(x, w, aux=aux, tile_hint=hint))
    t4 = timeit(lambda: ops.gemm_ex(x, w, colsum=cs, tile_hint=hint))
    print('%-18s gelu %.1f us (%.0f TF)  gelu+zout %.1f (%.0f)  aux+colsum %.1f (%.0f)  aux %.1f  colsum %.1f' % (
        name, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, t4), flush=True)
