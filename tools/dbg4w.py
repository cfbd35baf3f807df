import torch, sys
sys.path.insert(0, '/root/repo')
from vitcap_amd import ops
def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale
M, N, K = 9232, 2304, 768
a = _rand((M, K), 81).to(torch.bfloat16).cuda(); w = _rand((N, K), 82, 0.05).to(torch.bfloat16).cuda(); bias = _rand((N,), 83, 0.1).cuda()
ref = ops.gemm_bias_act(a, w, bias, tile_hint=32)
for h in (40, 41, 42):
    o = ops.gemm_bias_act(a, w, bias, tile_hint=h)
    d = (o.float() - ref.float()).abs()
    bad = (d > 0).nonzero()
    print('hint', h, 'max', float(d.max()), 'nbad', len(bad), 'first', bad[:5].tolist(), 'last', bad[-5:].tolist())
    if len(bad):
        rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
        print('  rows', rows[:10].tolist(), '...', rows[-5:].tolist(), len(rows), 'cols', cols[:10].tolist(), len(cols))
