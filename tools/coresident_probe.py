"""Does the decode step's HBM-bound attention kernel become RESIDENT next to the workgroups of a running 256x256 GEMM, or does it
wait for CUs to drain?  (GPU box.)  Stream A runs a train of large GEMMs (one tile per workgroup, every CU busy); stream B (high
priority, as the engine's decode stream) runs the decode attention kernel N times.  Reported: the attention train's wall time alone,
the GEMM train's alone, and both together -- if the attention launches ride along inside the GEMM's CUs the sum barely grows.
args: [gemm variant: qkv|proj|fc1|fc2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L

which = sys.argv[1] if len(sys.argv) > 1 else 'qkv'
shape = {'qkv': (2304, 768, L.ACT_NONE, False), 'proj': (768, 768, L.ACT_NONE, True), 'fc1': (3072, 768, L.ACT_GELU_ERF, False),
         'fc2': (768, 3072, L.ACT_NONE, True)}[which]
N, K, act, res = shape
M = 36928
a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
bias = torch.rand(N, device='cuda')
r = torch.rand(M, N, device='cuda') if res else None
out = torch.empty(M, N, device='cuda', dtype=torch.float32 if res else torch.bfloat16)
B, S = 64, 578
vis = ((torch.rand(B * S, 2304, device='cuda') * 2 - 1)).to(torch.bfloat16)
step = ((torch.rand(B * 2, 2304, device='cuda') * 2 - 1)).to(torch.bfloat16)
tkv = torch.zeros(B, 20, 2, 768, device='cuda', dtype=torch.bfloat16)
sa = torch.cuda.Stream()
sb = torch.cuda.Stream(priority=-1)
NG, NA = 20, 60


def gemms():
    with torch.cuda.stream(sa):
        for _ in range(NG):
            ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=5)


def attns():
    with torch.cuda.stream(sb):
        for _ in range(NA):
            ops.attn_decode_step(step, vis, tkv, B, S, 5)


def timed(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for f in (gemms, attns):
    f()
torch.cuda.synchronize()
tg = min(timed([gemms]) for _ in range(3))
ta = min(timed([attns]) for _ in range(3))
tb = min(timed([gemms, attns]) for _ in range(3))
print('%s: %d GEMMs alone %.3f ms | %d decode-attention launches alone %.3f ms | together %.3f ms  (sum %.3f; hidden %.0f %% of the attention time)'
      % (which, NG, tg, NA, ta, tb, tg + ta, 100.0 * (tg + ta - tb) / ta))
