"""Does the row pitch of A / W / C matter for the persistent 256x256 GEMM?  (HBM channel interleave probe, GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L
import ctypes as C
from vitcap_amd._lib import lib, check

def call(a, w, bias, r, out, act):
    M, K = a.shape
    N = w.shape[0]
    d = L.GemmDesc(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(0), ldr=r.stride(0) if r is not None else 0, act=act,
                   out_dtype=L.OUT_F32 if out.dtype == torch.float32 else L.OUT_BF16, row_group=0, out_group_rows=0, out_row_off=0,
                   res_periodic=0, tile_hint=12, split_k=0)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(lib.vitcap_gemm_bias_act(p(a), p(w), p(bias), p(r), p(out), C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'gemm')

def bench(M, N, K, act, f32, res, pa, pw, pc, iters=30):
    a = (torch.rand(M, K + pa, device='cuda') * 2 - 1).to(torch.bfloat16)[:, :K]
    w = ((torch.rand(N, K + pw, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)[:, :K]
    bias = torch.rand(N, device='cuda')
    r = torch.rand(M, N + pc, device='cuda')[:, :N] if res else None
    out = torch.empty(M, N + pc, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)[:, :N]
    for _ in range(3):
        call(a, w, bias, r, out, act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call(a, w, bias, r, out, act)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

M = 36928
for name, N, K, act, f32, res in (('qkv', 2304, 768, L.ACT_NONE, 0, False), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False),
                                  ('proj', 768, 768, L.ACT_NONE, 1, True), ('fc2', 768, 3072, L.ACT_NONE, 1, True)):
    row = []
    for pa, pw, pc in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (0, 0, 64), (64, 64, 64), (0, 0, 32), (128, 128, 128)):
        row.append('pad(A,W,C)=(%d,%d,%d) %.1f' % (pa, pw, pc, min(bench(M, N, K, act, f32, res, pa, pw, pc) for _ in range(2))))
    print('%-5s us: %s' % (name, ' | '.join(row)), flush=True)
