"""Per-tile s_memtime stamps of the persistent 256x256 GEMM: main loop / epilogue / total per tile slot (GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import ops, _lib as L
lib = L.lib
lib.vitcap_gemm_set_trace.argtypes = [ctypes.c_void_p]
lib.vitcap_gemm_set_trace.restype = None
M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
HINT = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for name, N, K, act, f32, res in (('qkv', 2304, 768, L.ACT_NONE, 0, False), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False), ('fc1-nogelu', 3072, 768, L.ACT_NONE, 0, False), ('qkv-gelu', 2304, 768, L.ACT_GELU_ERF, 0, False),
                                  ('proj', 768, 768, L.ACT_NONE, 1, True), ('fc2', 768, 3072, L.ACT_NONE, 1, True)):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.rand(N, device='cuda')
    r = torch.rand(M, N, device='cuda') if res else None
    out = torch.empty(M, N, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
    for _ in range(5):
        ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=HINT)
    tr = torch.zeros(256 * 2 * 8 * 4, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    lib.vitcap_gemm_set_trace(ctypes.c_void_p(tr.data_ptr()))
    for _ in range(5):
        ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=HINT)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_bias_act(a, w, bias, residual=r, act=act, out=out, tile_hint=HINT)
    e1.record()
    torch.cuda.synchronize()
    lib.vitcap_gemm_set_trace(ctypes.c_void_p(0))
    t = tr.cpu().view(256, 2, 8, 4)
    us = e0.elapsed_time(e1) * 1e3
    g0 = t[:, 0]
    valid = g0[:, :, 2] > 0
    nt = valid.sum(1)
    last = torch.stack([g0[i, int(nt[i]) - 1, 2] for i in range(256)])
    span = (last - g0[:, 0, 0]).double()
    tpu = float(span.max()) / us                    # ticks per us, assuming the longest workgroup spans the kernel
    xcd = torch.arange(256) % 8
    print('%s M=%d N=%d K=%d: kernel %.1f us by events; longest WG %.0f ticks -> %.0f ticks/us (upper bound)' % (name, M, N, K, us, float(span.max()), tpu))
    for s_ in range(8):
        v = valid[:, s_]
        if not v.any():
            break
        main = (g0[:, s_, 1] - g0[:, s_, 0])[v].double() / tpu
        epi = (g0[:, s_, 2] - g0[:, s_, 1])[v].double() / tpu
        st = torch.zeros(256, dtype=torch.float64)
        for x in range(8):
            sel = xcd == x
            st[sel] = (g0[sel, s_, 0] - g0[sel, 0, 0].min()).double() / tpu
        st = st[v]
        print('  slot %d: %3d WGs  start %.1f..%.1f  main loop %.2f (%.2f..%.2f)  epilogue %.2f (%.2f..%.2f) us' % (
            s_, int(v.sum()), float(st.min()), float(st.max()), float(main.mean()), float(main.min()), float(main.max()),
            float(epi.mean()), float(epi.min()), float(epi.max())))
