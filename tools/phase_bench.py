"""Time the three engine phases (encode / prefill / decode) separately, and decode on two streams (two half batches)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import weights as W
from vitcap_amd._lib import lib, check
from vitcap_amd.model import ImageCaptioning

def ev_time(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

m = ImageCaptioning().load_recipe(0).eval()
m.pack('cuda')
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '32,64').split(',')]:
    img = torch.from_numpy(W.gen_image_batch(B, 1)).cuda().to(torch.bfloat16)
    ws, need = m._workspace(B, m._packed[2])
    ids = torch.empty(B, 20, dtype=torch.int64, device='cuda'); lp = torch.empty(B, dtype=torch.float32, device='cuda')
    s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    enc = lambda: check(lib.vitcap_engine_encode(m._engine, p(img), 1, B, None, p(ws), need, s()), 'enc')
    pre = lambda: check(lib.vitcap_engine_prefill(m._engine, B, None, p(ws), need, s()), 'pre')
    dec = lambda: check(lib.vitcap_engine_decode(m._engine, B, None, p(ws), need, p(ids), p(lp), None, s()), 'dec')
    enc(); pre(); dec()
    print('B=%d encode %.3f ms  prefill %.3f ms  decode %.3f ms' % (B, ev_time(enc), ev_time(pre), ev_time(dec)))
