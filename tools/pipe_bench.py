"""Batch-level software pipeline: encode+prefill of batch i+1 on one stream while batch i decodes on another."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import weights as W
from vitcap_amd._lib import lib, check
from vitcap_amd.model import ImageCaptioning

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = ImageCaptioning().load_recipe(0).eval()
m.pack('cuda')
img = torch.from_numpy(W.gen_image_batch(B, 1)).cuda().to(torch.bfloat16)
slots = [m._workspace(B, m._packed[2], slot=i) for i in range(2)]
ids = [torch.empty(B, 20, dtype=torch.int64, device='cuda') for _ in range(2)]
lp = [torch.empty(B, dtype=torch.float32, device='cuda') for _ in range(2)]
p = lambda t: C.c_void_p(t.data_ptr())
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def run(K, pipelined):
    evA = [torch.cuda.Event() for _ in range(K)]
    evB = [torch.cuda.Event() for _ in range(K)]
    for i in range(K):
        ws, need = slots[i % 2]
        a = sA if pipelined else torch.cuda.current_stream()
        b = sB if pipelined else torch.cuda.current_stream()
        with torch.cuda.stream(a):
            if pipelined and i >= 2:
                a.wait_event(evB[i - 2])            # slot free again
            h = C.c_void_p(a.cuda_stream)
            check(lib.vitcap_engine_encode(m._engine, p(img), 1, B, None, p(ws), need, h), 'enc')
            check(lib.vitcap_engine_prefill(m._engine, B, None, p(ws), need, h), 'pre')
            evA[i].record(a)
        with torch.cuda.stream(b):
            if pipelined:
                b.wait_event(evA[i])
            h = C.c_void_p(b.cuda_stream)
            check(lib.vitcap_engine_decode(m._engine, B, None, p(ws), need, p(ids[i % 2]), p(lp[i % 2]), None, h), 'dec')
            evB[i].record(b)

for mode in (False, True, False, True):
    run(4, mode)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    run(K, mode)
    torch.cuda.synchronize()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / K
    print('B=%d %s: %.3f ms per batch, %.0f img/s' % (B, 'pipelined (2 streams)' if mode else 'sequential', ms, B / ms * 1e3))
m.generate(img)
ref = m.generate(img)[0][:, 0]
torch.cuda.synchronize()
print('ids equal to generate():', bool(torch.equal(ids[0], ref)), bool(torch.equal(ids[1], ref)))
