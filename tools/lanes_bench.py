"""N independent encode||decode pipelines ("lanes"), each fed a 1/N slice of every 64-image batch (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import weights as W
from vitcap_amd.model import ImageCaptioning

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = ImageCaptioning().load_recipe(0).eval()
m.pack('cuda')
img = torch.from_numpy(W.gen_image_batch(B, 1)).cuda().to(torch.bfloat16)
ref_ids, _ = m.generate(img)
ref_ids = ref_ids.clone()
for lanes in (1, 2, 4, 1, 2):
    parts = [c.contiguous() for c in img.chunk(lanes, 0)]
    def run(K):
        pend = []
        for i in range(K):
            cur = [m.generate_async(parts[l], lane=l) for l in range(lanes)]
            pend.append(cur)
            if len(pend) > 1:
                for h in pend.pop(0):
                    h.result()
        out = None
        for cur in pend:
            out = [h.result() for h in cur]
        return out
    out = run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(K)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / K
    ids = torch.cat([o[0] for o in out], 0)
    print('B=%d lanes=%d: %.3f ms per batch, %.0f img/s, ids equal to one-stream generate: %s' % (B, lanes, ms, B / ms * 1e3, bool(torch.equal(ids, ref_ids))), flush=True)
