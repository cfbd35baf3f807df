"""decode_streams = 2 timing probe (GPU box): args: n_iters, piped_first (0/1)"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from vitcap_amd.model import ImageCaptioning
from vitcap_amd import weights as W, _lib as L
n = int(sys.argv[1]); piped_first = int(sys.argv[2])
m = ImageCaptioning().load_recipe(0).eval(); m.pack('cuda')
img = torch.from_numpy(W.gen_image_batch(64, 1234)).cuda().to(torch.bfloat16).contiguous()
if piped_first:
    po = m.gen_options(gemm_mode=L.GEMM_TILES)
    m.prime_pipeline(64, img.device, opts=po)
    for _ in range(3): m.generate_async(img, opts=po).result()
for ds in (1, 2, 1, 2):
    opts = m.gen_options(decode_streams=ds)
    for _ in range(3): m.run(img, opts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.run(img, opts)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('n=%d piped_first=%d decode_streams=%d: %.2f ms/batch' % (n, piped_first, ds, (t2 - t0) / n * 1e3), flush=True)
