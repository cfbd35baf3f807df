"""Device JPEG back half against its HBM roofline, and the host front half against Pillow's whole decode (GPU box).
    python tools/jpeg_bench.py [batch] > gpurun_out/r06_jpeg_bench.txt

Per image the back half moves (algorithmic): coefficient blocks 2 B x 1.5 samples per pixel in (4:2:0), component planes 1.5 B out and
1.5 B back in, RGB 3 B out = 9 B per pixel; the kernels are byte work with no reuse to speak of, so the bound is HBM (8 TB/s peak,
~6.3 achievable).  Reports us per batch from HIP events over back-to-back launches and the achieved GB/s."""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from vitcap_amd import jpegdec as J
from vitcap_amd.imageio import CoefImage, jpeg_backhalf
from vitcap_amd._lib import JpegImage, check, lib
import ctypes as C


def synth(w, h, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 100 * np.sin(xx / (17 + seed % 13) + c) * np.cos(yy / (23 + seed % 7) - c) for c in range(3)], -1)
    img += g.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sizes = [(640, 480), (480, 640), (500, 375), (640, 427), (1024, 768), (500, 333), (375, 500), (800, 600)]
    datas = []
    for i in range(B):
        w, h = sizes[i % len(sizes)]
        b = io.BytesIO()
        Image.fromarray(synth(w, h, i)).save(b, format='JPEG', quality=90)
        datas.append(b.getvalue())
    # host: front half vs Pillow's whole decode, one thread
    t0 = time.perf_counter()
    for d in datas:
        J.decode_image(d)
    t_pil = (time.perf_counter() - t0) / B
    items = []
    t0 = time.perf_counter()
    for d in datas:
        items.append(CoefImage(*J.decode_coefs(d)))
    t_front = (time.perf_counter() - t0) / B
    px = sum(it.info.width * it.info.height for it in items)
    print('host, one thread, %d JPEGs (mean %.0f KB, %.2f Mpixel): Pillow whole decode %.2f ms / image, front half (parse + Huffman) %.2f ms / image'
          % (B, np.mean([len(d) for d in datas]) / 1024, px / B / 1e6, t_pil * 1e3, t_front * 1e3))
    dev = torch.device('cuda')
    # device buffers once; then the back half alone, back to back
    desc = (JpegImage * B)()
    keep = []
    for i, it in enumerate(items):
        t = torch.from_numpy(it.coefs).to(dev)
        pitch = (3 * it.info.width + 3) & ~3
        o = torch.empty((it.info.height, pitch), dtype=torch.uint8, device=dev)
        desc[i].info = it.info
        desc[i].coefs, desc[i].rgb, desc[i].pitch = t.data_ptr(), o.data_ptr(), pitch
        keep += [t, o]
    need = lib.vitcap_jpeg_backhalf_workspace_bytes(desc, B)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        check(lib.vitcap_jpeg_backhalf(desc, B, C.c_void_p(ws.data_ptr()), need, s), 'jpeg_backhalf')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        check(lib.vitcap_jpeg_backhalf(desc, B, C.c_void_p(ws.data_ptr()), need, s), 'jpeg_backhalf')
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    alg = 9.0 * px
    print('device back half, batch of %d (%.1f Mpixel): %.1f us per batch = %.0f GB/s of the 9 B per pixel it has to move (HBM roofline 8 000, '
          '~6 300 achievable): frac %.3f; %.2f us per image' % (B, px / 1e6, us, alg / us * 1e-3, alg / us * 1e-3 / 8000.0, us / B))


if __name__ == '__main__':
    main()
