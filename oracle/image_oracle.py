"""TEST INFRASTRUCTURE ONLY (see oracle/vitcap_oracle.py header): CPU statement of the reference's test-time image
transform, get_transform_vit_default (src/pipelines/uni_pipeline.py:1233-1256) =
    torchvision.transforms.Resize(int(floor(384 / crop_pct)), PIL.Image.BICUBIC) -> CenterCrop(384) -> ToTensor
    -> Normalize(mean=.5, std=.5)
on a decoded RGB uint8 image (the reference decodes with cv2.imdecode and flips BGR->RGB, transform.py:106-136).

Third-party arithmetic, none of it under /root/reference:
* torchvision (upstream pinned by README to the torch 1.6 era, 0.7.x; NOT installed in this image): only its size rule
  (`functional.resize` with an int: shorter side -> size, longer -> int(size * long / short)) and crop origin
  (`center_crop`: int(round((h - crop) / 2.0)), Python round-half-even) matter; both are restated below.
* Pillow (12.2.0 here, also on the GPU box): `Image.resize(..., BICUBIC)` IS the reference's resize, so the oracle calls
  it directly.  `resample_restated` is a numpy restatement of Pillow's libImaging/Resample.c 8-bit path
  (precompute_coeffs, normalize_coeffs_8bpc, horizontal pass rounding to uint8, then vertical pass); tests pin it
  against Pillow itself and pin the device kernels (and the C-ABI's host-side weight tables) against both.
Parity status: PINNED (against Pillow run in place; torchvision's two integer rules restated from its published source).
"""
import math

import numpy as np
from PIL import Image

PRECISION_BITS = 32 - 8 - 2


def resized_size(h, w, size):
    if w <= h:
        return int(size * h / w), size          # (new_h, new_w)
    return size, int(size * w / h)


def crop_origin(h, w, crop):
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


def transform_reference(img, size=384, crop=384):
    """img: uint8 (H,W,3) RGB.  Returns (cropped uint8 (3,crop,crop), normalised float32 (3,crop,crop))."""
    h, w = img.shape[:2]
    nh, nw = resized_size(h, w, size)
    pil = Image.fromarray(img, 'RGB').resize((nw, nh), Image.BICUBIC)
    top, left = crop_origin(nh, nw, crop)
    u8 = np.asarray(pil)[top:top + crop, left:left + crop]
    chw = np.ascontiguousarray(u8.transpose(2, 0, 1))
    t = chw.astype(np.float32) / np.float32(255.0)             # ToTensor
    return chw, (t - np.float32(0.5)) / np.float32(0.5)       # Normalize


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs_restated(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box -> (ksize, bounds (out,2), kk (out,ksize) int)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, out_size, axis):
    in_size = img.shape[axis]
    _, bounds, kk = coeffs_restated(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        lo, n = bounds[xx]
        acc = (src[lo:lo + n] * kk[xx, :n].reshape((n,) + (1,) * (src.ndim - 1))).sum(0) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resample_restated(img, out_h, out_w):
    """Pillow's two passes on uint8 (H,W,3): horizontal (rounds to uint8), then vertical."""
    tmp = _pass(img, out_w, 1) if out_w != img.shape[1] else img
    return _pass(tmp, out_h, 0) if out_h != img.shape[0] else tmp
