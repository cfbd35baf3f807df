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


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {'bicubic': (_bicubic, 2.0), 'bilinear': (_bilinear, 1.0)}


def coeffs_restated(in_size, out_size, filt='bicubic'):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box -> (ksize, bounds (out,2), kk (out,ksize) int)."""
    _bicubic, fsupport = _FILTERS[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
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


def _pass(img, out_size, axis, filt='bicubic'):
    in_size = img.shape[axis]
    _, bounds, kk = coeffs_restated(in_size, out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        lo, n = bounds[xx]
        acc = (src[lo:lo + n] * kk[xx, :n].reshape((n,) + (1,) * (src.ndim - 1))).sum(0) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resample_restated(img, out_h, out_w, filt='bicubic'):
    """Pillow's two passes on uint8 (H,W,3): horizontal (rounds to uint8), then vertical."""
    tmp = _pass(img, out_w, 1, filt) if out_w != img.shape[1] else img
    return _pass(tmp, out_h, 0, filt) if out_h != img.shape[0] else tmp


# ------------------------------------------------------------------------------------------------------------------
# Train-time transform, get_inception_train_transform (src/data_layer/transform.py:52-81) on a decoded RGB image:
#   RandomResizedCrop(384, scale=(small_scale or 0.08, 1), ratio (3/4, 4/3), PIL BILINEAR) -> ColorJitter(0.4, 0.4, 0.4)
#   -> RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5).
# Two halves:
# (1) the random PARAMETERS (crop box, the order and factors of the three jitter operations, the flip) -- drawn by
#     vitcap_amd/augment.py, restated from torchvision 0.7.0 (the release paired with the reference's pytorch==1.6.0;
#     torchvision is not installed here, so the random STREAM is parity-unpinned);
# (2) the deterministic IMAGE ARITHMETIC given those parameters, all of it Pillow's: crop + Image.resize(BILINEAR),
#     ImageEnhance.Brightness / Contrast / Color (= Image.blend with a black / mean-gray / grayscale image),
#     transpose(FLIP_LEFT_RIGHT).  `train_transform_reference` calls Pillow itself; `train_transform_restated` is the
#     numpy restatement of libImaging (Resample.c, Blend.c, Convert.c rgb2l) the device kernel follows.  PINNED to Pillow.
# ------------------------------------------------------------------------------------------------------------------
OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION = 0, 1, 2


def _finish(u8_hwc):
    chw = np.ascontiguousarray(u8_hwc.transpose(2, 0, 1))
    t = chw.astype(np.float32) / np.float32(255.0)
    return chw, (t - np.float32(0.5)) / np.float32(0.5)


def train_transform_reference(img, box, ops, flip, size=384):
    """img uint8 (H,W,3) RGB; box = (top, left, height, width); ops = [(op, factor), ...] in application order; flip bool.
    What torchvision's functional ops do on a PIL image (resized_crop, adjust_brightness/contrast/saturation, hflip)."""
    from PIL import ImageEnhance
    i, j, h, w = box
    pil = Image.fromarray(img, 'RGB').crop((j, i, j + w, i + h)).resize((size, size), Image.BILINEAR)
    for op, f in ops:
        enh = {OP_BRIGHTNESS: ImageEnhance.Brightness, OP_CONTRAST: ImageEnhance.Contrast, OP_SATURATION: ImageEnhance.Color}[op]
        pil = enh(pil).enhance(f)
    if flip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
    return _finish(np.asarray(pil))


def rgb_to_l(u8):
    """Convert.c rgb2l: ITU-R 601-2 luma in 16.16 fixed point."""
    x = u8.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_restated(in1, in2, alpha):
    """Blend.c ImagingBlend(in1, in2, float alpha): float32 arithmetic, truncation to uint8, clipping only when
    extrapolating (alpha outside [0, 1])."""
    a = np.float32(alpha)
    if a == 0.0:
        return in1.copy()
    if a == 1.0:
        return in2.copy()
    d = in2.astype(np.int32) - in1.astype(np.int32)
    t = in1.astype(np.float32) + a * d.astype(np.float32)           # float32 product, float32 sum
    if 0.0 <= a <= 1.0:
        return t.astype(np.uint8)                                   # (UINT8) cast of an in-range float: truncation
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t)).astype(np.uint8)


def enhance_restated(u8, op, factor):
    if op == OP_BRIGHTNESS:
        deg = np.zeros_like(u8)
    elif op == OP_CONTRAST:
        L = rgb_to_l(u8)
        mean = int(float(L.astype(np.int64).sum()) / L.size + 0.5)   # ImageStat mean (sum / count in doubles), int(mean + .5)
        deg = np.full_like(u8, mean)
    else:
        deg = np.repeat(rgb_to_l(u8)[..., None], 3, axis=2)
    return blend_restated(deg, u8, factor)


def train_transform_restated(img, box, ops, flip, size=384):
    i, j, h, w = box
    u8 = resample_restated(np.ascontiguousarray(img[i:i + h, j:j + w]), size, size, 'bilinear')
    for op, f in ops:
        u8 = enhance_restated(u8, op, f)
    if flip:
        u8 = u8[:, ::-1]
    return _finish(u8)
