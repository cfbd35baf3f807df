"""Input side of the captioning path: (key, base64 JPEG) rows -> decoded RGB bytes -> the reference's
test transform on the GPU -> the (B,3,384,384) batch `ImageCaptioning.forward` takes.

JPEG entropy decoding stays on the host (Pillow's libjpeg; the reference uses cv2.imdecode + BGR2RGB,
src/tools/common.py:23-31, src/data_layer/transform.py:106-136 -- both wrap libjpeg(-turbo)'s default ISLOW IDCT).
Everything after the decode -- bicubic resize, centre crop, /255, normalise, layout, dtype -- runs in
csrc/preproc.hip, bit-identical to torchvision-on-Pillow."""
import base64
import ctypes as C
import io

import numpy as np
import torch

from ._lib import Image as _ImageDesc
from ._lib import TrainAug as _TrainAug
from ._lib import check, lib


from .jpegdec import decode_image      # noqa: E402,F401  (kept importable from here; the loader's worker processes import jpegdec alone)
from ._lib import JpegImage as _JpegImage


class CoefImage(object):
    """A JPEG the host has only ENTROPY-decoded (vitcap_amd/jpegdec.py decode_coefs / the loader's workers): vitcap_jpeg_info + int16
    coefficient blocks.  ImagePreprocessor finishes the decode on the GPU (csrc/jpeg.hip: dequantisation, inverse DCT, chroma
    upsampling, YCbCr -> RGB -- Pillow's pixels bit for bit) in front of its resize."""
    __slots__ = ('info', 'coefs')

    def __init__(self, info, coefs):
        self.info, self.coefs = info, coefs

    @property
    def shape(self):
        return (int(self.info.height), int(self.info.width), 3)


def jpeg_backhalf(items, dev, keep):
    """[CoefImage] -> list of uint8 (H, pitch) device tensors holding the RGB rows (pitch = 3 W rounded up to 4 bytes; enqueued on the
    current stream).  `keep` collects every buffer that must outlive the kernels.
    Host work per batch is kept small (the loader's producer thread shares the GIL with the caption launcher): coefficient arrays that sit
    back to back in host memory -- the images of one worker task inside its shared-memory slab -- travel in ONE host -> device copy, and
    the RGB images are slices of one allocation."""
    B = len(items)
    desc = (_JpegImage * B)()
    # ---- host -> device: runs of adjacent arrays (gaps < 64 B: the slab's 16-byte alignment padding) as single copies
    addr = [it.coefs.ctypes.data for it in items]
    nbyt = [it.coefs.nbytes for it in items]
    dptr = [0] * B
    i = 0
    while i < B:
        j = i
        ok = items[i].coefs.flags.c_contiguous
        while ok and j + 1 < B and items[j + 1].coefs.flags.c_contiguous and 0 <= addr[j + 1] - (addr[j] + nbyt[j]) < 64:
            j += 1
        if not ok:
            src = np.array(items[i].coefs, copy=True, order='C').view(np.uint8).reshape(-1)
        elif j == i and items[i].coefs.flags.writeable:
            src = items[i].coefs.view(np.uint8).reshape(-1)
        else:       # one flat view over the whole run (the memory belongs to the items' base buffer, which the caller keeps alive)
            span = addr[j] + nbyt[j] - addr[i]
            src = np.ctypeslib.as_array((C.c_uint8 * span).from_address(addr[i]))
            keep.append([it.coefs for it in items[i:j + 1]])
        t = torch.from_numpy(src).to(dev, non_blocking=True)
        keep.append(t)
        base = t.data_ptr()
        for k in range(i, j + 1):
            dptr[k] = base + (addr[k] - addr[i])
        i = j + 1
    # ---- outputs: one allocation, 256-byte aligned slices
    geo, total = [], 0
    for it in items:
        h, w = int(it.info.height), int(it.info.width)
        pitch = (3 * w + 3) & ~3                  # rows start on a dword: the colour kernel stores 12 bytes as three dwords
        geo.append((total, h, pitch))
        total += (h * pitch + 255) & ~255
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    keep.append(buf)
    outs = []
    for k, (it, (off, h, pitch)) in enumerate(zip(items, geo)):
        o = buf[off:off + h * pitch].view(h, pitch)
        desc[k].info = it.info
        desc[k].coefs, desc[k].rgb, desc[k].pitch = dptr[k], o.data_ptr(), pitch
        outs.append(o)
    need = lib.vitcap_jpeg_backhalf_workspace_bytes(desc, B)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    keep.append(ws)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.vitcap_jpeg_backhalf(desc, B, C.c_void_p(ws.data_ptr()), need, s), 'jpeg_backhalf')
    return outs


def _device_images(images, dev, keep, desc):
    """Fills the vitcap_image descriptors `desc` for a list of decoded arrays and / or CoefImage items: arrays are copied to the device,
    entropy-decoded JPEGs are finished there first (one back-half launch for all of them).  `keep` collects the buffers that must
    outlive the enqueued kernels."""
    coef_idx = [i for i, im in enumerate(images) if isinstance(im, CoefImage)]
    decoded = dict(zip(coef_idx, jpeg_backhalf([images[i] for i in coef_idx], dev, keep))) if coef_idx else {}
    for i, im in enumerate(images):
        if i in decoded:
            t = decoded[i]
            keep.append(t)
            desc[i] = _ImageDesc(t.data_ptr(), int(im.info.height), int(im.info.width), t.shape[1])
            continue
        if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
            raise ValueError('image %d: expected uint8 (H,W,3) RGB or a CoefImage' % i)
        # Pillow hands out read-only arrays (torch wants a writable one: private copy); a writable C-contiguous view -- the
        # loader's shared-memory slabs -- goes to the device as it is
        src = im if (im.flags.writeable and im.flags.c_contiguous) else np.array(im, copy=True, order='C')
        t = torch.from_numpy(src).to(dev, non_blocking=True)
        keep.append(t)
        desc[i] = _ImageDesc(t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3)


class ImagePreprocessor(object):
    """get_transform_vit_default(is_train=False) (uni_pipeline.py:1233-1256) for a list of decoded images."""

    def __init__(self, device='cuda', test_crop_size=384, crop_pct=1.0, out_dtype=torch.bfloat16):
        import math
        self.dev = torch.device(device)
        self.crop = int(test_crop_size)
        self.resize_short = int(math.floor(test_crop_size / crop_pct))
        assert out_dtype in (torch.bfloat16, torch.float32)
        self.out_dtype = out_dtype
        self._ws = None

    def __call__(self, images, want_u8=False):
        B = len(images)
        dev_imgs, desc = [], (_ImageDesc * B)()
        _device_images(images, self.dev, dev_imgs, desc)
        need = lib.vitcap_image_preproc_workspace_bytes(desc, B, self.resize_short, self.crop)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        out = torch.empty((B, 3, self.crop, self.crop), dtype=self.out_dtype, device=self.dev)
        u8 = torch.empty((B, 3, self.crop, self.crop), dtype=torch.uint8, device=self.dev) if want_u8 else None
        s = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        check(lib.vitcap_image_preproc(desc, B, self.resize_short, self.crop, int(self.out_dtype == torch.bfloat16),
                                       C.c_void_p(out.data_ptr()), C.c_void_p(u8.data_ptr()) if want_u8 else None,
                                       C.c_void_p(self._ws.data_ptr()), need, s), 'image_preproc')
        self._keep = dev_imgs             # the raw images must outlive the enqueued kernels
        return (out, u8) if want_u8 else out


class TrainImagePreprocessor(object):
    """get_transform_vit_default(is_train=True) (uni_pipeline.py:1258-1264) = get_inception_train_transform for a list of
    decoded images and their drawn parameters (vitcap_amd/augment.py): crop + bilinear resize, colour jitter, flip,
    ToTensor, Normalize on the GPU (csrc/preproc.hip), bit-identical to torchvision-on-Pillow for the same parameters."""

    def __init__(self, device='cuda', train_crop_size=384, out_dtype=torch.bfloat16):
        self.dev = torch.device(device)
        self.size = int(train_crop_size)
        assert out_dtype in (torch.bfloat16, torch.float32)
        self.out_dtype = out_dtype
        self._ws = None

    def __call__(self, images, params, want_u8=False):
        B = len(images)
        assert len(params) == B
        dev_imgs, desc, aug = [], (_ImageDesc * B)(), (_TrainAug * B)()
        _device_images(images, self.dev, dev_imgs, desc)
        for i, (im, pr) in enumerate(zip(images, params)):
            top, left, h, w = pr['box']
            a = aug[i]
            a.top, a.left, a.height, a.width, a.flip = int(top), int(left), int(h), int(w), int(bool(pr['flip']))
            ops = list(pr['ops'])
            if len(ops) > 3:
                raise ValueError('image %d: at most three colour operations' % i)
            for o in range(3):
                a.op[o], a.factor[o] = (int(ops[o][0]), float(ops[o][1])) if o < len(ops) else (-1, 1.0)
        need = lib.vitcap_image_train_preproc_workspace_bytes(desc, aug, B, self.size)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        out = torch.empty((B, 3, self.size, self.size), dtype=self.out_dtype, device=self.dev)
        u8 = torch.empty((B, 3, self.size, self.size), dtype=torch.uint8, device=self.dev) if want_u8 else None
        s = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        check(lib.vitcap_image_train_preproc(desc, aug, B, self.size, int(self.out_dtype == torch.bfloat16),
                                             C.c_void_p(out.data_ptr()), C.c_void_p(u8.data_ptr()) if want_u8 else None,
                                             C.c_void_p(self._ws.data_ptr()), need, s), 'image_train_preproc')
        self._keep = dev_imgs
        return (out, u8) if want_u8 else out
