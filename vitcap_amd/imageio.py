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
from ._lib import check, lib


def decode_image(data):
    """base64 str/bytes or raw JPEG/PNG bytes -> uint8 (H,W,3) RGB."""
    from PIL import Image
    if isinstance(data, str):
        data = base64.b64decode(data)
    elif isinstance(data, (bytes, bytearray)) and not (data[:2] == b'\xff\xd8' or data[:4] == b'\x89PNG'):
        data = base64.b64decode(data)
    img = Image.open(io.BytesIO(data))
    if img.mode != 'RGB':
        img = img.convert('RGB')         # cv2.IMREAD_COLOR also yields 3 channels for grey / palette / alpha inputs
    return np.asarray(img)


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
        for i, im in enumerate(images):
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
                raise ValueError('image %d: expected uint8 (H,W,3) RGB' % i)
            t = torch.from_numpy(np.array(im, copy=True, order='C')).to(self.dev, non_blocking=True)
            dev_imgs.append(t)
            desc[i] = _ImageDesc(t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3)
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
