"""Host-side image decode for loader worker PROCESSES: imports nothing but the standard library, numpy and Pillow, so that a spawned
worker starts in a fraction of a second and never touches the GPU runtime (vitcap_amd.imageio imports torch and the HIP library).

The reference decodes inside DataLoader worker processes (uni_pipeline.py:333-339, transform.py:106-136: cv2.imdecode); threads of
one Python process top out near 2 000 images/s on the GPU box however many are started (base64 and the numpy copy hold the GIL:
600 images/s with one thread, 2 040 with 8, 1 470 with 64 -- tools/input_side_bench.py), one GPU captions 3 700."""
import base64
import io

import numpy as np


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


def decode_many(blobs):
    """One task of a worker process: a handful of images per call amortises the task hand-off."""
    return [decode_image(b) for b in blobs]


# ---- decode into shared memory: the decoded pixels of an image are ~1 MB; returned through the pool's result pipe they cap a loader at
# ~2 000 images/s whatever the number of workers (the parent unpickles 1.8 GB/s single-threaded: measured, 8 / 16 / 32 processes all
# 1 700-2 200 images/s).  A worker therefore writes the pixels into a shared-memory slab the parent named in the task and returns
# only (offset, height, width) per image.
_SHM = {}


def _attach(name):
    from multiprocessing import shared_memory
    shm = _SHM.get(name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=name)
        _SHM[name] = shm
    return shm


def decode_into(shm_name, blobs):
    """Decode `blobs` into the slab `shm_name` back to back (16-byte aligned).  Returns per image (offset, h, w), or the array
    itself for an image that no longer fits the slab (the parent then gets it through the pipe, as decode_many would send it)."""
    shm = _attach(shm_name)
    buf = np.frombuffer(shm.buf, dtype=np.uint8)
    out, off = [], 0
    for b in blobs:
        im = decode_image(b)
        n = im.size
        if off + n <= buf.size:
            buf[off:off + n] = im.reshape(-1)
            out.append((off, im.shape[0], im.shape[1]))
            off = (off + n + 15) & ~15
        else:
            out.append(im)
    return out
