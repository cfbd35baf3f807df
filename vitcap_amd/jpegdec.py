"""Host-side image decode for loader worker PROCESSES: this module imports nothing but the standard library, numpy and Pillow and never
touches the GPU runtime (vitcap_amd.imageio imports torch and the HIP library).  Note that a SPAWNED worker re-imports the parent's
`__main__` module as well (run.py / bench.py import torch: ~1.5 s and ~300 MB per worker at start-up, nothing per task).

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


# ---- rows read BY THE WORKER (round 5): the parent used to read every TSV row and pickle its base64 blob (180 KB per image, 0.5 GB/s at
# 3 000 images/s) into the task -- one Python thread that also shares the GIL with the caption launcher.  A task now names the TSV file
# and the row numbers; the worker seeks, reads, base64-decodes and JPEG-decodes, and the parent handles slab offsets only.
_TSV = {}


def _rows(tsv_path, row_ids):
    import threading
    from .tsv import TSVFile                 # os / os.path only
    key = (tsv_path, threading.get_ident())  # `loader_threads: true` runs these tasks in threads of one process: one handle each
    t = _TSV.get(key)
    if t is None:
        t = _TSV[key] = TSVFile(tsv_path)
    return [t[i] for i in row_ids]


def decode_rows_into(shm_name, tsv_path, row_ids):
    """-> (keys, items): items as decode_into returns them."""
    recs = _rows(tsv_path, row_ids)
    return [r[0] for r in recs], decode_into(shm_name, [r[-1] for r in recs])


def decode_rows(tsv_path, row_ids):
    """-> (keys, images): the pipe-return fallback (no shared memory)."""
    recs = _rows(tsv_path, row_ids)
    return [r[0] for r in recs], decode_many([r[-1] for r in recs])


def pin_worker(counter, first, stride):
    """Pool initializer (VITCAP_LOADER_CPUS): this worker takes the next CPU of the sequence first, first + stride, ..."""
    import os
    with counter.get_lock():
        i = counter.value
        counter.value += 1
    try:
        os.sched_setaffinity(0, {first + i * stride})
    except OSError:
        pass
