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


# ---- front half only (round 6): marker parsing + Huffman decoding on the host (libvitcap_jpeg.so, include/vitcap_jpeg.h); dequantisation,
# inverse DCT, chroma upsampling and the colour conversion run on the GPU (csrc/jpeg.hip), bit-identical to Pillow's decoder
import ctypes as _C
import os as _os


class JpegInfo(_C.Structure):
    """vitcap_jpeg_info (include/vitcap_jpeg.h)."""
    _fields_ = [('abi', _C.c_int32), ('width', _C.c_int32), ('height', _C.c_int32), ('ncomp', _C.c_int32),
                ('hs', _C.c_int32 * 3), ('vs', _C.c_int32 * 3), ('blocks_w', _C.c_int32 * 3), ('blocks_h', _C.c_int32 * 3),
                ('samp_w', _C.c_int32 * 3), ('samp_h', _C.c_int32 * 3), ('block0', _C.c_int32 * 3), ('nblocks', _C.c_int32),
                ('qt', (_C.c_uint16 * 64) * 3)]


JPEG_ABI = 1
JPEG_OK, JPEG_EINVAL, JPEG_EUNSUPPORTED = 0, 1, 2
_jpeg_lib = None


def jpeg_lib():
    """The host front half, or None when libvitcap_jpeg.so has not been built (callers then decode with Pillow)."""
    global _jpeg_lib
    if _jpeg_lib is None:
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'libvitcap_jpeg.so')
        if not _os.path.exists(path):
            _jpeg_lib = False
            return None
        lib = _C.CDLL(path)
        lib.vitcap_jpeg_abi.restype = _C.c_int
        if lib.vitcap_jpeg_abi() != JPEG_ABI:
            raise RuntimeError('libvitcap_jpeg.so has ABI %d, this module expects %d: rebuild (make -C vitcap_amd/csrc)' % (lib.vitcap_jpeg_abi(), JPEG_ABI))
        lib.vitcap_jpeg_parse.argtypes = [_C.c_void_p, _C.c_size_t, _C.POINTER(JpegInfo)]
        lib.vitcap_jpeg_parse.restype = _C.c_int
        lib.vitcap_jpeg_decode_coefs.argtypes = [_C.c_void_p, _C.c_size_t, _C.POINTER(JpegInfo), _C.c_void_p]
        lib.vitcap_jpeg_decode_coefs.restype = _C.c_int
        lib.vitcap_jpeg_last_error.restype = _C.c_char_p
        _jpeg_lib = lib
    return _jpeg_lib or None


def _jpeg_bytes(data):
    if isinstance(data, str):
        data = base64.b64decode(data)
    elif isinstance(data, (bytes, bytearray)) and not (data[:2] == b'\xff\xd8' or data[:4] == b'\x89PNG'):
        data = base64.b64decode(data)
    return bytes(data)


def jpeg_parse(data):
    """-> (JpegInfo, raw bytes) for a baseline JPEG inside the supported subset, else (None, raw bytes)."""
    lib = jpeg_lib()
    raw = _jpeg_bytes(data)
    if lib is None or raw[:2] != b'\xff\xd8':
        return None, raw
    info = JpegInfo()
    rc = lib.vitcap_jpeg_parse(raw, len(raw), _C.byref(info))
    return (info if rc == JPEG_OK else None), raw


def jpeg_coefs_into(raw, info, out):
    """Huffman-decodes `raw` into the int16 array `out` (info.nblocks * 64 elements, C-contiguous, writable).  False: corrupt stream."""
    assert out.dtype == np.int16 and out.size >= info.nblocks * 64 and out.flags.c_contiguous
    return jpeg_lib().vitcap_jpeg_decode_coefs(raw, len(raw), _C.byref(info), out.ctypes.data_as(_C.c_void_p)) == JPEG_OK


def decode_coefs(data):
    """-> (JpegInfo, int16 coefficients [nblocks * 64]) or None when the stream must go through Pillow (decode_image)."""
    info, raw = jpeg_parse(data)
    if info is None:
        return None
    out = np.empty(info.nblocks * 64, np.int16)
    return (info, out) if jpeg_coefs_into(raw, info, out) else None


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


def decode_into(shm_name, blobs, device_jpeg=False):
    """Decode `blobs` into the slab `shm_name` back to back (16-byte aligned).  Returns per image (offset, h, w), or the array
    itself for an image that no longer fits the slab (the parent then gets it through the pipe, as decode_many would send it).
    device_jpeg: a baseline JPEG is only ENTROPY-decoded here -- its int16 coefficient blocks go into the slab and the item is
    ('coef', offset, info bytes) (the GPU finishes the decode, csrc/jpeg.hip); anything else takes the Pillow path as before."""
    shm = _attach(shm_name)
    buf = np.frombuffer(shm.buf, dtype=np.uint8)
    out, off = [], 0
    for b in blobs:
        if device_jpeg:
            info, raw = jpeg_parse(b)
            if info is not None:
                n = info.nblocks * 128
                if off + n <= buf.size and jpeg_coefs_into(raw, info, buf[off:off + n].view(np.int16)):
                    out.append(('coef', off, bytes(info)))
                    off = (off + n + 15) & ~15
                    continue
            b = raw
        im = decode_image(b)
        n = im.size
        if off + n <= buf.size:
            buf[off:off + n] = im.reshape(-1)
            out.append((off, im.shape[0], im.shape[1]))
            off = (off + n + 15) & ~15
        else:
            out.append(im)
    return out


def coef_item(item, slab_buf):
    """('coef', offset, info bytes) of decode_into -> (JpegInfo, int16 view of the slab)."""
    info = JpegInfo.from_buffer_copy(item[2])
    return info, np.ndarray((info.nblocks * 64,), dtype=np.int16, buffer=slab_buf, offset=item[1])


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


def decode_rows_into(shm_name, tsv_path, row_ids, device_jpeg=False):
    """-> (keys, items): items as decode_into returns them."""
    recs = _rows(tsv_path, row_ids)
    return [r[0] for r in recs], decode_into(shm_name, [r[-1] for r in recs], device_jpeg)


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
