"""Device JPEG back half, CPU side (no GPU): the host entropy decoder (vitcap_amd/libvitcap_jpeg.so, include/vitcap_jpeg.h) followed by the
numpy restatement of libjpeg-turbo's back half (oracle/jpeg_backhalf.py) must give Pillow's decoded pixels BIT FOR BIT -- that pins the
oracle the GPU kernels (csrc/jpeg.hip, tests/test_hip_jpeg.py) are compared with -- and streams outside the supported subset must be
refused, not mis-decoded."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_backhalf as O
from vitcap_amd import jpegdec as J

SIZES = [(384, 384), (640, 480), (480, 640), (500, 375), (333, 500), (17, 23), (8, 8), (1024, 683), (97, 211)]     # (W, H)


def synth(w, h, seed):
    """Photo-like content: smooth gradients + texture + hard edges (every AC band and the clamps get exercised)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / (7.0 + c) + seed) * np.cos(yy / (11.0 - c)) for c in range(3)], -1)
    img += rng.randn(h, w, 3) * 25
    img[h // 3:h // 2, w // 4:w // 2] = rng.randint(0, 2, 3) * 255
    return np.clip(img, 0, 255).astype(np.uint8)


def jpeg_bytes(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format='JPEG', **kw)
    return b.getvalue()


def pillow_rgb(data):
    im = Image.open(io.BytesIO(data))
    return np.asarray(im.convert('RGB') if im.mode != 'RGB' else im)


needs_lib = pytest.mark.skipif(J.jpeg_lib() is None, reason='libvitcap_jpeg.so not built')


@needs_lib
@pytest.mark.parametrize('sub', [0, 1, 2], ids=['444', '422', '420'])
@pytest.mark.parametrize('wh', SIZES, ids=lambda s: '%dx%d' % s)
def test_front_half_plus_oracle_back_half_equals_pillow(wh, sub):
    w, h = wh
    for q, seed in ((90, 1), (50, 2), (20, 3)):
        data = jpeg_bytes(synth(w, h, seed), quality=q, subsampling=sub)
        got = J.decode_coefs(data)
        if got is None:                      # chroma planes of <= 2 samples are outside the subset (jdsample.c replicates them)
            assert sub != 0 and (w + 1) // 2 <= 2
            continue
        info, coefs = got
        assert (info.width, info.height, info.ncomp) == (w, h, 3)
        np.testing.assert_array_equal(O.backhalf(info, coefs), pillow_rgb(data))


@needs_lib
def test_grey_optimised_tables_and_restart_markers():
    arr = synth(321, 203, 7)
    grey = jpeg_bytes(arr[:, :, 0], quality=80)
    info, coefs = J.decode_coefs(grey)
    assert info.ncomp == 1
    np.testing.assert_array_equal(O.backhalf(info, coefs), pillow_rgb(grey))
    opt = jpeg_bytes(arr, quality=85, optimize=True)                 # per-image Huffman tables
    info, coefs = J.decode_coefs(opt)
    np.testing.assert_array_equal(O.backhalf(info, coefs), pillow_rgb(opt))
    seen_rst = False
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
        try:
            rst = jpeg_bytes(arr, quality=75, **kw)
        except TypeError:
            continue                                                  # this Pillow cannot write restart markers
        if b'\xff\xdd' not in rst:
            continue                                                  # option silently ignored by this Pillow
        seen_rst = True
        info, coefs = J.decode_coefs(rst)
        np.testing.assert_array_equal(O.backhalf(info, coefs), pillow_rgb(rst))
    print('restart-marker streams tested:', seen_rst)


@needs_lib
def test_unsupported_streams_are_refused():
    arr = synth(120, 90, 5)
    assert J.decode_coefs(jpeg_bytes(arr, quality=80, progressive=True)) is None
    cmyk = io.BytesIO()
    Image.fromarray(arr).convert('CMYK').save(cmyk, format='JPEG')
    assert J.decode_coefs(cmyk.getvalue()) is None
    png = io.BytesIO()
    Image.fromarray(arr).save(png, format='PNG')
    assert J.decode_coefs(png.getvalue()) is None
    good = jpeg_bytes(arr, quality=80)
    info, raw = J.jpeg_parse(good)
    assert info is not None
    J.jpeg_coefs_into(raw[:len(raw) // 8], info, np.empty(info.nblocks * 64, np.int16))      # truncated: zeros or an error, never a crash
    assert J.jpeg_parse(b'\xff\xd8\xff')[0] is None


@needs_lib
def test_quantisation_tables_and_geometry():
    data = jpeg_bytes(synth(100, 60, 9), quality=75, subsampling=2)
    info, _ = J.decode_coefs(data)
    assert list(info.hs) == [2, 1, 1] and list(info.vs) == [2, 1, 1]
    assert list(info.blocks_w) == [14, 7, 7] and list(info.blocks_h) == [8, 4, 4]
    assert list(info.samp_w) == [100, 50, 50] and list(info.samp_h) == [60, 30, 30]
    assert info.nblocks == 14 * 8 + 2 * 28 and list(info.block0) == [0, 112, 140]
    q = Image.open(io.BytesIO(data)).quantization
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50,
          43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    for c, t in ((0, 0), (1, 1), (2, 1)):
        tab = list(q[t])
        nat = [0] * 64                       # older Pillows hand the tables out in zigzag order, newer ones in natural order
        for i in range(64):
            nat[zz[i]] = tab[i]
        assert list(info.qt[c]) in (tab, nat)


@needs_lib
def test_worker_hand_off_through_a_shared_memory_slab():
    """What a loader worker does with `device_jpeg`: baseline JPEGs leave the worker as coefficient blocks inside the slab
    (('coef', offset, info bytes)), anything the front half refuses (progressive, PNG) as pixels; the parent rebuilds both views."""
    import base64
    from multiprocessing import shared_memory
    arrs = [synth(200, 150, 1), synth(64, 48, 2), synth(120, 90, 3), synth(80, 60, 4)]
    png = io.BytesIO()
    Image.fromarray(arrs[3]).save(png, format='PNG')
    blobs = [base64.b64encode(jpeg_bytes(arrs[0], quality=85)), jpeg_bytes(arrs[1], quality=70, subsampling=0),
             jpeg_bytes(arrs[2], quality=80, progressive=True), png.getvalue()]
    shm = shared_memory.SharedMemory(create=True, size=1 << 20)
    try:
        items = J.decode_into(shm.name, blobs, device_jpeg=True)
        assert [it[0] == 'coef' for it in items] == [True, True, False, False]
        offs = [it[1] if it[0] == 'coef' else it[0] for it in items]
        assert offs == sorted(offs) and all(o % 16 == 0 for o in offs)
        for it, blob in zip(items, blobs):
            raw = J._jpeg_bytes(blob)
            if it[0] == 'coef':
                info, coefs = J.coef_item(it, shm.buf)
                got = O.backhalf(info, coefs)
                del coefs
            else:
                off, h, w = it
                got = np.ndarray((h, w, 3), dtype=np.uint8, buffer=shm.buf, offset=off).copy()
            np.testing.assert_array_equal(got, pillow_rgb(raw))
        plain = J.decode_into(shm.name, blobs[:2])                 # device_jpeg off: pixels, as in rounds 1-5
        assert all(len(it) == 3 and it[0] != 'coef' for it in plain)
    finally:
        J._SHM.pop(shm.name, None)
        shm.close()
        shm.unlink()


@needs_lib
def test_front_half_survives_corrupt_streams():
    """Bit flips, truncations, spliced garbage, header damage: the front half either refuses the stream or fills the coefficient array --
    it never reads or writes out of bounds (a crash would take a loader worker down)."""
    import random
    rnd = random.Random(1)
    base = [jpeg_bytes(synth(160, 120, i), quality=q, subsampling=s) for i, (q, s) in enumerate([(85, 2), (60, 1), (95, 0)])]
    try:
        rst = jpeg_bytes(synth(160, 120, 9), quality=75, restart_marker_blocks=4)
        if b'\xff\xdd' in rst:
            base.append(rst)
    except TypeError:
        pass
    decoded = refused = 0
    for it in range(800):
        d = bytearray(rnd.choice(base))
        mode = it % 4
        if mode == 0:
            for _ in range(rnd.randint(1, 8)):
                d[rnd.randrange(len(d))] = rnd.randrange(256)
        elif mode == 1:
            d = d[:rnd.randrange(2, len(d))]
        elif mode == 2:
            i = rnd.randrange(len(d))
            d[i:i + rnd.randint(1, 40)] = bytes(rnd.randrange(256) for _ in range(rnd.randint(0, 40)))
        else:
            d[rnd.randrange(2, min(700, len(d)))] = rnd.randrange(256)       # header region
        got = J.decode_coefs(bytes(d))
        if got is None:
            refused += 1
        else:
            decoded += 1
            info, coefs = got
            assert coefs.size == info.nblocks * 64 and 0 < info.width * info.height <= 1 << 26
    assert decoded > 0 and refused > 0
