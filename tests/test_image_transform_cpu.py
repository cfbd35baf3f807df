"""Input side: the oracle's restatement of Pillow's 8-bit bicubic resampling against Pillow itself,
and the C-ABI's host-side weight tables / geometry against the restatement (no GPU needed)."""
import ctypes as C

import numpy as np
import pytest
from PIL import Image

from oracle import image_oracle as IO


def _img(h, w, seed):
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    big = np.asarray(Image.fromarray(base, 'RGB').resize((w, h), Image.BILINEAR)).copy()
    noise = g.integers(-20, 21, size=big.shape)
    return np.clip(big.astype(np.int64) + noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize('h,w,oh,ow', [(480, 640, 384, 512), (333, 500, 384, 576), (640, 427, 575, 384), (200, 300, 384, 576),
                                      (384, 384, 384, 384), (97, 131, 40, 57)])
def test_restatement_equals_pillow(h, w, oh, ow):
    img = _img(h, w, h * 1000 + w)
    want = np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh), Image.BICUBIC))
    got = IO.resample_restated(img, oh, ow)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_cabi_weight_tables_and_geometry():
    from vitcap_amd._lib import check, lib
    for in_size, out_size in [(640, 512), (480, 384), (427, 384), (300, 576), (5000, 384), (384, 384), (385, 384)]:
        ks, bounds, kk = IO.coeffs_restated(in_size, out_size)
        b = np.zeros((out_size, 2), dtype=np.int32)
        k = np.zeros((out_size, ks), dtype=np.int32)
        ks_c = C.c_int(0)
        check(lib.vitcap_resample_coeffs(in_size, out_size, C.byref(ks_c), b.ctypes.data_as(C.c_void_p),
                                         k.ctypes.data_as(C.c_void_p), k.size), 'resample_coeffs')
        assert ks_c.value == ks and np.array_equal(b, bounds) and np.array_equal(k, kk), (in_size, out_size)
        assert (k.sum(1) - (1 << 22)).__abs__().max() <= ks          # rows sum to 1.0 up to rounding of each tap
    g = np.random.default_rng(5)
    for _ in range(300):
        h, w = int(g.integers(384, 1400)), int(g.integers(384, 1400))
        oh, ow, y0, x0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.vitcap_resized_geometry(h, w, 384, 384, C.byref(oh), C.byref(ow), C.byref(y0), C.byref(x0)), 'geometry')
        nh, nw = IO.resized_size(h, w, 384)
        assert (oh.value, ow.value) == (nh, nw) and (y0.value, x0.value) == IO.crop_origin(nh, nw, 384)
    with pytest.raises(RuntimeError):
        oh = C.c_int()
        check(lib.vitcap_resized_geometry(100, 100, 200, 384, C.byref(oh), C.byref(oh), C.byref(oh), C.byref(oh)), 'geometry')


def test_tsv_roundtrip(tmp_path):
    from vitcap_amd.tsv import TSVFile, generate_lineidx, tsv_writer
    rows = [('k%d' % i, 'x' * (i * 7 % 50), b'bytes%d' % i) for i in range(57)]
    f = str(tmp_path / 'sub' / 'a.tsv')
    tsv_writer(rows, f)
    t = TSVFile(f)
    assert len(t) == 57 and t[0] == ['k0', '', 'bytes0'] and t[56] == ['k56', 'x' * (56 * 7 % 50), 'bytes56']
    assert [r[0] for r in t] == ['k%d' % i for i in range(57)]
    import os
    os.remove(str(tmp_path / 'sub' / 'a.lineidx.8b'))       # text index only
    assert TSVFile(f)[31][0] == 'k31'
    os.remove(str(tmp_path / 'sub' / 'a.lineidx'))
    with pytest.raises(FileNotFoundError):
        TSVFile(f)[0]
    generate_lineidx(f)
    assert TSVFile(f).get_key(40) == 'k40'
    with pytest.raises(IndexError):
        TSVFile(f)[57]


def test_train_transform_restatement_equals_pillow():
    """The numpy restatement of the train-time image arithmetic (bilinear Resample.c on the cropped image, Blend.c float32
    blend, rgb2l, mean-gray) == Pillow itself, over drawn parameters and the ends of the factor range."""
    import numpy as np
    from oracle import image_oracle as IO
    from vitcap_amd.augment import TrainAugmentation
    rng = np.random.default_rng(5)
    aug = TrainAugmentation(seed=11)
    for t in range(6):
        H, W = int(rng.integers(120, 520)), int(rng.integers(120, 640))
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([np.sin(xx / 31.0 + c) * 90 + np.cos(yy / 17.0) * 50 + 128 for c in range(3)], -1)
        img = np.clip(img + rng.normal(0, 30, img.shape), 0, 255).astype(np.uint8)
        pr = aug.params(H, W, index=t)
        if t == 4:
            pr['ops'] = [(0, 1.4), (2, 0.6), (1, 1.4)]
        if t == 5:
            pr['ops'] = [(1, 0.6), (2, 1.4), (0, 0.6)]
        a8, af = IO.train_transform_reference(img, pr['box'], pr['ops'], pr['flip'])
        b8, bf = IO.train_transform_restated(img, pr['box'], pr['ops'], pr['flip'])
        assert np.array_equal(a8, b8) and np.array_equal(af, bf), (t, pr)


def test_augmentation_parameters():
    """RandomResizedCrop / ColorJitter / flip parameter logic (torchvision 0.7 restated): boxes inside the image with area
    and aspect ratio in range, factors in [0.6, 1.4], every operation once in a shuffled order, deterministic in
    (seed, epoch, index), the central-crop fallback for extreme aspect ratios."""
    import math
    import random
    from vitcap_amd.augment import TrainAugmentation, random_resized_crop_params
    aug = TrainAugmentation(seed=7)
    orders, flips, areas = set(), 0, []
    for i in range(400):
        H, W = 300 + (i * 7) % 400, 280 + (i * 13) % 500
        p = aug.params(H, W, index=i)
        top, left, h, w = p['box']
        assert 0 <= top and 0 <= left and h > 0 and w > 0 and top + h <= H and left + w <= W
        frac = h * w / float(H * W)
        assert 0.07 < frac <= 1.0 and 0.70 < w / float(h) < 1.40, (p, H, W)
        areas.append(frac)
        assert sorted(o for o, _ in p['ops']) == [0, 1, 2] and all(0.6 <= f <= 1.4 for _, f in p['ops'])
        orders.add(tuple(o for o, _ in p['ops']))
        flips += p['flip']
        assert p == aug.params(H, W, index=i) and p != aug.params(H, W, index=i, epoch=1)
    assert len(orders) == 6 and 140 < flips < 260 and 0.4 < sum(areas) / len(areas) < 0.68
    # an image so elongated that no drawn box fits in 10 attempts falls back to the central crop at the nearest ratio
    box = random_resized_crop_params(random.Random(0), 20, 2000, scale=(0.9, 1.0))
    assert box == (0, (2000 - int(round(20 * 4. / 3.))) // 2, 20, int(round(20 * 4. / 3.)))
    assert aug.params(64, 64, 0)['box'][2] <= 64 and math.isfinite(aug.params(64, 64, 0)['ops'][0][1])
