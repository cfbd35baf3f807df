"""Device JPEG back half (csrc/jpeg.hip, vitcap_jpeg_backhalf) on the MI355X: coefficient blocks from the host entropy decoder ->
pixels BIT-IDENTICAL to Pillow's decoder (libjpeg-turbo defaults: ISLOW inverse DCT, fancy upsampling, fixed-point YCbCr -> RGB), for
every sampling mode / size of the CPU test set (tests/test_jpeg_cpu.py pins the numpy oracle against the same Pillow pixels); and the
input side end to end: ImagePreprocessor fed with CoefImage items gives the same batch, bit for bit, as fed with Pillow-decoded arrays."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

from tests.test_jpeg_cpu import SIZES, jpeg_bytes, pillow_rgb, synth

pytestmark = pytest.mark.gpu


def _coef_images(datas):
    from vitcap_amd import jpegdec as J
    from vitcap_amd.imageio import CoefImage
    out = []
    for d in datas:
        got = J.decode_coefs(d)
        assert got is not None
        out.append(CoefImage(*got))
    return out


@pytest.mark.parametrize('sub', [0, 1, 2], ids=['444', '422', '420'])
def test_backhalf_equals_pillow_bit_for_bit(sub):
    from vitcap_amd.imageio import jpeg_backhalf
    datas = []
    for i, (w, h) in enumerate(SIZES):
        if sub != 0 and (w + 1) // 2 <= 2:
            continue
        for q in (92, 60, 25):
            datas.append(jpeg_bytes(synth(w, h, 10 * i + q), quality=q, subsampling=sub))
    keep = []
    outs = jpeg_backhalf(_coef_images(datas), torch.device('cuda'), keep)       # ONE launch pair for the whole mixed-size batch
    torch.cuda.synchronize()
    for d, o in zip(datas, outs):
        want = pillow_rgb(d)
        got = o.cpu().numpy()[:, :want.shape[1] * 3].reshape(want.shape)       # rows are padded to a multiple of 4 bytes
        assert np.array_equal(got, want), 'pixels differ: %d of %d (max |d| %d)' % (
            int((got != want).sum()), want.size, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_grey_restart_markers_and_extreme_coefficients():
    from vitcap_amd.imageio import jpeg_backhalf
    arr = synth(321, 203, 7)
    datas = [jpeg_bytes(arr[:, :, 0], quality=80), jpeg_bytes(arr, quality=85, optimize=True), jpeg_bytes(arr, quality=2), jpeg_bytes(arr, quality=100)]
    # saturated content: hard black / white checkerboard at quality 10 overshoots the [0, 255] range inside the IDCT (range-limit table)
    chk = ((np.indices((160, 200)).sum(0) // 3) % 2 * 255).astype(np.uint8)
    datas.append(jpeg_bytes(np.stack([chk, 255 - chk, chk], -1), quality=10, subsampling=2))
    try:
        rst = jpeg_bytes(arr, quality=75, restart_marker_blocks=5)
        if b'\xff\xdd' in rst:
            datas.append(rst)
    except TypeError:
        pass
    keep = []
    outs = jpeg_backhalf(_coef_images(datas), torch.device('cuda'), keep)
    torch.cuda.synchronize()
    for d, o in zip(datas, outs):
        want = pillow_rgb(d)
        assert np.array_equal(o.cpu().numpy()[:, :want.shape[1] * 3].reshape(want.shape), want)


def test_preprocessor_accepts_entropy_decoded_jpegs():
    """The loader's new hand-off: CoefImage items and Pillow-decoded arrays, mixed in one batch, give the SAME (B,3,384,384) tensor
    bit for bit -- including an image the front half refuses (progressive), which arrives decoded by Pillow."""
    from vitcap_amd import jpegdec as J
    from vitcap_amd.imageio import CoefImage, ImagePreprocessor
    pre = ImagePreprocessor('cuda', 384, 0.9)
    datas = [jpeg_bytes(synth(w, h, 3 * i), quality=85, subsampling=(i % 3)) for i, (w, h) in enumerate([(640, 480), (480, 640), (500, 375), (1024, 683), (431, 433)])]
    datas.append(jpeg_bytes(synth(600, 450, 99), quality=85, progressive=True))
    ref_imgs = [J.decode_image(d) for d in datas]
    new_imgs = []
    for d in datas:
        got = J.decode_coefs(d)
        new_imgs.append(CoefImage(*got) if got is not None else J.decode_image(d))
    assert sum(isinstance(x, CoefImage) for x in new_imgs) == 5
    a, a8 = pre(ref_imgs, want_u8=True)
    b, b8 = pre(new_imgs, want_u8=True)
    torch.cuda.synchronize()
    assert torch.equal(a8, b8) and torch.equal(a, b)


def test_train_preprocessor_accepts_entropy_decoded_jpegs():
    """Training input side (dataset.CaptionTrainSet.sample with device_jpeg): the same drawn augmentation applied to a CoefImage and to
    the Pillow-decoded array gives the same bytes and the same normalised tensor."""
    from vitcap_amd import jpegdec as J
    from vitcap_amd.augment import TrainAugmentation
    from vitcap_amd.imageio import CoefImage, TrainImagePreprocessor
    datas = [jpeg_bytes(synth(w, h, 5 * i + 1), quality=88, subsampling=(2 - i % 3)) for i, (w, h) in enumerate([(640, 480), (480, 640), (500, 375), (333, 500)])]
    ref = [J.decode_image(d) for d in datas]
    new = [CoefImage(*J.decode_coefs(d)) for d in datas]
    aug = TrainAugmentation(seed=11)
    params = [aug.params(im.shape[0], im.shape[1], index=i, epoch=2) for i, im in enumerate(ref)]
    assert [n.shape for n in new] == [r.shape for r in ref]
    pre = TrainImagePreprocessor('cuda', out_dtype=torch.float32)
    a, a8 = pre(ref, params, want_u8=True)
    b, b8 = pre(new, params, want_u8=True)
    torch.cuda.synchronize()
    assert torch.equal(a8, b8) and torch.equal(a, b)
