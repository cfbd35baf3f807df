"""Device image transform (csrc/preproc.hip) against the oracle = torchvision's rules on Pillow's resize: bytes after the
crop bit-identical, normalised fp32 tensor bit-identical, bf16 = round-to-nearest-even of that."""
import base64
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    from PIL import Image
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    big = np.asarray(Image.fromarray(base, 'RGB').resize((w, h), Image.BILINEAR)).copy()
    return np.clip(big.astype(np.int64) + g.integers(-25, 26, size=big.shape), 0, 255).astype(np.uint8)


SIZES = [(480, 640), (640, 480), (333, 500), (427, 640), (384, 384), (385, 1201), (1080, 1920), (500, 375), (384, 1000)]


def test_batch_of_mixed_sizes_bit_exact():
    from oracle import image_oracle as IO
    from vitcap_amd.imageio import ImagePreprocessor
    imgs = [_img(h, w, 17 * i + 1) for i, (h, w) in enumerate(SIZES)]
    pre = ImagePreprocessor('cuda', out_dtype=torch.float32)
    out, u8 = pre(imgs, want_u8=True)
    torch.cuda.synchronize()
    for i, im in enumerate(imgs):
        want_u8, want_f = IO.transform_reference(im)
        assert np.array_equal(u8[i].cpu().numpy(), want_u8), 'bytes differ for %s' % (SIZES[i],)
        assert np.array_equal(out[i].cpu().numpy(), want_f), 'normalised tensor differs for %s' % (SIZES[i],)
    out_b = ImagePreprocessor('cuda', out_dtype=torch.bfloat16)(imgs)
    assert torch.equal(out_b, out.to(torch.bfloat16))
    assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0


def test_jpeg_rows_to_caption_batch(tmp_path):
    """(key, base64 JPEG) rows in the reference's TSV format -> decode -> device transform -> greedy captions; the
    batch fed to the engine equals the oracle's transform of the same decoded bytes."""
    from PIL import Image
    from oracle import image_oracle as IO
    from vitcap_amd.imageio import ImagePreprocessor, decode_image
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.tsv import TSVFile, tsv_writer
    rows, raw = [], []
    for i, (h, w) in enumerate(SIZES[:4]):
        buf = io.BytesIO()
        Image.fromarray(_img(h, w, 100 + i), 'RGB').save(buf, format='JPEG', quality=92)
        rows.append(('img%d' % i, base64.b64encode(buf.getvalue())))
        raw.append(buf.getvalue())
    tsv_writer(rows, str(tmp_path / 'test.tsv'))
    t = TSVFile(str(tmp_path / 'test.tsv'))
    assert len(t) == 4 and t.get_key(2) == 'img2'
    decoded = [decode_image(t[i][1]) for i in range(4)]
    for d, r in zip(decoded, raw):
        assert np.array_equal(d, np.asarray(Image.open(io.BytesIO(r)).convert('RGB')))
    batch = ImagePreprocessor('cuda', out_dtype=torch.float32)(decoded)
    want = np.stack([IO.transform_reference(d)[1] for d in decoded])
    assert np.array_equal(batch.cpu().numpy(), want)
    m = ImageCaptioning().load_recipe(0).eval()
    ids, lp = m({'image': batch.contiguous(), 'key': [r[0] for r in rows]})
    assert ids.shape == (4, 1, 20) and torch.isfinite(lp).all()
