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


def test_run_py_eval_on_image_tsv(tmp_path, monkeypatch):
    """run.py -c yaml on the reference's data layout data/<name>/<split>.tsv: predictions come back as the reference's
    predict TSV (+ .lineidx files), one row per key, equal to captioning the oracle-transformed images directly."""
    import json
    import yaml
    from PIL import Image
    import run
    from oracle import image_oracle as IO
    from vitcap_amd.imageio import decode_image
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.tsv import TSVFile, tsv_writer
    monkeypatch.chdir(tmp_path)
    enc = tmp_path / 'enc'
    enc.mkdir()
    toks = ['[PAD]'] + ['w%d' % i for i in range(1, 30522)]
    toks[100], toks[101], toks[102], toks[103] = '[UNK]', '[CLS]', '[SEP]', '[MASK]'
    (enc / 'vocab.txt').write_text('\n'.join(toks) + '\n')
    rows = []
    for i, (h, w) in enumerate(SIZES[:5]):
        buf = io.BytesIO()
        Image.fromarray(_img(h, w, 300 + i), 'RGB').save(buf, format='JPEG', quality=90)
        rows.append(('coco_%d' % i, base64.b64encode(buf.getvalue())))
    tsv_writer(rows, str(tmp_path / 'data' / 'toy' / 'test.tsv'))
    sd = ImageCaptioning().load_recipe(0).state_dict()
    ck = tmp_path / 'base.pt'
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}, 'iteration': 0}, ck)
    cfg = {'type': 'pipeline_eval_multi', 'all_test_data': [{'test_data': 'toy', 'test_split': 'test'}],
           'param': {'full_expid': 'E', 'max_iter': 10, 'model_file': str(ck), 'text_encoder_type': str(enc), 'tagemb': 'cls',
                     'test_batch_size': 2, 'force_predict': True, 'crop_pct': 1.0, 'test_crop_size': 384,
                     'pipeline_type': {'from': 'vitcap_amd.pipeline', 'import': 'CaptionUniPipeline'}}}
    yf = tmp_path / 'exp.yaml'
    yf.write_text(yaml.safe_dump(cfg))
    kw = run.parse_general_args(['-c', str(yf)])
    getattr(run, kw.pop('type'))(**kw)
    pred = TSVFile(str(ck) + '.toy.test.predict.tsv')
    assert [r[0] for r in pred] == [r[0] for r in rows]
    m = ImageCaptioning().load_recipe(0).eval()
    batch = torch.from_numpy(np.stack([IO.transform_reference(decode_image(r[1]))[1] for r in rows])).cuda().to(torch.bfloat16)
    ids, lp = m({'image': batch, 'key': None})
    for i, r in enumerate(pred):
        cap = json.loads(r[1])[0]
        want = ' '.join('w%d' % t for t in ids[i, 0].tolist() if t not in (0, 101, 102))
        assert cap['caption'] == want, (i, cap['caption'], want)
        assert abs(cap['conf'] - float(torch.exp(lp[i, 0]))) < 1e-5


def test_train_transform_bit_exact():
    """Train-time transform (crop + bilinear resize, colour jitter in the drawn order, flip, normalise) on the device ==
    Pillow doing the same operations (oracle.image_oracle.train_transform_reference), bytes and fp32 tensor, for drawn
    parameters plus hand-picked corner cases: whole-image box, a 1:1 box of exactly 384x384 (no resampling), up-scaling of
    a tiny box, factors at both ends of the jitter range, a single operation, no operation."""
    from oracle import image_oracle as IO
    from vitcap_amd.augment import TrainAugmentation
    from vitcap_amd.imageio import TrainImagePreprocessor
    imgs = [_img(h, w, 31 * i + 5) for i, (h, w) in enumerate(SIZES)]
    aug = TrainAugmentation(seed=3)
    params = [aug.params(im.shape[0], im.shape[1], index=i, epoch=1) for i, im in enumerate(imgs)]
    extra = [
        (imgs[0], {'box': (0, 0, 480, 640), 'ops': [(0, 1.4), (1, 1.4), (2, 1.4)], 'flip': True}),
        (imgs[6], {'box': (100, 200, 384, 384), 'ops': [(2, 0.6), (1, 0.6), (0, 0.6)], 'flip': False}),
        (imgs[1], {'box': (10, 20, 17, 23), 'ops': [(1, 1.0), (0, 0.9999)], 'flip': True}),
        (imgs[2], {'box': (0, 0, 333, 500), 'ops': [], 'flip': False}),
        (imgs[5], {'box': (1, 0, 384, 1201), 'ops': [(2, 0.0), (0, 1.25)], 'flip': True}),
        (imgs[4], {'box': (0, 0, 384, 384), 'ops': [(1, 0.0)], 'flip': False}),
    ]
    all_imgs = imgs + [e[0] for e in extra]
    all_params = params + [e[1] for e in extra]
    pre = TrainImagePreprocessor('cuda', out_dtype=torch.float32)
    out, u8 = pre(all_imgs, all_params, want_u8=True)
    torch.cuda.synchronize()
    for i, (im, pr) in enumerate(zip(all_imgs, all_params)):
        want_u8, want_f = IO.train_transform_reference(im, pr['box'], pr['ops'], pr['flip'])
        got = u8[i].cpu().numpy()
        assert np.array_equal(got, want_u8), 'bytes differ for case %d %r: %d pixels, max %d' % (
            i, pr, int((got != want_u8).sum()), int(np.abs(got.astype(int) - want_u8.astype(int)).max()))
        assert np.array_equal(out[i].cpu().numpy(), want_f), 'normalised tensor differs for case %d' % i
    out_b = TrainImagePreprocessor('cuda', out_dtype=torch.bfloat16)(all_imgs, all_params)
    assert torch.equal(out_b, out.to(torch.bfloat16))


def test_train_transform_rejects_bad_parameters():
    from vitcap_amd.imageio import TrainImagePreprocessor
    pre = TrainImagePreprocessor('cuda')
    im = _img(100, 120, 1)
    with pytest.raises(RuntimeError, match='crop box'):
        pre([im], [{'box': (50, 0, 60, 120), 'ops': [], 'flip': False}])
    with pytest.raises(RuntimeError, match='twice'):
        pre([im], [{'box': (0, 0, 100, 120), 'ops': [(0, 1.0), (0, 1.1)], 'flip': False}])
