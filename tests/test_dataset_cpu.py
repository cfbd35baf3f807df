"""Host side of the real-data training batches (vitcap_amd/dataset.py): file naming, the (image, caption) index, per-sample
decode + tensorize + tag label + augmentation parameters, the distributed epoch order, the threaded loader."""
import base64
import io
import json

import numpy as np
import pytest
import torch

from vitcap_amd.dataset import (CaptionIdx, CaptionTrainSet, TagLabelTensorizer, TrainBatchLoader, data_file,
                                epoch_indices)
from vitcap_amd.tensorizer import CaptionTensorizer
from vitcap_amd.tokenizer import BertWordPieceTokenizer
from vitcap_amd.tsv import tsv_writer

WORDS = 'a man woman dog cat horse riding sitting on the street bench red blue two people table pizza'.split()
VOCAB = ['[PAD]'] + ['[unused%d]' % i for i in range(1, 100)] + ['[UNK]', '[CLS]', '[SEP]', '[MASK]'] + sorted(set(WORDS))


def make_dataset(root, n_img=7, with_num=False):
    from PIL import Image
    g = np.random.default_rng(0)
    d = root / 'toy'
    d.mkdir()
    img_rows, cap_rows, lab_rows, num_rows = [], [], [], []
    for i in range(n_img):
        h, w = int(g.integers(90, 200)), int(g.integers(90, 260))
        buf = io.BytesIO()
        Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8), 'RGB').save(buf, format='JPEG', quality=90)
        key = 'k%d' % i
        caps = [{'caption': ' '.join(g.choice(WORDS, size=int(g.integers(3, 9))))} for _ in range(1 + i % 3)]
        img_rows.append((key, base64.b64encode(buf.getvalue())))
        cap_rows.append((key, json.dumps(caps)))
        lab_rows.append((key, json.dumps([{'class': 'dog', 'conf': 0.9}, {'class': 'cat', 'conf': 0.1}])))
        num_rows.append((key, str(len(caps))))
    tsv_writer(img_rows, str(d / 'train.tsv'))
    tsv_writer(cap_rows, str(d / 'train.caption.tsv'))
    tsv_writer(lab_rows, str(d / 'train.label.vvinvl.tsv'))
    if with_num:
        tsv_writer(num_rows, str(d / 'train.num_caption.tsv'))
    return sum(1 + i % 3 for i in range(n_img))


def test_file_naming():
    assert data_file('data', 'coco', 'train') == 'data/coco/train.tsv'
    assert data_file('data', 'coco', 'train', 'caption') == 'data/coco/train.caption.tsv'
    assert data_file('data', 'coco', 'train', 'label', 'vinvl') == 'data/coco/train.label.vvinvl.tsv'
    assert data_file('data', 'coco', 'train', 'label', 0) == 'data/coco/train.label.tsv'


@pytest.mark.parametrize('with_num', [False, True])
def test_samples_and_loader(tmp_path, with_num):
    n = make_dataset(tmp_path, with_num=with_num)
    tok = BertWordPieceTokenizer(tokens=VOCAB)
    idx = CaptionIdx(str(tmp_path), 'toy', 'train')
    assert len(idx) == n and idx[0] == ('k0', 0, 0) and idx[2] == ('k1', 1, 1)
    tz = CaptionTensorizer(tok, max_seq_length=70, max_seq_a_length=20, is_train=True)
    ds = CaptionTrainSet(str(tmp_path), 'toy', tz, TagLabelTensorizer(tok, encode='bert'), label_version='vinvl', device_jpeg=False)
    s = ds.sample(2, epoch=0)
    assert s['rgb'].dtype == np.uint8 and s['rgb'].ndim == 3 and s['key'] == 'k1'
    # default (round 6): a baseline JPEG leaves the sample only ENTROPY-decoded (the GPU finishes it, imageio.TrainImagePreprocessor);
    # the augmentation is drawn from the same image size, so every other field of the sample is unchanged
    from vitcap_amd.imageio import CoefImage
    from vitcap_amd.jpegdec import jpeg_lib
    if jpeg_lib() is not None:
        ds_c = CaptionTrainSet(str(tmp_path), 'toy', tz, TagLabelTensorizer(tok, encode='bert'), label_version='vinvl')
        sc = ds_c.sample(2, epoch=0)
        assert isinstance(sc['rgb'], CoefImage) and sc['rgb'].shape == s['rgb'].shape and sc['aug'] == s['aug']
        assert torch.equal(sc['input_ids'], s['input_ids']) and torch.equal(sc['label'], s['label'])
        from oracle import jpeg_backhalf as JO
        assert np.array_equal(JO.backhalf(sc['rgb'].info, sc['rgb'].coefs), s['rgb'])
    assert s['input_ids'].shape == (70,) and s['attention_mask'].shape == (70, 70) and s['masked_ids'].shape == (3,)
    assert s['token_type_ids'].shape == (70,) and 'segment_ids' not in s
    assert int(s['input_ids'][0]) == tok.vocab['[CLS]'] and int(s['masked_pos'].sum()) == int((s['masked_ids'] != 0).sum())
    assert s['label'][tok.vocab['dog']] == 1 and s['label'][tok.vocab['cat']] == 0          # conf 0.1 < threshold
    for w in s['caption'].split():
        assert s['label'][tok.vocab[w]] == 1                                                 # encode='bert'
    top, left, h, w = s['aug']['box']
    assert top + h <= s['rgb'].shape[0] and left + w <= s['rgb'].shape[1]
    s2 = ds.sample(2, epoch=0)
    assert torch.equal(s['input_ids'], s2['input_ids']) and s['aug'] == s2['aug']           # a function of (seed, epoch, index)
    assert ds.sample(2, epoch=1)['aug'] != s['aug']
    assert ds.captions_of(1) == [c['caption'] for c in json.loads(ds.captions[1][1])]

    calls = []

    def fake_tf(images, params):
        calls.append(len(images))
        return torch.zeros(len(images), 3, 8, 8)

    seen = []
    for rank in range(2):
        ld = TrainBatchLoader(ds, per_gpu=3, image_transform=fake_tf, rank=rank, world=2, seed=5, workers=2, want_captions=True)
        b = next(ld)
        assert b['input_ids'].shape == (3, 70) and b['attention_mask'].shape == (3, 70, 70) and b['label'].shape == (3, len(VOCAB))
        assert b['image'].shape == (3, 3, 8, 8) and len(b['key']) == 3 and len(b['captions']) == 3
        b2 = next(ld)
        seen.append(b['key'] + b2['key'])
        ld.close()
    want0, want1 = epoch_indices(n, 0, 5, 0, 2), epoch_indices(n, 0, 5, 1, 2)
    assert len(want0) == len(want1) == (n + 1) // 2 and sorted(set(want0 + want1)) == list(range(n))
    assert seen[0][:3] == [idx[i][0] for i in want0[:3]] and seen[1][:3] == [idx[i][0] for i in want1[:3]]


def test_epoch_indices_are_a_sharded_permutation():
    for n, world in ((10, 1), (10, 4), (7, 3)):
        parts = [epoch_indices(n, 3, 11, r, world) for r in range(world)]
        assert len({len(p) for p in parts}) == 1
        assert sorted(set(sum(parts, []))) == list(range(n))
        assert parts != [epoch_indices(n, 4, 11, r, world) for r in range(world)] or n < 3
    assert epoch_indices(5, 0, 1, shuffle=False) == [0, 1, 2, 3, 4]


def test_loader_surfaces_worker_errors(tmp_path):
    make_dataset(tmp_path, n_img=2)
    tok = BertWordPieceTokenizer(tokens=VOCAB)
    tz = CaptionTensorizer(tok, max_seq_length=70, max_seq_a_length=20, is_train=True)
    ds = CaptionTrainSet(str(tmp_path), 'toy', tz, TagLabelTensorizer(tok, encode='nltk'))    # no POS tagger available
    ld = TrainBatchLoader(ds, per_gpu=2, image_transform=lambda a, b: None, workers=1)
    with pytest.raises(RuntimeError, match='nltk'):
        next(ld)
