"""Real-data training batches for the captioning path (SURVEY 8f ranks 1 + 3): what the reference's
`get_len_dataset(is_train=True)` + `get_transform(is_train=True)` + collate produce
(src/pipelines/tagger_caption_uni_pipeline_expanding_bertemb.py:358-518), without torch's DataLoader.

  data/<name>/train.tsv                 key \\t ... \\t base64 JPEG          (LoadImage, transform.py:106-136)
  data/<name>/train.caption.tsv         key \\t json [{"caption": str}, ...] (LoadCaption, transform.py:168-188)
  data/<name>/train.num_caption.tsv     key \\t n         optional: derived from the caption file when absent
  data/<name>/train.label[.v<ver>].tsv  key \\t json [{"class": str, "conf": float}, ...]   optional (LoadLabel)
File naming follows TSVDataset.get_data (src/tools/tsv/tsv_io.py:529-553).

One sample = one (image, caption) pair (CaptionIdxTSVDataset, dataset.py:35-75).  Per sample: decode the JPEG on the host,
draw the augmentation parameters (augment.py), tensorize the caption (tensorizer.py, text_b = '' because the pipeline
hard-codes add_od_labels False at ..._bertemb.py:424), build the tag `label` vector (CaptionTaggerTensorizer,
dataset.py:774-820).  Per batch: crop/resize/jitter/flip/normalise of all images on the GPU (csrc/preproc.hip).

Deviations, stated: (i) `encode='nltk'` (the reference's default way of adding caption nouns/adjectives to `label`) needs
nltk's tokenizer + POS tagger, which this image does not have: pass `pos_tagger=` (a callable caption -> [(word, tag)]) or
use `encode='bert'` (the reference's other branch: every WordPiece of the caption).  `label` only feeds the tag loss,
which this pipeline reports but does not add to the objective (..._bertemb.py:170), so the trained weights do not depend
on it.  (ii) Shuffling is a seeded permutation per epoch split across ranks like torch's DistributedSampler (pad to a
multiple of world size, stride by rank); the reference's sampler stack (uni_pipeline.py:263-339) reduces to that for one
dataset without composite splits."""
import json
import os.path as op
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import torch

from .augment import TrainAugmentation
from .imageio import CoefImage, decode_image
from .jpegdec import decode_coefs, jpeg_lib
from .tsv import TSVFile


def data_file(root, data, split, t=None, version=None):
    """TSVDataset.get_data naming: <split>.tsv, <split>.<t>.tsv, <split>.<t>.v<version>.tsv."""
    base = op.join(root, data)
    if t is None:
        return op.join(base, '{}.tsv'.format(split))
    if version is None or version in (0, 'None', '0'):
        return op.join(base, '{}.{}.tsv'.format(split, t))
    return op.join(base, '{}.{}.v{}.tsv'.format(split, t, version))


class CaptionIdx(object):
    """CaptionIdxTSVDataset: the list of (key, idx_img, idx_cap), one entry per caption."""

    def __init__(self, root, data, split, caption_version=None):
        f = data_file(root, data, split, 'num_caption', caption_version)
        if op.isfile(f):
            num = [(r[0], int(r[1])) for r in TSVFile(f)]
        else:
            num = [(r[0], len(json.loads(r[1]))) for r in TSVFile(data_file(root, data, split, 'caption', caption_version))]
        self.k_img_cap = [(k, i, c) for i, (k, n) in enumerate(num) for c in range(n)]

    def __len__(self):
        return len(self.k_img_cap)

    def __getitem__(self, idx):
        return self.k_img_cap[idx]


class TagLabelTensorizer(object):
    """CaptionTaggerTensorizer.tensorize for category 'bert' (dataset.py:774-820): multi-hot over the BERT vocabulary of
    (a) the detector tags with conf >= threshold, word by word, and (b) tags taken from the caption."""

    def __init__(self, tokenizer, threshold=0.2, encode='nltk', caption_only=False, pos_tagger=None):
        if encode not in ('nltk', 'bert', None):
            raise ValueError('encode %r' % (encode,))
        self.tok, self.threshold, self.encode, self.caption_only, self.pos_tagger = tokenizer, threshold, encode, caption_only, pos_tagger

    def _id(self, token):
        return self.tok.convert_tokens_to_ids([token])[0]

    def tensorize(self, labels, caption=None):
        v = torch.zeros(self.tok.vocab_size)
        if not self.caption_only:
            for tag in labels or []:
                if tag['conf'] >= self.threshold:
                    for t in tag['class'].split(' '):
                        v[self._id(t)] = 1
        if caption is not None:
            if self.encode == 'nltk':
                if self.pos_tagger is None:
                    self.pos_tagger = resolve_nltk_tagger()
                for word, pos in self.pos_tagger(caption):
                    if pos in ('JJ', 'NN', 'NNP'):
                        for t in word.split(' '):
                            v[self._id(t)] = 1
            elif self.encode == 'bert':
                for i in self.tok.convert_tokens_to_ids(self.tok.tokenize(caption)):
                    v[i] = 1
        return {'label': v}


def nltk_pos_tag(caption):
    """The reference's own tagger (dataset.py:801-804).  A module-level function: the tensorizer stays picklable (spawned loader
    workers, copy.deepcopy)."""
    import nltk
    return nltk.pos_tag(nltk.word_tokenize(caption))


def resolve_nltk_tagger():
    """encode='nltk': nltk AND its punkt / averaged_perceptron_tagger data must be there -- probed once on a dummy caption, so that a
    missing package (ImportError) or missing data (LookupError, raised inside the loader thread on the first caption otherwise)
    both give the same explanation."""
    try:
        nltk_pos_tag('a dog on a bench')
    except (ImportError, LookupError) as e:
        raise RuntimeError("encode='nltk' needs nltk.word_tokenize + nltk.pos_tag and their data (punkt, averaged_perceptron_tagger), "
                           "which are not usable here (%s: %s): pass pos_tagger=callable(caption) -> [(word, tag)] or set encode: bert"
                           % (type(e).__name__, str(e).strip().splitlines()[0] if str(e).strip() else ''))
    return nltk_pos_tag


class CaptionTrainSet(object):
    def __init__(self, root, data, tensorizer, tagger, split='train', caption_version=None, label_version=None,
                 augmentation=None, device_jpeg=None):
        # device_jpeg: None = on when the host front half (libvitcap_jpeg.so) is built
        self.device_jpeg = (jpeg_lib() is not None) if device_jpeg is None else (bool(device_jpeg) and jpeg_lib() is not None)
        self.idx = CaptionIdx(root, data, split, caption_version)
        self.images = TSVFile(data_file(root, data, split))
        self.captions = TSVFile(data_file(root, data, split, 'caption', caption_version))
        lf = data_file(root, data, split, 'label', label_version)
        self.labels = TSVFile(lf) if op.isfile(lf) else None
        self.tensorizer, self.tagger = tensorizer, tagger
        self.aug = augmentation or TrainAugmentation()
        self._lock = threading.Lock()          # TSVFile keeps one file position per object

    def __len__(self):
        return len(self.idx)

    def captions_of(self, idx_img):
        with self._lock:
            row = self.captions[idx_img]
        return [c['caption'] for c in json.loads(row[1])]

    def sample(self, i, epoch=0):
        """-> dict(rgb uint8 HWC, aug params, caption, text tensors, label)."""
        key, idx_img, idx_cap = self.idx[i]
        with self._lock:
            img_row = self.images[idx_img]
            cap_row = self.captions[idx_img]
            lab_row = self.labels[idx_img] if self.labels is not None else None
        # device JPEG back half (round 6): a baseline JPEG is only entropy-decoded here (ctypes call: the GIL is released, like inside
        # Pillow's decoder); the GPU finishes it in front of the crop / resize (imageio.TrainImagePreprocessor)
        got = decode_coefs(img_row[-1]) if self.device_jpeg else None
        rgb = CoefImage(*got) if got is not None else decode_image(img_row[-1])
        caption = json.loads(cap_row[1])[idx_cap]['caption']
        labels = json.loads(lab_row[1]) if lab_row is not None else []
        # the masking draws of tensorize_ab come from Python's global `random` in the reference; here a per-sample generator
        # keeps a batch a function of (seed, epoch, index) under any thread schedule
        rng = random.Random((self.aug.seed * 7919 + epoch) * 2147483647 + i)
        out = self.tensorizer.tensorize_ab(caption, '', rng=rng)
        out['token_type_ids'] = out.pop('segment_ids')           # RenameKey (..._bertemb.py:516)
        out.update(self.tagger.tensorize(labels, caption))
        out['rgb'] = rgb
        out['aug'] = self.aug.params(rgb.shape[0], rgb.shape[1], index=i, epoch=epoch)
        out['caption'], out['key'], out['idx_img'] = caption, key, idx_img
        return out


def epoch_indices(n, epoch, seed, rank=0, world=1, shuffle=True):
    """torch.utils.data.DistributedSampler's index rule on a seeded permutation."""
    idx = list(range(n))
    if shuffle:
        random.Random(seed * 1000003 + epoch).shuffle(idx)
    total = (n + world - 1) // world * world
    idx += idx[:total - n]
    return idx[rank:total:world]


TENSOR_KEYS = ('input_ids', 'attention_mask', 'masked_pos', 'masked_ids', 'token_type_ids', 'label')


class TrainBatchLoader(object):
    """Endless iterator of training batches: text tensors stacked like the default collate, `image` produced on the GPU by
    `image_transform(list of rgb, list of params)`.  Decoding/tensorizing runs on `workers` threads (Pillow's JPEG decoder
    releases the GIL); one batch is prepared ahead on a background thread."""

    def __init__(self, dataset, per_gpu, image_transform, rank=0, world=1, seed=0, workers=8, want_captions=False, start_iter=0):
        self.ds, self.B, self.tf = dataset, int(per_gpu), image_transform
        self.rank, self.world, self.seed, self.want_captions = rank, world, seed, want_captions
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.q = queue.Queue(maxsize=2)
        self._stop = False
        self._start_iter = start_iter
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _host_batches(self):
        epoch, it = 0, 0
        while True:
            idx = epoch_indices(len(self.ds), epoch, self.seed, self.rank, self.world)
            for s in range(0, len(idx) - self.B + 1, self.B):           # drop the ragged tail like the reference's batch sampler
                if it >= self._start_iter:
                    yield epoch, idx[s:s + self.B]
                it += 1
            if len(idx) < self.B:
                raise ValueError('dataset shard of %d samples is smaller than the per-GPU batch %d' % (len(idx), self.B))
            epoch += 1

    def _produce(self):
        try:
            for epoch, ids in self._host_batches():
                if self._stop:
                    return
                samples = list(self.pool.map(lambda i: self.ds.sample(i, epoch), ids))
                self.q.put(samples)
        except BaseException as e:       # surface worker errors in the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        samples = self.q.get()
        if isinstance(samples, BaseException):
            raise samples
        batch = {k: torch.stack([s[k] for s in samples]) for k in TENSOR_KEYS}
        batch['image'] = self.tf([s['rgb'] for s in samples], [s['aug'] for s in samples])
        batch['key'] = [s['key'] for s in samples]
        if self.want_captions:
            batch['captions'] = [self.ds.captions_of(s['idx_img']) for s in samples]
        return batch

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
