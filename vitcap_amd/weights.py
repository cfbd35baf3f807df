"""Deterministic, framework-independent weight recipe for the ViTCAP hot path.

There is no network on the GPU box and no released checkpoint in the reference tree, so every
parity test and the benchmark run on seeded random-init weights.  The recipe is a counter-based
integer hash of ``(seed, tensor name, flat index)`` mapped to ``uniform(-a, a)`` with
``a = std * sqrt(3)`` -- no transcendental functions, so it is bit-reproducible on any host.

The key names and shapes are the reference's checkpoint layout:
``module.bert.*`` / ``module.cls.*`` from ``ViTCAP.state_dict()`` wrapped by ``ImageCaptioning``
(src/pipelines/tagger_caption_uni_pipeline_expanding_bertemb.py:23-40) and
``image_encoder.module.*`` from the ``InputAsDict``-wrapped timm ViT
(src/pipelines/tagger_caption_uni_pipeline_expanding_bertemb.py:750-778).

Unlike the reference's own init (zeros for biases, ones for LayerNorm gains --
src/layers/bert/modeling_bert.py:578-589, timm vision_transformer.py:391-398) every tensor gets
non-trivial values here so that a dropped bias or gain shows up in the parity tests.

Recipe version 2 (round 2), made so that token-exact comparison of the bf16 device path with the fp32
reference is well conditioned (SURVEY.md section 7 "hard parts", measured in DESIGN.md section 5):

* every matrix / embedding table ('w' kind) is rounded to the nearest bfloat16 value, so the fp32
  reference and the bf16 device path hold IDENTICAL weights and only activation rounding separates them
  (logit noise 3.0e-3 -> 1.8e-3 rms);
* the two vocabulary biases (``*.predictions.bias``, kind 'vbias') follow a unigram-like prior: a
  sum of 12 uniforms (Irwin-Hall, std 1 nat, exact in fp32 adds) instead of uniform(+-0.035) -- trained
  heads have such a prior, and it is noise-free on both sides (biases stay fp32), which widens the typical
  top-2 margin without touching the data-dependent part of the logits;
* the [SEP] bias equals the largest bias of the table, so [SEP] is a live candidate at every step
  (beam hypotheses finish at many lengths) without winning the greedy argmax outright.
"""
import hashlib
import zlib
from collections import OrderedDict

import numpy as np

HIDDEN = 768
HEADS = 12
HEAD_DIM = 64
INTER = 3072
VOCAB = 30522
MAX_POS = 512
TYPE_VOCAB = 2
N_PATCH = 576
N_VIS = 577
VIT_DEPTH = 12
SPLIT_BLOCKS = 4
DEC_LAYERS = 4
IMG = 384
PATCH = 16


def _vit_block(prefix):
    d = OrderedDict()
    d[prefix + '.norm1.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.norm1.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.attn.qkv.weight'] = ((3 * HIDDEN, HIDDEN), 'w')
    d[prefix + '.attn.qkv.bias'] = ((3 * HIDDEN,), 'bias')
    d[prefix + '.attn.proj.weight'] = ((HIDDEN, HIDDEN), 'w')
    d[prefix + '.attn.proj.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.norm2.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.norm2.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.mlp.fc1.weight'] = ((INTER, HIDDEN), 'w')
    d[prefix + '.mlp.fc1.bias'] = ((INTER,), 'bias')
    d[prefix + '.mlp.fc2.weight'] = ((HIDDEN, INTER), 'w')
    d[prefix + '.mlp.fc2.bias'] = ((HIDDEN,), 'bias')
    return d


def _bert_embeddings(prefix):
    d = OrderedDict()
    d[prefix + '.word_embeddings.weight'] = ((VOCAB, HIDDEN), 'w')
    d[prefix + '.position_embeddings.weight'] = ((MAX_POS, HIDDEN), 'w')
    d[prefix + '.token_type_embeddings.weight'] = ((TYPE_VOCAB, HIDDEN), 'w')
    d[prefix + '.LayerNorm.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.LayerNorm.bias'] = ((HIDDEN,), 'bias')
    return d


def _lm_head(prefix):
    d = OrderedDict()
    d[prefix + '.predictions.bias'] = ((VOCAB,), 'vbias')
    d[prefix + '.predictions.transform.dense.weight'] = ((HIDDEN, HIDDEN), 'w')
    d[prefix + '.predictions.transform.dense.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.predictions.transform.LayerNorm.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.predictions.transform.LayerNorm.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.predictions.decoder.weight'] = ((VOCAB, HIDDEN), 'w')
    return d


def _bert_layer(prefix):
    d = OrderedDict()
    for n in ('query', 'key', 'value'):
        d['%s.attention.self.%s.weight' % (prefix, n)] = ((HIDDEN, HIDDEN), 'w')
        d['%s.attention.self.%s.bias' % (prefix, n)] = ((HIDDEN,), 'bias')
    d[prefix + '.attention.output.dense.weight'] = ((HIDDEN, HIDDEN), 'w')
    d[prefix + '.attention.output.dense.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.attention.output.LayerNorm.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.attention.output.LayerNorm.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.intermediate.dense.weight'] = ((INTER, HIDDEN), 'w')
    d[prefix + '.intermediate.dense.bias'] = ((INTER,), 'bias')
    d[prefix + '.output.dense.weight'] = ((HIDDEN, INTER), 'w')
    d[prefix + '.output.dense.bias'] = ((HIDDEN,), 'bias')
    d[prefix + '.output.LayerNorm.weight'] = ((HIDDEN,), 'ln_w')
    d[prefix + '.output.LayerNorm.bias'] = ((HIDDEN,), 'bias')
    return d


def state_dict_spec():
    """Ordered {checkpoint key: (shape, kind)} for the 288 tensors of the reference checkpoint."""
    d = OrderedDict()
    d.update(_bert_embeddings('module.bert.embeddings'))
    d.update(_bert_embeddings('module.bert.extra_embeddings'))
    for i in range(VIT_DEPTH):
        d.update(_vit_block('module.bert.encoder.blocks.%d' % i))
    for i in range(SPLIT_BLOCKS):
        d.update(_vit_block('module.bert.encoder.tag_blocks.%d' % i))
    for n in ('caption_pooler', 'pooler'):
        d['module.bert.%s.dense.weight' % n] = ((HIDDEN, HIDDEN), 'w')
        d['module.bert.%s.dense.bias' % n] = ((HIDDEN,), 'bias')
    d.update(_lm_head('module.bert.tag_logit'))
    for i in range(DEC_LAYERS):
        d.update(_bert_layer('module.bert.decoder.layer.%d' % i))
    d.update(_lm_head('module.cls'))
    d['image_encoder.module.cls_token'] = ((1, 1, HIDDEN), 'w')
    d['image_encoder.module.pos_embed'] = ((1, N_VIS, HIDDEN), 'w')
    d['image_encoder.module.patch_embed.proj.weight'] = ((HIDDEN, 3, PATCH, PATCH), 'w')
    d['image_encoder.module.patch_embed.proj.bias'] = ((HIDDEN,), 'bias')
    d['image_encoder.module.head.weight'] = ((1000, HIDDEN), 'w')
    d['image_encoder.module.head.bias'] = ((1000,), 'bias')
    return d


TIED_SRC = 'module.bert.embeddings.word_embeddings.weight'
TIED_DST = 'module.cls.predictions.decoder.weight'

_STD = {'w': 0.02, 'bias': 0.02, 'ln_w': 0.1}
_SQRT3 = np.float32(1.7320508)


def _hash_u32(key, n):
    """lowbias32-style avalanche of (key ^ index) over index = 0..n-1, vectorised uint32."""
    x = np.arange(n, dtype=np.uint32)
    x ^= np.uint32(key)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    # second round keyed by the rotated key so that tensors sharing low key bits decorrelate
    x += np.uint32(((key << 13) | (key >> 19)) & 0xffffffff)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    return x


RECIPE_VERSION = 2
VBIAS_STD = np.float32(1.0)      # nats; Irwin-Hall(12) has unit variance
SEP_ID = 102


def bf16_round(a):
    """float32 array -> nearest-even bfloat16 value, still stored as float32 (pure integer arithmetic)."""
    u = np.ascontiguousarray(a, dtype=np.float32).reshape(-1).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7fff) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xffff0000)
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def _uniform01(name, n, seed):
    key = zlib.crc32(('%d|%s' % (seed, name)).encode()) & 0xffffffff
    with np.errstate(over='ignore'):
        x = _hash_u32(key, n)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)      # [0, 1), exact


def gen_tensor(name, shape, kind, seed=0, vbias_std=None, bf16_exact=True):
    """One tensor of the recipe as a float32 numpy array (bit-reproducible).  `vbias_std`: standard deviation (nats) of the
    vocabulary biases' unigram-like prior (default VBIAS_STD = 1, recipe v2; the image-dependent golden family uses 0.25).
    `bf16_exact=False` leaves the matrices at their fp32 values (the recipe rounds them to bf16-representable values so that the
    reference and the device hold IDENTICAL weights): the case of a real fp32 checkpoint, whose weights `pack()` has to round."""
    n = int(np.prod(shape))
    if kind == 'vbias':
        acc = np.zeros(n, dtype=np.float32)
        for j in range(12):                       # fixed order: every add is one IEEE fp32 operation
            acc = acc + _uniform01('%s|g%d' % (name, j), n, seed)
        out = (acc - np.float32(6.0)) * (VBIAS_STD if vbias_std is None else np.float32(vbias_std))
        out[SEP_ID] = out.max()
        return out.reshape(shape)
    u = _uniform01(name, n, seed)
    v = u * np.float32(2.0) - np.float32(1.0)                              # [-1, 1), exact
    a = np.float32(_STD[kind]) * _SQRT3
    out = v * a
    if kind == 'ln_w':
        out = out + np.float32(1.0)
    if kind == 'w' and bf16_exact:
        out = bf16_round(out)
    return out.reshape(shape)


def make_state_dict(seed=0, tie_weights=True, keys=None, vbias_std=None, bf16_exact=True):
    """Numpy state dict under the reference's checkpoint key names.

    With ``tie_weights`` the LM-head decoder weight is the word-embedding tensor itself
    (src/layers/bert/modeling_bert.py:728-730); the same array object is returned under both keys.
    """
    spec = state_dict_spec()
    sd = OrderedDict()
    for name, (shape, kind) in spec.items():
        if keys is not None and name not in keys:
            continue
        if tie_weights and name == TIED_DST and TIED_SRC in sd:
            sd[name] = sd[TIED_SRC]
            continue
        sd[name] = gen_tensor(name, shape, kind, seed, vbias_std, bf16_exact)
    return sd


def tensor_digest(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype=np.float32).tobytes()).hexdigest()[:16]


def synthetic_images(batch, seed=1234):
    """uniform(-1,1) images, same hash recipe keyed by ('image', seed) -- SURVEY.md section 8d."""
    return gen_image_batch(batch, seed)


def gen_structured_images(batch, seed):
    """Synthetic images that DIFFER from each other where the model can see it (uniform noise images all have the same patch
    statistics, so random-init attention -- near-uniform -- summarises every one of them to the same vector and the caption
    hardly depends on the image): per image a colour offset in [-0.6, 0.6]^3, a 6 x 6 grid of 64-pixel blocks with their own
    colour offsets in [-0.3, 0.3]^3, and uniform noise of amplitude 0.1; values stay inside [-1, 1].  Same hash generator as
    gen_image_batch, keyed by ('simage', seed): bit-reproducible float32."""
    def u(tag, n):
        key = zlib.crc32(('simage|%d|%s' % (seed, tag)).encode()) & 0xffffffff
        with np.errstate(over='ignore'):
            x = _hash_u32(key, n)
        return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)) * np.float32(2.0) - np.float32(1.0)
    base = u('base', batch * 3).reshape(batch, 3, 1, 1) * np.float32(0.6)
    blocks = u('blocks', batch * 3 * 36).reshape(batch, 3, 6, 6) * np.float32(0.3)
    blocks = np.repeat(np.repeat(blocks, 64, axis=2), 64, axis=3)
    noise = u('noise', batch * 3 * IMG * IMG).reshape(batch, 3, IMG, IMG) * np.float32(0.1)
    return (base + blocks + noise).astype(np.float32)


def gen_image_batch(batch, seed):
    n = batch * 3 * IMG * IMG
    key = zlib.crc32(('image|%d' % seed).encode()) & 0xffffffff
    with np.errstate(over='ignore'):
        x = _hash_u32(key, n)
    u = (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u * np.float32(2.0) - np.float32(1.0)).reshape(batch, 3, IMG, IMG)
