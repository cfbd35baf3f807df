"""CPU oracle for the ViTCAP captioning hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the path named by
BASELINE.json ``north_star``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped package ``vitcap_amd``
never does (it fails loudly when the HIP library is missing).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference itself
(/root/reference, with the container-only shims of SURVEY.md section 8c) on the seeded weights of
``vitcap_amd/weights.py`` and commits its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.

Weights are addressed by the reference's checkpoint key names (``module.bert...``,
``module.cls...``, ``image_encoder.module...``) in a flat dict of torch
tensors.  All dense arithmetic is delegated to ATen (nn.functional), as in the reference
(SURVEY.md section 8c "third-party arithmetic").

Two formulations of the greedy decoder are provided:

* ``greedy_as_written``   -- what the reference executes: the whole 16-block ViT and the whole
  joint sequence are recomputed at each of the 19 steps (modeling_utils.py:798-867 with
  ``past=None``; SURVEY.md headline 4).
* ``greedy_incremental``  -- the algebraically identical incremental form the HIP path computes
  (encoder + visual-row decoder prefill once, 2 query rows per step against cached K/V).  With
  ``emulate_bf16=True`` it rounds operands to bfloat16 at exactly the points where the HIP
  kernels store bf16, so device token ids can be compared bit-for-bit.
"""
import math

import numpy as np

import torch
import torch.nn.functional as F

HID = 768
HEADS = 12
HD = 64
V = 30522
N_VIS = 577
MAX_LEN = 20
OD_LEN = 50
BOS, EOS, PAD, MASK = 101, 102, 0, 103


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def to_torch(sd_np):
    """numpy state dict (vitcap_amd.weights.make_state_dict) -> torch fp32 tensors (shared storage kept)."""
    out, seen = {}, {}
    for k, v in sd_np.items():
        if id(v) not in seen:
            seen[id(v)] = torch.from_numpy(v)
        out[k] = seen[id(v)]
    return out


def gelu_erf(x):
    """src/layers/bert/activations.py:16-24 (`_gelu_python`) and nn.GELU in timm Mlp."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], eps)


# --------------------------------------------------------------------------------------------
# a1  patch embed + cls + pos   (timm vision_transformer.py:253-275, 411-426)
# --------------------------------------------------------------------------------------------
def patch_embed(sd, image):
    p = 'image_encoder.module.'
    x = F.conv2d(image, sd[p + 'patch_embed.proj.weight'], sd[p + 'patch_embed.proj.bias'], stride=16)
    x = x.flatten(2).transpose(1, 2)                                        # (B,576,768) row-major (h,w)
    cls = sd[p + 'cls_token'].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1) + sd[p + 'pos_embed']
    return x                                                                # blocks=[], norm=Identity


# --------------------------------------------------------------------------------------------
# a2-a4  ViT block   (timm vision_transformer.py:142-250; LayerNorm eps 1e-6 at :352)
# --------------------------------------------------------------------------------------------
def vit_attention(sd, p, x):
    B, N, C = x.shape
    qkv = _lin(sd, p + '.qkv', x).reshape(B, N, 3, HEADS, C // HEADS).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (HD ** -0.5)                         # mask is all-zero (bert:1415)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(sd, p + '.proj', x)


def vit_mlp(sd, p, x):
    return _lin(sd, p + '.fc2', F.gelu(_lin(sd, p + '.fc1', x)))


def vit_block(sd, p, x):
    x = x + vit_attention(sd, p + '.attn', _ln(sd, p + '.norm1', x, 1e-6))
    x = x + vit_mlp(sd, p + '.mlp', _ln(sd, p + '.norm2', x, 1e-6))
    return x


# --------------------------------------------------------------------------------------------
# a5  split encoder   (modeling_bert.py:458-478)
# --------------------------------------------------------------------------------------------
def split_encoder(sd, img_feats, depth=12, split_blocks=4):
    h = img_feats
    tag_h = None
    for i in range(depth):
        if i == depth - split_blocks:
            tag_h = h
        h = vit_block(sd, 'module.bert.encoder.blocks.%d' % i, h)
    for i in range(split_blocks):
        tag_h = vit_block(sd, 'module.bert.encoder.tag_blocks.%d' % i, tag_h)
    return h, tag_h


# --------------------------------------------------------------------------------------------
# a6 / a10  pooler + LM-style head   (modeling_bert.py:515-563, 651-658)
# --------------------------------------------------------------------------------------------
def pooler(sd, p, hidden):
    return torch.tanh(_lin(sd, p + '.dense', hidden[:, 0]))


def lm_head(sd, p, x):
    h = _lin(sd, p + '.predictions.transform.dense', x)
    h = gelu_erf(h)
    h = _ln(sd, p + '.predictions.transform.LayerNorm', h, 1e-12)
    return F.linear(h, sd[p + '.predictions.decoder.weight']) + sd[p + '.predictions.bias']


def tag_head(sd, tag_hidden, topk=50):
    """modeling_bert.py:1424-1432.  Returns (logit, prob_topk, pred_topk, topk_len)."""
    logit = lm_head(sd, 'module.bert.tag_logit', pooler(sd, 'module.bert.pooler', tag_hidden))
    prob, pred = torch.sigmoid(logit).topk(topk, dim=1, largest=True)
    return logit, prob, pred, (prob >= 0.2).sum(dim=1)


# --------------------------------------------------------------------------------------------
# a7  text embeddings   (modeling_bert.py:208-237, 1381-1406)
# --------------------------------------------------------------------------------------------
def bert_embeddings(sd, p, input_ids, position_ids=None, token_type_ids=None):
    if position_ids is None:
        position_ids = torch.arange(input_ids.shape[1]).unsqueeze(0).expand_as(input_ids)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = (sd[p + '.word_embeddings.weight'][input_ids]
         + sd[p + '.position_embeddings.weight'][position_ids]
         + sd[p + '.token_type_embeddings.weight'][token_type_ids])
    return _ln(sd, p + '.LayerNorm', e, 1e-12)


def encode_tag_to_embedding(sd, pred_topk, cls_emb_weight=None, caption_len=20):
    p = 'module.bert.embeddings'
    w = cls_emb_weight if cls_emb_weight is not None else sd[p + '.word_embeddings.weight']
    pos = (torch.arange(pred_topk.shape[1]) + caption_len).unsqueeze(0).expand_as(pred_topk)
    e = w[pred_topk] + sd[p + '.position_embeddings.weight'][pos] + sd[p + '.token_type_embeddings.weight'][0]
    return _ln(sd, p + '.LayerNorm', e, 1e-12)


# --------------------------------------------------------------------------------------------
# a9  post-LN BERT layer   (modeling_bert.py:275-437)
# --------------------------------------------------------------------------------------------
def bert_layer(sd, p, x, ext_mask, keep=None, p_drop=0.0, hid=None, p_hid=0.0):
    """``keep`` (B,12,S,S) bool + ``p_drop``: the training-mode dropout on the attention probabilities
    (modeling_bert.py:330-333) with the keep decisions made explicit.  ``hid`` = (m_attention_output, m_output), each a (B,S,768)
    multiplier (keep / (1 - p), hidden_keep_joint): nn.Dropout(hidden_dropout_prob) on the two dense outputs (BertSelfOutput :355,
    BertOutput :417)."""
    B, S, _ = x.shape

    def heads(t):
        return t.view(B, S, HEADS, HD).permute(0, 2, 1, 3)
    q = heads(_lin(sd, p + '.attention.self.query', x))
    k = heads(_lin(sd, p + '.attention.self.key', x))
    v = heads(_lin(sd, p + '.attention.self.value', x))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(HD)
    s = s + ext_mask
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep.to(pr.dtype) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p_drop)))
    ctx = torch.matmul(pr, v).permute(0, 2, 1, 3).contiguous().view(B, S, HID)
    ao = _lin(sd, p + '.attention.output.dense', ctx)
    if hid is not None:
        ao = ao * hid[0].to(ao.dtype)
    a = _ln(sd, p + '.attention.output.LayerNorm', ao + x, 1e-12)
    i = gelu_erf(_lin(sd, p + '.intermediate.dense', a))
    o = _lin(sd, p + '.output.dense', i)
    if hid is not None:
        o = o * hid[1].to(o.dtype)
    return _ln(sd, p + '.output.LayerNorm', o + a, 1e-12)


# --------------------------------------------------------------------------------------------
# a8  masks   (tagger_caption_uni_pipeline_expanding_bertemb.py:57-85; dataset.py:377-390 test mode)
# --------------------------------------------------------------------------------------------
def test_text_inputs(batch, max_len=MAX_LEN, od_len=OD_LEN, n_tag_visible=0):
    """What CaptionTensorizer emits at test time (dataset.py:218-219, 326, 377-390; notebook cell 15).

    ``n_tag_visible`` > 0: the mask ``tensorize_ab`` builds when a ``text_b`` of n tag tokens is attached (add_od_labels,
    dataset.py:240-252, 387-390): the n tag slots see each other, every caption row sees them (the shipped
    pipeline hard-codes text_b away, ..._bertemb.py:424)."""
    T = max_len + od_len
    input_ids = torch.zeros(batch, T, dtype=torch.long)
    input_ids[:, 0] = BOS
    input_ids[:, 1:max_len - 1] = MASK
    input_ids[:, max_len - 1] = EOS
    am = torch.zeros(T, T)
    am[:max_len, :max_len] = torch.tril(torch.ones(max_len, max_len))
    if n_tag_visible:
        n = int(n_tag_visible)
        am[max_len:max_len + n, max_len:max_len + n] = 1          # L-L
        am[:max_len, max_len:max_len + n] = 1                      # C-L
    attention_mask = am.unsqueeze(0).expand(batch, T, T).clone()
    return input_ids, attention_mask


def construct_attn_mask(attention_mask, num_img_feats):
    """seq2seq branch of ImageCaptioning.construct_attn_mask (..._bertemb.py:57-85)."""
    B, T, _ = attention_mask.shape
    top = torch.cat((attention_mask, torch.ones(B, T, num_img_feats)), dim=2)
    bottom = torch.cat((torch.zeros(B, num_img_feats, T), torch.ones(B, num_img_feats, num_img_feats)), dim=2)
    return torch.cat((top, bottom), dim=1)


# --------------------------------------------------------------------------------------------
# ViTSplitCLSEmbModel.forward as written   (modeling_bert.py:1408-1516)
# --------------------------------------------------------------------------------------------
def joint_forward(sd, input_ids, img_feats, attention_mask, position_ids, token_type_ids,
                  tagemb='cls', topk=50, enc=None, attn_keep=None, p_drop=0.0, hid_keep=None, p_hid=0.0):
    """Returns (sequence_output (B,S,768), tag_logit (B,V)).  ``enc`` may carry a precomputed
    (hidden, tag_hidden) pair -- the encoder is a pure function of img_feats."""
    input_ids = input_ids.clone()
    hidden, tag_hidden = enc if enc is not None else split_encoder(sd, img_feats)
    logit, prob, pred_topk, topk_len = tag_head(sd, tag_hidden, topk)
    pred_topk = pred_topk.clone()
    cls_w = sd['module.cls.predictions.decoder.weight']
    L = input_ids.shape[1]
    if int(topk_len[0]) + 20 <= L:                                           # :1435 branch A
        pred_topk[:, -1] = EOS
        emb = bert_embeddings(sd, 'module.bert.embeddings', input_ids, position_ids, token_type_ids)
        if tagemb == 'cls':
            tag_emb = cls_w[pred_topk]                                       # raw F.embedding (:1456)
        else:
            tag_emb = encode_tag_to_embedding(sd, pred_topk, None)
        emb[:, -pred_topk.shape[1]:] = tag_emb
    else:                                                                    # :1473 branch B
        start_id = L - topk_len
        input_ids[:, start_id] = EOS
        pred_topk[:, -1] = EOS
        if tagemb == 'cls':
            tag_emb = encode_tag_to_embedding(sd, pred_topk, cls_w)
        else:
            tag_emb = bert_embeddings(sd, 'module.bert.extra_embeddings', pred_topk,
                                      position_ids[:, -pred_topk.shape[1]:])
        emb = bert_embeddings(sd, 'module.bert.embeddings', input_ids, position_ids, token_type_ids)
        emb[:, -pred_topk.shape[1]:] = tag_emb
    enc_out = torch.cat([tag_hidden[:, 0, :].unsqueeze(1), hidden], 1)       # :1493
    am = torch.cat([attention_mask, attention_mask[:, -1].unsqueeze(1)], dim=1)
    am = torch.cat([am, torch.ones(am.shape[0], am.shape[1], 1)], dim=2)
    ext = (1.0 - am.unsqueeze(1)) * -10000.0                                 # :1498-1501
    if hid_keep is not None:              # BertEmbeddings.dropout on the text rows (:236); the encoder rows are concatenated after it
        emb = emb * hid_keep['emb'].to(emb.dtype)
    x = torch.cat((emb, enc_out), 1)
    for i in range(4):
        x = bert_layer(sd, 'module.bert.decoder.layer.%d' % i, x, ext,
                       None if attn_keep is None else attn_keep[i], p_drop,
                       None if hid_keep is None else (hid_keep['ao'][i], hid_keep['out'][i]), p_hid)
    return x, logit


def encode_forward_infer(sd, input_ids, img_feats, attention_mask, position_ids, token_type_ids,
                         tagemb='cls', enc=None):
    """ViTCAP.encode_forward(is_training=False): class logits on all text rows (modeling_bert.py:808-812)."""
    seq, _ = joint_forward(sd, input_ids, img_feats, attention_mask, position_ids, token_type_ids, tagemb, enc=enc)
    return lm_head(sd, 'module.cls', seq[:, :input_ids.shape[1]])


# --------------------------------------------------------------------------------------------
# a11/a12  greedy decode, as written   (modeling_bert.py:825-1001; modeling_utils.py:768-886)
# --------------------------------------------------------------------------------------------
def _remove_rows_cols(t, rs, re, cs, ce):
    t00, t01 = t[:, :rs, :cs], t[:, :rs, ce:]
    t10, t11 = t[:, re:, :cs], t[:, re:, ce:]
    return torch.cat([torch.cat([t00, t01], dim=2), torch.cat([t10, t11], dim=2)], dim=1)


def _eos_list(eos):
    """`eos_token_ids` of generate(): one id or a list; every entry ends a sequence, entry 0 is the one written when a sequence
    is closed by force (greedy: modeling_utils.py:862-871; beam: 1024-1026, 1096-1097)."""
    return [int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)]


def apply_repetition_penalty(logits, prefix_ids, penalty):
    """CTRL penalty as generate() applies it (modeling_utils.py:828-836, 955-963): for every distinct token of the
    sequence so far, logit < 0 ? logit * penalty : logit / penalty.  In place on (rows, V) logits."""
    if penalty == 1.0:
        return logits
    for i in range(logits.shape[0]):
        for tok in set(prefix_ids[i].tolist()):
            if logits[i, tok] < 0:
                logits[i, tok] *= penalty
            else:
                logits[i, tok] /= penalty
    return logits


def greedy_as_written(sd, image, tagemb='cls', max_length=MAX_LEN, od_labels_start_posid=20,
                      reuse_encoder=False, return_trace=False, repetition_penalty=1.0, eos=EOS, n_tag_visible=0, max_steps=None):
    """Reference greedy decode.  ``reuse_encoder=True`` computes the (step-invariant) ViT encoder
    once instead of 19 times -- same numbers, used only to keep CPU tests fast.  ``max_steps`` stops after that many
    decode steps (bench.py's bounded CPU-baseline sample; the returned caption is then a prefix)."""
    B = image.shape[0]
    img_feats = patch_embed(sd, image)
    input_ids0, am = test_text_inputs(B, max_length, n_tag_visible=n_tag_visible)
    full_mask = construct_attn_mask(am, img_feats.shape[1])
    od_label_ids = input_ids0[:, max_length:]
    od_len = od_label_ids.shape[1]
    od_start = max(od_labels_start_posid, max_length)
    pos = torch.cat([torch.arange(max_length), torch.arange(od_start, od_start + od_len)])
    full_pos = pos.unsqueeze(0).expand(B, -1)
    full_tt = torch.zeros(B, max_length + od_len, dtype=torch.long)
    enc = split_encoder(sd, img_feats) if reuse_encoder else None

    ids = torch.full((B, 1), BOS, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    logprobs, unf_hist, trace = [], [], []
    cur_len = 1
    while cur_len < max_length:
        step_ids = torch.cat([ids, torch.full((B, 1), MASK, dtype=torch.long)], dim=1)
        curr = step_ids.shape[1]
        mask = _remove_rows_cols(full_mask, curr, max_length, curr, max_length)
        tt = torch.cat([full_tt[:, :curr], full_tt[:, max_length:]], dim=1)
        pp = torch.cat([full_pos[:, :curr], full_pos[:, max_length:]], dim=1)
        step_ids = torch.cat([step_ids, od_label_ids], dim=1)
        logits = encode_forward_infer(sd, step_ids, img_feats, mask, pp, tt, tagemb, enc=enc)
        nxt_logits = apply_repetition_penalty(logits[:, cur_len, :].clone(), ids, repetition_penalty)
        nxt = torch.argmax(nxt_logits, dim=-1)
        sc = torch.gather(F.log_softmax(nxt_logits, dim=-1), -1, nxt.unsqueeze(-1))
        if return_trace:
            top2 = nxt_logits.topk(2, dim=-1).values
            trace.append({'logits_row': nxt_logits.clone(), 'margin': (top2[:, 0] - top2[:, 1]).clone()})
        logprobs.append(sc)
        unf_hist.append(unfinished)
        add = nxt * unfinished + PAD * (1 - unfinished)
        ids = torch.cat([ids, add.unsqueeze(-1)], dim=-1)
        for eid in _eos_list(eos):
            unfinished = unfinished * add.ne(eid).long()
        cur_len += 1
        if unfinished.max() == 0:
            break
        if max_steps is not None and cur_len - 1 >= max_steps:
            break
    if cur_len == max_length:
        ids[:, -1].masked_fill_(unfinished.bool(), _eos_list(eos)[0])
    lp = torch.cat(logprobs, dim=1)
    uh = torch.stack(unf_hist, dim=1).float()
    lp = (lp * uh).sum(dim=1) / uh.sum(dim=1)
    if ids.shape[1] < max_length:
        ids = torch.cat([ids, ids.new_full((B, max_length - ids.shape[1]), PAD)], dim=1)
    out = (ids.unsqueeze(1), lp.unsqueeze(1))
    return out + (trace,) if return_trace else out


# --------------------------------------------------------------------------------------------
# incremental formulation (what the HIP path computes); optional bf16 rounding emulation
# --------------------------------------------------------------------------------------------
class _R:
    """Rounding policy: identity (fp32) or round-to-nearest-even to bfloat16 and back."""

    def __init__(self, on):
        self.on = on

    def __call__(self, t):
        return t.to(torch.bfloat16).to(torch.float32) if self.on else t


def _rw(sd, r):
    """Weights as the device holds them: matrices/embeddings in bf16, biases and LN params fp32."""
    if not r.on:
        return sd
    out, seen = {}, {}
    for k, v in sd.items():
        if v.dim() >= 2 and k.endswith('.weight'):
            if id(v) not in seen:
                seen[id(v)] = r(v)
            out[k] = seen[id(v)]
        else:
            out[k] = v
    return out


def encoder_incremental(sdw, image, r):
    """Device pipeline for a1-a5 with its rounding points.

    residual stream x: fp32.  LN output, qkv, attention output, GELU output: bf16.
    GEMMs: bf16 operands, fp32 accumulate, fp32 bias/residual epilogue.
    Softmax: scores fp32, P rounded to bf16 before P.V, row sum accumulated from the same rounded P (dense kernel).
    """
    p = 'image_encoder.module.'
    B = image.shape[0]
    img = r(image)
    patches = img.reshape(B, 3, 24, 16, 24, 16).permute(0, 2, 4, 1, 3, 5).reshape(B, 576, 768)
    w = sdw[p + 'patch_embed.proj.weight'].reshape(768, 768)
    x = patches @ w.t() + sdw[p + 'patch_embed.proj.bias']
    x = torch.cat((sdw[p + 'cls_token'].expand(B, -1, -1), x), dim=1) + sdw[p + 'pos_embed']

    def block(pref, x):
        h = r(_ln(sdw, pref + '.norm1', x, 1e-6))
        qkv = r(_lin(sdw, pref + '.attn.qkv', h))
        o = r(attn_rounded(qkv, x.shape[1], r))
        x = x + _lin(sdw, pref + '.attn.proj', o)
        h = r(_ln(sdw, pref + '.norm2', x, 1e-6))
        g = r(F.gelu(_lin(sdw, pref + '.mlp.fc1', h)))
        return x + _lin(sdw, pref + '.mlp.fc2', g)

    tag = None
    for i in range(12):
        if i == 8:
            tag = x
        x = block('module.bert.encoder.blocks.%d' % i, x)
    for i in range(4):
        tag = block('module.bert.encoder.tag_blocks.%d' % i, tag)
    return x, tag


C_LOG2 = float(torch.tensor(0.125, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32))


def softmax_pv_rounded(s_raw, v, r, sum_rounded=False):
    """Device softmax.V in the log2 domain: m = ceil(max(s) * c) with c = fp32(0.125*log2 e) (row max rounded UP to
    an integer so every online-softmax rescale is an exact power of two), P = exp2(fma(s, c, -m)) -- the fused
    multiply-add is reproduced exactly by doing it in float64 -- bf16(P) for the product.  Row sum: from the unrounded
    P in the decode-step kernels (vector ALU), from bf16(P) in the dense kernel (`sum_rounded`: it sums through the
    matrix pipe, ones . P^T, csrc/attn.hip).  Mathematically softmax(s/8) @ v (modeling_bert.py:320-336)."""
    c32 = torch.tensor(C_LOG2, dtype=torch.float32)
    m = torch.ceil(s_raw.max(dim=-1, keepdim=True).values * c32)
    t = (s_raw.double() * float(C_LOG2) - m.double()).float()
    e = torch.exp2(t)
    den = r(e).sum(dim=-1, keepdim=True) if sum_rounded else e.sum(dim=-1, keepdim=True)
    return (r(e) @ v) / den


def attn_rounded(qkv, S, r):
    """softmax(QK^T/8)V over packed (B,S,2304) qkv with the device's rounding points."""
    B = qkv.shape[0]
    q, k, v = qkv.view(B, S, 3, HEADS, HD).permute(2, 0, 3, 1, 4)
    o = softmax_pv_rounded(q @ k.transpose(-1, -2), v, r, sum_rounded=True)
    return o.transpose(1, 2).reshape(B, S, HID)


def tag_head_rounded(sdw, tag_hidden, r, topk=50):
    """a6 with the device rounding points (engine.cpp encode tail)."""
    p = 'module.bert.tag_logit.predictions'
    pooled = r(torch.tanh(_lin(sdw, 'module.bert.pooler.dense', r(tag_hidden[:, 0]))))
    h = _ln(sdw, p + '.transform.LayerNorm', gelu_erf(_lin(sdw, p + '.transform.dense', pooled)), 1e-12)
    logit = F.linear(r(h), sdw[p + '.decoder.weight']) + sdw[p + '.bias']
    prob, pred = torch.sigmoid(logit).topk(topk, dim=1, largest=True)
    return logit, prob, pred, (prob >= 0.2).sum(dim=1)


def _dec_w(sdw, i):
    p = 'module.bert.decoder.layer.%d' % i
    wqkv = torch.cat([sdw[p + '.attention.self.%s.weight' % n] for n in ('query', 'key', 'value')], 0)
    bqkv = torch.cat([sdw[p + '.attention.self.%s.bias' % n] for n in ('query', 'key', 'value')], 0)
    return p, wqkv, bqkv


def _post(sdw, p, ctx, x, r):
    """BertSelfOutput + BertIntermediate + BertOutput on rows (..., 768) with device rounding points."""
    a = _ln(sdw, p + '.attention.output.LayerNorm', _lin(sdw, p + '.attention.output.dense', r(ctx)) + x, 1e-12)
    i = r(gelu_erf(_lin(sdw, p + '.intermediate.dense', r(a))))
    return _ln(sdw, p + '.output.LayerNorm', _lin(sdw, p + '.output.dense', i) + a, 1e-12)


def greedy_incremental(sd, image, emulate_bf16=False, max_length=MAX_LEN, return_trace=False, sampler=None,
                       repetition_penalty=1.0, eos=EOS):
    """Greedy caption via encoder-once + visual prefill + 2-row incremental steps.

    Equivalent to ``greedy_as_written`` under the shipped test mask: caption row i attends caption
    rows <= i and all 578 visual rows; visual rows attend visual rows only; the 50 tag slots are
    attended by nothing and their outputs are unused (SURVEY.md headline 5), so they are skipped.
    """
    r = _R(emulate_bf16)
    sdw = _rw(sd, r)
    B = image.shape[0]
    hidden, tag_hidden = encoder_incremental(sdw, image, r)
    tags = tag_head_rounded(sdw, tag_hidden, r)
    vis = torch.cat([tag_hidden[:, :1], hidden], dim=1)                      # (B,578,768) fp32
    S = vis.shape[1]
    # ---- prefill: visual rows through the 4 decoder layers, keep per-layer K/V --------------
    kv = []
    x = vis
    for i in range(4):
        p, wqkv, bqkv = _dec_w(sdw, i)
        qkv = r(F.linear(r(x), wqkv, bqkv))
        kv.append((qkv[..., 768:1536].clone(), qkv[..., 1536:].clone()))
        ctx = attn_rounded(qkv, S, r)
        x = _post(sdw, p, ctx, x, r)
    # ---- steps -------------------------------------------------------------------------------
    e = 'module.bert.embeddings'
    ids = torch.full((B, max_length), PAD, dtype=torch.long)
    ids[:, 0] = BOS
    unf = torch.ones(B, dtype=torch.long)
    sum_lp = torch.zeros(B)
    cnt = torch.zeros(B)
    tk = [torch.zeros(B, max_length, HID) for _ in range(4)]                 # text K cache per layer
    tv = [torch.zeros(B, max_length, HID) for _ in range(4)]
    trace = []
    for t in range(1, max_length):
        tok = torch.stack([ids[:, t - 1], torch.full((B,), MASK, dtype=torch.long)], dim=1)   # (B,2)
        pos = torch.tensor([t - 1, t])
        x = (sdw[e + '.word_embeddings.weight'][tok] + sdw[e + '.position_embeddings.weight'][pos]
             + sdw[e + '.token_type_embeddings.weight'][0])
        x = _ln(sdw, e + '.LayerNorm', x, 1e-12)                             # (B,2,768) fp32
        for i in range(4):
            p, wqkv, bqkv = _dec_w(sdw, i)
            qkv = r(F.linear(r(x), wqkv, bqkv))                              # (B,2,2304)
            q = qkv[..., :768].view(B, 2, HEADS, HD).transpose(1, 2)         # (B,H,2,64)
            tk[i][:, t - 1] = qkv[:, 0, 768:1536]
            tv[i][:, t - 1] = qkv[:, 0, 1536:]
            # keys: visual 578, cached text rows 0..t-1, MASK row (visible to itself only)
            K = torch.cat([kv[i][0], tk[i][:, :t], qkv[:, 1:2, 768:1536]], dim=1)
            Vv = torch.cat([kv[i][1], tv[i][:, :t], qkv[:, 1:2, 1536:]], dim=1)
            Kh = K.view(B, -1, HEADS, HD).transpose(1, 2)
            Vh = Vv.view(B, -1, HEADS, HD).transpose(1, 2)
            s = q @ Kh.transpose(-1, -2)                                       # (B,H,2,S+t+1) raw dot products
            s[:, :, 0, -1] = float('-inf')                                   # row t-1 cannot see MASK row t
            ctx = softmax_pv_rounded(s, Vh, r)
            ctx = ctx.transpose(1, 2).reshape(B, 2, HID)
            x = _post(sdw, p, ctx, x, r)
        hrow = x[:, 1]                                                       # MASK row
        c = 'module.cls.predictions'
        h = _ln(sdw, c + '.transform.LayerNorm', gelu_erf(_lin(sdw, c + '.transform.dense', r(hrow))), 1e-12)
        logits = F.linear(r(h), sdw[c + '.decoder.weight']) + sdw[c + '.bias']
        apply_repetition_penalty(logits, ids[:, :t], repetition_penalty)
        if sampler is not None:                                                # do_sample branch (a12)
            nxt, lp, smargin = sampler(logits, t)
            if return_trace:
                trace.append({'logits_row': logits.clone(), 'margin': smargin})
        else:
            nxt = torch.argmax(logits, dim=-1)
            lp = torch.gather(F.log_softmax(logits, dim=-1), -1, nxt.unsqueeze(-1)).squeeze(-1)
            if return_trace:
                top2 = logits.topk(2, dim=-1).values
                trace.append({'logits_row': logits.clone(), 'margin': (top2[:, 0] - top2[:, 1]).clone()})
        sum_lp += lp * unf
        cnt += unf
        add = nxt * unf + PAD * (1 - unf)
        ids[:, t] = add
        for eid in _eos_list(eos):
            unf = unf * add.ne(eid).long()
    ids[:, -1].masked_fill_(unf.bool(), _eos_list(eos)[0])
    out = (ids.unsqueeze(1), (sum_lp / cnt).unsqueeze(1))
    if return_trace:
        return out + ({'steps': trace, 'hidden': hidden, 'tag_hidden': tag_hidden, 'tags': tags},)
    return out



# --------------------------------------------------------------------------------------------
# a12  sampling branch   (modeling_utils.py:839-851, top_k_top_p_filtering 1103-1135)
def top_k_top_p_filter(logits, top_k=0, top_p=1.0, min_tokens_to_keep=1):
    """Returns a copy of ``logits`` (B,V) with every filtered entry set to -inf.

    top-k: k = max(top_k, min_tokens_to_keep); everything strictly below the k-th largest value goes (ties with it stay).
    top-p: rank descending, p = softmax over the row as it stands after top-k; entry of rank i goes iff the
    cumulative probability of ranks 0..i-1 exceeds top_p (so the entry that crosses the threshold stays and rank 0
    always stays).  With min_tokens_to_keep = n > 1 (the beam-sampling call, modeling_utils.py:970-972) the reference clears
    the removal flags of ranks 0..n-1 BEFORE shifting them right by one (1125-1130), so ranks 0..n always stay: n + 1 tokens."""
    x = logits.clone()
    V = x.shape[-1]
    if top_k > 0:
        k = min(max(int(top_k), int(min_tokens_to_keep), 1), V)
        kth = torch.topk(x, k, dim=-1).values[..., -1:]
        x = torch.where(x < kth, torch.full_like(x, float('-inf')), x)
    if top_p < 1.0:
        order = torch.argsort(x, dim=-1, descending=True)
        ranked = torch.gather(x, -1, order)
        cum = torch.cumsum(F.softmax(ranked, dim=-1), dim=-1)
        before = torch.cat([torch.zeros_like(cum[..., :1]), cum[..., :-1]], dim=-1)   # mass of the better ranks
        drop_ranked = before > top_p
        drop_ranked[..., 0] = False
        if min_tokens_to_keep > 1:
            drop_ranked[..., :min_tokens_to_keep + 1] = False
        drop = torch.zeros_like(drop_ranked).scatter(-1, order, drop_ranked)
        x = torch.where(drop, torch.full_like(x, float('-inf')), x)
    return x


def _lowbias32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def rng_mix(h, v):
    """Counter hash of vitcap_amd/csrc/rng.h (vc_mix), vectorised over uint32 arrays."""
    h = np.asarray(h, dtype=np.uint32)
    v = np.asarray(v, dtype=np.uint32)
    with np.errstate(over='ignore'):
        return _lowbias32(h ^ (v + np.uint32(0x9e3779b9) + (h << np.uint32(6)) + (h >> np.uint32(2))))


def rng_uniform(r):
    return ((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)


def gumbel_noise(seed, b, t, V):
    h = rng_mix(rng_mix(np.uint32(seed & 0xffffffff), np.uint32(b)), np.uint32(t))
    u = rng_uniform(rng_mix(h, np.arange(V, dtype=np.uint32)))
    return -np.log(-np.log(u.astype(np.float32))).astype(np.float32)


def make_sampler(temperature=1.0, top_k=0, top_p=1.0, seed=0):
    """sampler(logits (B,V), t) -> (token (B,), logprob (B,), margin (B,)).

    The reference draws with torch.multinomial(softmax(filtered)); its generator stream cannot be reproduced by
    another implementation, so the draw is restated as the Gumbel-max form of the same distribution,
    argmax(filtered + G), G = -log(-log(u)), u a counter-based uniform of (seed, b, t, column):
    P[argmax = i] = softmax(filtered)_i exactly (tests/test_oracle_golden.py checks this against the softmax
    frequencies).  The log-prob is taken on the filtered, temperature-scaled logits as modeling_utils.py:850-851."""
    def sampler(logits, t):
        x = logits / temperature if temperature != 1.0 else logits
        x = top_k_top_p_filter(x, top_k, top_p)
        B, V = x.shape
        g = torch.from_numpy(np.stack([gumbel_noise(seed, b, t, V) for b in range(B)]))
        sc = x + g
        top2 = sc.topk(2, dim=-1)
        tok = top2.indices[:, 0]
        lp = torch.gather(F.log_softmax(x, dim=-1), -1, tok.unsqueeze(-1)).squeeze(-1)
        return tok, lp, (top2.values[:, 0] - top2.values[:, 1])
    return sampler


def sample_incremental(sd, image, temperature=1.0, top_k=0, top_p=1.0, seed=0, emulate_bf16=False,
                       return_trace=False):
    return greedy_incremental(sd, image, emulate_bf16=emulate_bf16, return_trace=return_trace,
                              sampler=make_sampler(temperature, top_k, top_p, seed))

# --------------------------------------------------------------------------------------------
# a13  beam search   (modeling_utils.py:888-1100, BeamHypotheses 1138-1180)
# --------------------------------------------------------------------------------------------
class BeamHypotheses:
    """n-best list exactly as modeling_utils.py:1138-1180 (n_hyp = num_keep_best, max_length-1, length_penalty).
    ``margin`` tracks the smallest gap of any comparison that decided something (test conditioning only)."""

    def __init__(self, n_hyp, max_length, length_penalty):
        self.max_length = max_length - 1
        self.length_penalty = length_penalty
        self.n_hyp = n_hyp
        self.hyp = []
        self.worst_score = 1e9
        self.margin = float('inf')

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self.hyp) >= self.n_hyp:
            self.margin = min(self.margin, abs(score - self.worst_score))
        if len(self.hyp) < self.n_hyp or score > self.worst_score:
            self.hyp.append((score, hyp))
            if len(self.hyp) > self.n_hyp:
                ss = sorted([(sc, idx) for idx, (sc, _) in enumerate(self.hyp)])
                del self.hyp[ss[0][1]]
                self.worst_score = ss[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs):
        if len(self.hyp) < self.n_hyp:
            return False
        bound = best_sum_logprobs / self.max_length ** self.length_penalty
        self.margin = min(self.margin, abs(self.worst_score - bound))
        return self.worst_score >= bound


def gumbel_top2(x, seed, t, row0=0):
    """Two draws WITHOUT replacement from softmax(x) per row (what torch.multinomial(p, num_samples=2) returns, first draw
    first) in Gumbel-top-k form: the two largest of x + G, G the counter-based noise of row (row0 + r), step t."""
    R, V = x.shape
    g = torch.from_numpy(np.stack([gumbel_noise(seed, row0 + r, t, V) for r in range(R)]))
    return (x + g).topk(2, dim=-1).indices


def beam_sample_candidates(logits, beam_scores, B, K, t, temperature=1.0, top_k=0, top_p=1.0, seed=0, draw=None):
    """The do_sample branch of _generate_beam_search (modeling_utils.py:966-985): per beam row temperature, the top-k /
    top-p filter with min_tokens_to_keep = 2, TWO words drawn without replacement, score = log_softmax(filtered)[word] + the
    row's beam score.  "Match shape of greedy beam search" then views the (B*K, 2) draws as (B, 2K) -- position p holds draw
    p % 2 of beam p // 2 -- but adds `arange(K) * V` REPEATED twice as the beam offset, so position p is attributed to beam
    p % K.  Restated as written: the hypothesis that continues with the word at position p is beam p % K's prefix, its score
    comes from beam p // 2.  The candidates are consumed in position order (nothing sorts them)."""
    x = logits / temperature if temperature != 1.0 else logits
    x = top_k_top_p_filter(x, top_k, top_p, min_tokens_to_keep=2)
    words = draw(x, t) if draw is not None else gumbel_top2(x, seed, t)          # (B*K, 2)
    Vn = x.shape[1]
    sc = torch.gather(F.log_softmax(x, dim=-1), -1, words) + beam_scores[:, None]
    off = (torch.arange(K) * Vn).repeat(B, 2)
    return sc.reshape(B, 2 * K), words.reshape(B, 2 * K) + off


def beam_bookkeeping(step_logits_fn, B, num_beams, max_length=MAX_LEN, length_penalty=1.0, num_keep_best=1,
                     repetition_penalty=1.0, eos=EOS, return_margins=False, sample=None):
    """The reference's beam driver around an abstract model: ``step_logits_fn(input_ids (B*beams, cur_len), beam_idx)``
    returns the next-token logits (B*beams, V) for the current prefixes (``beam_idx`` = the re-ordering applied since
    the previous call, None at the first).  Returns (decoded (B,keep,max_length), logprobs (B,keep)).

    With ``return_margins`` also (B, max_length-1): per image and step the smallest score gap among the comparisons that
    decided the step's outcome -- the gap between the last candidate the scan consumed and the next one (a swap there
    changes which beams survive / which hypotheses finish), every ``BeamHypotheses.add`` / ``is_done`` comparison, and at
    the end the gaps between the kept hypotheses' final scores.  Order WITHIN the consumed candidates does not matter (the
    surviving set and the hypothesis scores are the same).  inf for steps of images that were already done."""
    K = num_beams
    eos_ids = _eos_list(eos)
    input_ids = torch.full((B * K, 1), BOS, dtype=torch.long)
    hyps = [BeamHypotheses(num_keep_best, max_length, length_penalty) for _ in range(B)]
    beam_scores = torch.zeros(B, K)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    done = [False] * B
    cur_len = 1
    beam_idx = None
    margins = torch.full((B, max_length - 1), float('inf'))
    while cur_len < max_length:
        logits = apply_repetition_penalty(step_logits_fn(input_ids, beam_idx).clone(), input_ids, repetition_penalty)
        Vn = logits.shape[1]
        if sample is not None:           # do_sample with beams: ``sample`` = kwargs of beam_sample_candidates
            assert not return_margins
            next_scores, next_words = beam_sample_candidates(logits, beam_scores, B, K, cur_len, **sample)
        else:
            scores = F.log_softmax(logits, dim=-1)
            _scores = (scores + beam_scores[:, None]).view(B, K * Vn)
            next_scores, next_words = torch.topk(_scores, 2 * K, dim=1, largest=True, sorted=True)
            ext = torch.topk(_scores, 2 * K + 1, dim=1, largest=True, sorted=True).values if return_margins else None
        nb = []
        for b in range(B):
            hyps[b].margin = float('inf')
            done[b] = done[b] or hyps[b].is_done(next_scores[b].max().item())
            if done[b]:
                nb.extend([(0, PAD, 0)] * K)
                margins[b, cur_len - 1] = hyps[b].margin
                continue
            sent = []
            used = 0
            for idx, sc in zip(next_words[b], next_scores[b]):
                beam_id = int(idx) // Vn
                word_id = int(idx) % Vn
                used += 1
                if word_id in eos_ids or cur_len + 1 == max_length:
                    hyps[b].add(input_ids[b * K + beam_id, :cur_len].clone(), sc.item())
                else:
                    sent.append((sc, word_id, b * K + beam_id))
                if len(sent) == K:
                    break
            if return_margins:
                m = hyps[b].margin
                if cur_len + 1 < max_length:         # at the last step every candidate becomes a hypothesis: no boundary
                    m = min(m, float(ext[b, used - 1] - ext[b, used]))
                margins[b, cur_len - 1] = m
            if len(sent) == 0:
                sent = [(0, PAD, 0)] * K
            nb.extend(sent)
        beam_scores = torch.tensor([float(x[0]) for x in nb])
        beam_words = torch.tensor([int(x[1]) for x in nb], dtype=torch.long)
        beam_idx = torch.tensor([int(x[2]) for x in nb], dtype=torch.long)
        input_ids = torch.cat([input_ids[beam_idx, :], beam_words.unsqueeze(1)], dim=-1)
        cur_len += 1
        if all(done):
            break
    decoded = torch.full((B, num_keep_best, max_length), PAD, dtype=torch.long)
    logprobs = torch.full((B, num_keep_best), -1e5)
    for i, h in enumerate(hyps):
        hs = torch.tensor([x[0] for x in h.hyp])
        _, best = torch.topk(hs, min(num_keep_best, len(hs)), largest=True)
        for bi, hi in enumerate(best):
            conf, hyp = h.hyp[int(hi)]
            logprobs[i, bi] = conf
            decoded[i, bi, :len(hyp)] = hyp
            decoded[i, bi, len(hyp)] = eos_ids[0]
        if len(hs) > 1:
            srt = torch.sort(hs, descending=True).values
            margins[i, -1] = min(float(margins[i, -1]), float((srt[:-1] - srt[1:]).min()))
    if return_margins:
        return decoded, logprobs, margins
    return decoded, logprobs


def _as_written_stepper(sd, image, K, tagemb='cls', max_length=MAX_LEN):
    """``step(input_ids (B*K, cur_len), beam_idx)`` -> next-token logits (B*K, V) with the model re-run on the full prefix
    (past=None: ``_do_output_past`` is False for this model, modeling_bert.py:1072), K sequences per image."""
    B = image.shape[0]
    img_feats = patch_embed(sd, image)
    enc1 = split_encoder(sd, img_feats)
    input_ids0, am = test_text_inputs(B, max_length)
    full_mask = construct_attn_mask(am, img_feats.shape[1])
    rep = lambda x: x.unsqueeze(1).expand(x.shape[0], K, *x.shape[1:]).reshape(x.shape[0] * K, *x.shape[1:])
    img_feats_k, full_mask_k = rep(img_feats), rep(full_mask)
    enc_k = (rep(enc1[0]), rep(enc1[1]))
    od = rep(input_ids0[:, max_length:])
    pos_full = torch.cat([torch.arange(max_length), torch.arange(20, 20 + OD_LEN)])

    def step(input_ids, beam_idx):
        cur = input_ids.shape[1]
        ids = torch.cat([input_ids, torch.full((B * K, 1), MASK, dtype=torch.long), od], dim=1)
        curr = cur + 1
        mask = _remove_rows_cols(full_mask_k, curr, max_length, curr, max_length)
        pp = torch.cat([pos_full[:curr], pos_full[max_length:]]).unsqueeze(0).expand(B * K, -1)
        tt = torch.zeros(B * K, curr + OD_LEN, dtype=torch.long)
        logits = encode_forward_infer(sd, ids, img_feats_k, mask, pp, tt, tagemb, enc=enc_k)
        return logits[:, cur, :]
    return step


def beam_as_written(sd, image, num_beams, tagemb='cls', max_length=MAX_LEN, length_penalty=1.0, num_keep_best=1,
                    repetition_penalty=1.0, eos=EOS, sample=None):
    """Reference beam search with the model re-run on the full prefix at every step (past=None)."""
    step = _as_written_stepper(sd, image, num_beams, tagemb, max_length)
    return beam_bookkeeping(step, image.shape[0], num_beams, max_length, length_penalty, num_keep_best, repetition_penalty,
                            eos=eos, sample=sample)


def _incremental_stepper(sd, image, K, emulate_bf16=False, max_length=MAX_LEN, trace=None):
    """``step(input_ids (B*K, cur_len), beam_idx)`` on the incremental formulation (what the HIP path computes): encoder and
    visual prefill once per image, per-sequence text K/V caches re-ordered by ``beam_idx`` (the parents chosen since the
    previous call), K sequences per image."""
    r = _R(emulate_bf16)
    sdw = _rw(sd, r)
    B = image.shape[0]
    hidden, tag_hidden = encoder_incremental(sdw, image, r)
    vis = torch.cat([tag_hidden[:, :1], hidden], dim=1)
    S = vis.shape[1]
    kv = []
    x = vis
    for i in range(4):
        p, wqkv, bqkv = _dec_w(sdw, i)
        qkv = r(F.linear(r(x), wqkv, bqkv))
        kv.append((qkv[..., 768:1536].clone(), qkv[..., 1536:].clone()))
        if i == 3:
            break
        x = _post(sdw, p, attn_rounded(qkv, S, r), x, r)
    rep = lambda t: t.unsqueeze(1).expand(B, K, *t.shape[1:]).reshape(B * K, *t.shape[1:])
    kvk = [(rep(k), rep(v)) for k, v in kv]
    e = 'module.bert.embeddings'
    N = B * K
    tk = [torch.zeros(N, max_length, HID) for _ in range(4)]
    tv = [torch.zeros(N, max_length, HID) for _ in range(4)]

    def step(input_ids, beam_idx):
        t = input_ids.shape[1]
        if beam_idx is not None:
            for i in range(4):
                tk[i][:] = tk[i][beam_idx]
                tv[i][:] = tv[i][beam_idx]
        tok = torch.stack([input_ids[:, t - 1], torch.full((N,), MASK, dtype=torch.long)], dim=1)
        pos = torch.tensor([t - 1, t])
        x = (sdw[e + '.word_embeddings.weight'][tok] + sdw[e + '.position_embeddings.weight'][pos]
             + sdw[e + '.token_type_embeddings.weight'][0])
        x = _ln(sdw, e + '.LayerNorm', x, 1e-12)
        for i in range(4):
            p, wqkv, bqkv = _dec_w(sdw, i)
            qkv = r(F.linear(r(x), wqkv, bqkv))
            q = qkv[..., :768].view(N, 2, HEADS, HD).transpose(1, 2)
            tk[i][:, t - 1] = qkv[:, 0, 768:1536]
            tv[i][:, t - 1] = qkv[:, 0, 1536:]
            Kc = torch.cat([kvk[i][0], tk[i][:, :t], qkv[:, 1:2, 768:1536]], dim=1)
            Vc = torch.cat([kvk[i][1], tv[i][:, :t], qkv[:, 1:2, 1536:]], dim=1)
            Kh = Kc.view(N, -1, HEADS, HD).transpose(1, 2)
            Vh = Vc.view(N, -1, HEADS, HD).transpose(1, 2)
            sc = q @ Kh.transpose(-1, -2)
            sc[:, :, 0, -1] = float('-inf')
            ctx = softmax_pv_rounded(sc, Vh, r).transpose(1, 2).reshape(N, 2, HID)
            x = _post(sdw, p, ctx, x, r)
        c = 'module.cls.predictions'
        h = _ln(sdw, c + '.transform.LayerNorm', gelu_erf(_lin(sdw, c + '.transform.dense', r(x[:, 1]))), 1e-12)
        logits = F.linear(r(h), sdw[c + '.decoder.weight']) + sdw[c + '.bias']
        if trace is not None:
            trace.append(logits.clone())
        return logits
    return step


def beam_incremental(sd, image, num_beams, emulate_bf16=False, max_length=MAX_LEN, return_trace=False, length_penalty=1.0,
                     num_keep_best=1, repetition_penalty=1.0, eos=EOS, return_margins=False, sample=None):
    """Beam search on the incremental formulation (what the HIP path computes): encoder and visual prefill once per
    image, per-sequence text K/V caches re-ordered by the chosen parent beams."""
    trace = [] if return_trace else None
    step = _incremental_stepper(sd, image, num_beams, emulate_bf16, max_length, trace)
    out = beam_bookkeeping(step, image.shape[0], num_beams, max_length, length_penalty, num_keep_best, repetition_penalty, eos=eos,
                           return_margins=return_margins, sample=sample)
    return out + (trace,) if return_trace else out


# --------------------------------------------------------------------------------------------
# f4  constrained beam search   (src/tools/captioning/utils_cbs.py:26-374 search, 377-443 selection, 646-871 FSM builder;
#     hooks in ViTCAP.generate modeling_bert.py:949-953, 994, 1035-1057; step = _decode_step modeling_utils.py:747-766)
# --------------------------------------------------------------------------------------------
CBS_MASKED = -1e20          # utils_cbs.py:240: a transition the FSM does not allow scores -1e20 (NOT -inf)


def cbs_search(step_logits_fn, fsm, num_beams, max_length=MAX_LEN, eos=EOS, return_margins=False, no_repeat=False,
               bad_ending_ids=None):
    """``ConstrainedBeamSearch.search`` as ViTCAP.generate drives it (use_hypo=False, no decoding_constraint_flag, no
    bad_ending_ids, per_node_beam_size = beam_size, state = None because the model returns no past).

    fsm (B, S, S, V) uint8: fsm[b, s1, s2, w] = 1 iff word w moves image b's machine from state s1 to s2.  Every image carries
    S * K sequences, slot (s, k) = beam k of state s; ``step_logits_fn(curr_ids (B*S*K, cur_len), parents)`` returns the
    next-token logits, ``parents`` = the global slot each slot's prefix was copied from since the previous call (None first).

    * first step (:127-152): log-softmax of the FIRST B ROWS of the expanded batch -- ``[:batch_size]`` on an image-major
      (B*S*K, V) tensor, i.e. row b belongs to image b // (S*K): for B > 1 every image starts from image 0's distribution
      (restated as written); words not allowed out of state 0 into state i are -inf; K best words per state.
    * every later step (:169-319): a slot whose last word is an EOS id may only continue with an EOS id at cost 0; for target
      state i the candidates of slot (s, k) are its words masked by fsm[b, s, i] (-1e20 where not allowed), K best per slot,
      plus the slot's running score, K best overall -> slot (i, 0..K-1), parent = the candidate's slot.  The loop stops early
      once every slot of the batch ends in EOS.
    * ``no_repeat`` = generate's ``decoding_constraint_flag`` (:187-190): a live slot may not repeat its last word (-inf);
      ``bad_ending_ids`` (:192-198): a live slot whose last word is in the list may not end (every EOS id -inf).  Both act on the
      log-probabilities before the finished-slot override and not on the first step.
    Ties (e.g. among the -1e20 fillers of a state fewer than K allowed candidates reach) fall as torch.topk leaves them; the
    device breaks them by lowest flat index, tests use constraints whose valid states never depend on that.

    Returns (beams (B, S, K, T) without the BOS column, T <= max_length - 1, scores (B, S, K)).  ``return_margins``: also
    (B, S) the smallest gap, over the steps, between neighbours among the K+1 best FINITE candidates of a state (inf if
    fewer than two)."""
    B, S, _, Vn = fsm.shape
    K = num_beams
    G = S * K
    eos_ids = _eos_list(eos)
    allowed = fsm.to(torch.bool)
    NEG = float('-inf')
    margins = torch.full((B, S), float('inf'))

    curr = torch.full((B * G, 1), BOS, dtype=torch.long)
    lp0 = F.log_softmax(step_logits_fn(curr, None), dim=-1)[:B]
    start = lp0[:, None, :].expand(B, S, Vn).masked_fill(~allowed[:, 0], NEG)
    last, words = start.topk(K)                                        # (B, S, K): running scores, first words
    if return_margins:
        margins = _finite_gap(start.topk(K + 1).values, K)
    preds, backs = [words.reshape(B, G)], []
    after_end = torch.full((Vn,), NEG)
    after_end[eos_ids] = 0.0
    base = (torch.arange(B) * G)[:, None]
    for _ in range(max_length - 2):
        lastw = preds[-1].reshape(-1)
        fin = torch.zeros_like(lastw, dtype=torch.bool)
        for e in eos_ids:
            fin |= lastw == e
        if bool(fin.all()):
            break
        curr = torch.cat([curr, lastw[:, None]], dim=1)
        parents = (backs[-1] + base).reshape(-1) if backs else None
        lp = F.log_softmax(step_logits_fn(curr, parents), dim=-1)
        if no_repeat:
            lp = lp.scatter(1, lastw[:, None], NEG)
        if bad_ending_ids:
            prev_bad = torch.zeros_like(fin)
            for w in bad_ending_ids:
                prev_bad |= lastw == int(w)
            for e in eos_ids:
                lp[prev_bad, e] = NEG
        lp = torch.where(fin[:, None], after_end[None], lp).view(B, S, K, Vn)
        nw = torch.empty(B, S, K, dtype=torch.long)
        nb = torch.empty(B, S, K, dtype=torch.long)
        ns = torch.empty(B, S, K)
        for i in range(S):
            m = lp.masked_fill(~allowed[:, :, i, None, :], CBS_MASKED)  # (B, S, K, V): into state i
            top, cls = m.topk(K)                                        # K best words of every slot
            summed = (top + last[..., None]).reshape(B, G * K)
            sc, idx = summed.topk(K)
            nw[:, i], nb[:, i], ns[:, i] = cls.reshape(B, G * K).gather(1, idx), idx // K, sc
            if return_margins:
                margins[:, i] = torch.minimum(margins[:, i], _finite_gap(summed.topk(min(K + 1, G * K)).values, K))
        preds.append(nw.reshape(B, G))
        backs.append(nb.reshape(B, G))
        last = ns
        curr = curr.view(B, G, -1).gather(1, backs[-1][..., None].expand(B, G, curr.shape[1])).reshape(B * G, -1)
    # walk the back-pointers (:321-350; the reference asserts that this equals its re-ordered curr_ids)
    seqs = [preds[-1]]
    bp = backs[-1] if backs else None
    for t in range(len(preds) - 2, -1, -1):
        seqs.append(preds[t].gather(1, bp))
        if t > 0:
            bp = backs[t - 1].gather(1, bp)
    beams = torch.stack(list(reversed(seqs)), dim=2).view(B, S, K, -1)
    full = torch.cat([curr, preds[-1].reshape(-1, 1)], dim=1)[:, 1:].view(B, S, K, -1)
    assert bool((beams == full).all())
    if return_margins:
        return beams, last, margins
    return beams, last


def _finite_gap(sorted_vals, K):
    v = sorted_vals[..., :K + 1]
    if v.shape[-1] < 2:
        return torch.full(v.shape[:-1], float('inf'))
    gap = v[..., :-1] - v[..., 1:]
    ok = (v[..., :-1] > -1e19) & (v[..., 1:] > -1e19)
    return torch.where(ok, gap, torch.full_like(gap, float('inf'))).min(-1).values


def cbs_select_best(beams, scores, num_constraints, min_constraints_to_satisfy, eos=EOS, return_margins=False):
    """``select_best_beam_with_constraints`` (utils_cbs.py:377-443): per image, among the MAIN states s < 2**given whose bit
    count is >= min(given, min_constraints_to_satisfy), the best beam (index 0) of the state with the largest
    score / (number of non-EOS tokens + 1); torch.argmax = first maximum.  Returns (ids (B, T), logprobs (B,))."""
    B = beams.shape[0]
    eos_ids = _eos_list(eos)
    out_ids, out_lp, gaps = [], [], []
    for b in range(B):
        given = int(num_constraints[b])
        need = min(given, int(min_constraints_to_satisfy))
        valid = [s for s in range(2 ** given) if bin(s).count('1') >= need]
        vb = beams[b, valid, 0, :]
        keep = torch.ones_like(vb)
        for e in eos_ids:
            keep = keep * vb.ne(e).long()
        norm = scores[b, valid, 0] / (keep.sum(1) + 1)
        j = int(torch.argmax(norm))
        out_ids.append(vb[j])
        out_lp.append(norm[j])
        srt = torch.sort(norm, descending=True).values
        gaps.append(float(srt[0] - srt[1]) if len(valid) > 1 else float('inf'))
    if return_margins:
        return torch.stack(out_ids).long(), torch.stack(out_lp), torch.tensor(gaps)
    return torch.stack(out_ids).long(), torch.stack(out_lp)


def cbs_as_written(sd, image, fsm, num_constraints, num_beams, min_constraints_to_satisfy=2, tagemb='cls', max_length=MAX_LEN,
                   eos=EOS, no_repeat=False, bad_ending_ids=None):
    """ViTCAP.generate(use_cbs=True) (modeling_bert.py:1035-1057) with the model re-run on the full prefix at every step."""
    S = fsm.shape[1]
    step = _as_written_stepper(sd, image, S * num_beams, tagemb, max_length)
    beams, scores = cbs_search(lambda ids, parents: step(ids, parents), fsm, num_beams, max_length, eos, no_repeat=no_repeat,
                               bad_ending_ids=bad_ending_ids)
    return cbs_select_best(beams, scores, num_constraints, min_constraints_to_satisfy, eos)


def cbs_incremental(sd, image, fsm, num_constraints, num_beams, min_constraints_to_satisfy=2, emulate_bf16=False,
                    max_length=MAX_LEN, eos=EOS, return_margins=False, no_repeat=False, bad_ending_ids=None):
    """The same on the incremental formulation: the S*K slots of an image share its visual K/V, text K/V caches follow the
    back-pointers."""
    S = fsm.shape[1]
    step = _incremental_stepper(sd, image, S * num_beams, emulate_bf16, max_length)
    res = cbs_search(step, fsm, num_beams, max_length, eos, return_margins=return_margins, no_repeat=no_repeat,
                     bad_ending_ids=bad_ending_ids)
    sel = cbs_select_best(res[0], res[1], num_constraints, min_constraints_to_satisfy, eos, return_margins=return_margins)
    if return_margins:
        return sel[0], sel[1], res[2], sel[2], res[0], res[1]
    return sel


def fsm_build(constraints, vocab_size, max_given_constraints=3, max_words_per_constraint=4):
    """``FiniteStateMachineBuilder.build`` (utils_cbs.py:733-871) on TOKEN IDS: ``constraints`` = list (<= max_given) of
    constraints, each a list of words, each word a list of word-form token ids.  Main states 0 .. 2**max_given - 1 (bit n-1
    set = constraint n satisfied) loop on every word; constraint n links every main state without bit n-1 to the one with it
    through a chain of sub-states, one per word but the last; a sub-state falls back to the chain's main state on any word
    but the expected forms.  Returns (fsm (T, T, V) uint8 with T = 2**max_given * max_words, next unused sub-state)."""
    nmain = 2 ** max_given_constraints
    T = nmain * max_words_per_constraint
    fsm = torch.zeros(T, T, vocab_size, dtype=torch.uint8)
    for s in range(nmain):
        fsm[s, s, :] = 1
    sub = nmain

    def connect(frm, to, forms, reset):
        for wid in forms:
            fsm[frm, to, wid] = 1
            fsm[frm, frm, wid] = 0
        if reset is not None:
            fsm[frm, frm, :] = 0
            fsm[frm, reset, :] = 1
            for wid in forms:
                fsm[frm, reset, wid] = 0

    for n, words in enumerate(constraints, start=1):
        words = words[:max_words_per_constraint]
        stride = 2 ** (n - 1)
        frm = 0
        while frm < nmain:
            for _ in range(stride):
                cur = frm
                for i, forms in enumerate(words):
                    if i != len(words) - 1:
                        connect(cur, sub, forms, frm)
                        cur = sub
                        sub += 1
                    else:
                        connect(cur, frm + stride, forms, frm)
                frm += 1
            frm += stride
    return fsm, sub


# --------------------------------------------------------------------------------------------
# a14 / a15-T  cross-entropy training step   (modeling_bert.py:751-807, 661-690; loss.py:5-22;
#              trainer.py:95-142; optimization.py:151-210; ..._bertemb.py:280-356)
# --------------------------------------------------------------------------------------------
from vitcap_amd.synthetic import synthetic_train_inputs  # noqa: E402,F401  (host-side batch layout, shared with bench.py)


def label_smoothed_kl(logits, target, eps=0.1):
    """BertCaptioningLoss (modeling_bert.py:661-690) without drop_worst."""
    n_class = logits.size(1)
    one_hot = torch.zeros_like(logits).scatter(1, target.view(-1, 1), 1)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1)
    log_prb = F.log_softmax(logits, dim=1)
    return F.kl_div(log_prb, one_hot, reduction='none').sum(1).mean()


def focal_neg_loss(pred, target, alpha=0.5, gamma=1.0):
    """FocalLossWithLogitsNegLoss (loss.py:5-22), summed as in modeling_bert.py:789-791."""
    sp = pred.sigmoid()
    loss = (target == 1) * alpha * torch.pow(1. - sp, gamma) * F.logsigmoid(pred)
    loss = loss + (target == 0) * (1 - alpha) * torch.pow(sp, gamma) * F.logsigmoid(-pred)
    return (-loss).sum()


def dropout_keep(layer_seed, B, rows, p_drop):
    """Keep decisions (B,12,rows,rows) bool of vitcap_amd/csrc/rng.h vc_drop_keep for one decoder layer, rows indexed in
    the device's [578 visual | 20 caption] order."""
    thr = np.uint32(int(float(np.float32(p_drop)) * 4294967296.0))
    q = (np.arange(rows, dtype=np.uint32)[:, None] << np.uint32(10))
    k = np.arange(rows, dtype=np.uint32)[None, :]
    out = np.empty((B, HEADS, rows, rows), dtype=bool)
    for b in range(B):
        hb = rng_mix(np.uint32(layer_seed & 0xffffffff), np.uint32(b))
        for h in range(HEADS):
            stream = rng_mix(hb, np.uint32(h))
            out[b, h] = _lowbias32(stream ^ q ^ k) >= thr
    return out


def dropout_keep_joint(layer_seed, B, p_drop, n_text=70, max_len=MAX_LEN, n_vis=578):
    """The same decisions re-indexed to the reference's joint sequence [70 text | tag CLS | 577 visual]; the 50 tag /
    padding slots (not materialised on the device, and not able to reach the loss) are always kept."""
    S = n_text + n_vis
    dev = np.full(S, -1, dtype=np.int64)
    dev[:max_len] = n_vis + np.arange(max_len)
    dev[n_text:] = np.arange(n_vis)
    k598 = dropout_keep(layer_seed, B, n_vis + max_len, p_drop)
    ok = dev >= 0
    keep = np.ones((B, HEADS, S, S), dtype=bool)
    idx = np.where(ok)[0]
    keep[:, :, idx[:, None], idx[None, :]] = k598[:, :, dev[idx][:, None], dev[idx][None, :]]
    return torch.from_numpy(keep)


def hidden_keep(seed, B, p, rows=598, D=HID):
    """Keep decisions (B, rows, 768) bool of vitcap_hidden_dropout (csrc/train.hip) for one site: rows in the device's
    [578 visual | text] order; stream = vc_drop_stream(seed, b, 0x48), keep iff lowbias32(stream ^ (row << 10) ^ col) >= p * 2^32."""
    thr = np.uint32(int(float(np.float32(p)) * 4294967296.0))
    r = (np.arange(rows, dtype=np.uint32)[:, None] << np.uint32(10))
    d = np.arange(D, dtype=np.uint32)[None, :]
    out = np.empty((B, rows, D), dtype=bool)
    for b in range(B):
        stream = rng_mix(rng_mix(np.uint32(seed & 0xffffffff), np.uint32(b)), np.uint32(0x48))
        out[b] = _lowbias32(stream ^ r ^ d) >= thr
    return out


def hidden_keep_joint(hseed, B, p, n_text=70, max_len=MAX_LEN, n_vis=578):
    """All hidden-dropout sites of one training forward, re-indexed to the reference's joint sequence [70 text | tag CLS | 577 visual]
    (``hseed(layer, site)`` as TrainEngine derives its seeds: site 1 attention.output, 2 output, layer 4 / site 3 the embeddings).
    Returned as multipliers keep / (1 - p); the 50 tag / padding text slots have no device counterpart and cannot reach the loss: 1."""
    S = n_text + n_vis
    dev = np.full(S, -1, dtype=np.int64)
    dev[:max_len] = n_vis + np.arange(max_len)
    dev[n_text:] = np.arange(n_vis)
    idx = np.where(dev >= 0)[0]

    hs = np.float32(1.0) / (np.float32(1.0) - np.float32(p))          # the kernel's fp32 scale

    def site(l, kind):
        k = hidden_keep(hseed(l, kind), B, p, n_vis + max_len)
        full = np.ones((B, S, HID), dtype=np.float32)                 # multiplier: keep / (1 - p) where the device has the row, else 1
        full[:, idx] = k[:, dev[idx]].astype(np.float32) * hs
        return torch.from_numpy(full)
    emb = site(4, 3)[:, :n_text]
    return {'emb': emb, 'ao': [site(l, 1) for l in range(4)], 'out': [site(l, 2) for l in range(4)]}


def train_losses_as_written(sd, image, batch, tagemb='cls', layer_seeds=None, p_drop=0.0, hseed=None, p_hid=0.0):
    """ImageCaptioning.forward(train) -> ViTCAP.encode_forward(is_training=True): the full 648-row joint sequence.
    ``layer_seeds`` (4 ints) + ``p_drop`` switch the decoder's attention dropout on with the device's keep decisions;
    None = dropout off.  Returns (masked_loss, tag_loss, class_logits)."""
    img_feats = patch_embed(sd, image)
    full = construct_attn_mask(batch['attention_mask'], img_feats.shape[1])
    keep = None
    if layer_seeds is not None:
        keep = [dropout_keep_joint(sv, image.shape[0], p_drop, n_text=batch['input_ids'].shape[1]) for sv in layer_seeds]
    hk = None
    if hseed is not None and p_hid > 0:       # hidden-state dropout with the device's keep decisions (hseed: (layer, site) -> seed)
        hk = hidden_keep_joint(hseed, image.shape[0], p_hid, n_text=batch['input_ids'].shape[1])
    seq, tag_logit = joint_forward(sd, batch['input_ids'], img_feats, full, None, batch['token_type_ids'], tagemb,
                                   attn_keep=keep, p_drop=p_drop, hid_keep=hk, p_hid=p_hid)
    T = batch['masked_pos'].shape[-1]
    rows = seq[:, :T][batch['masked_pos'] == 1]
    class_logits = lm_head(sd, 'module.cls', rows)
    tgt = batch['masked_ids'][batch['masked_ids'] != 0]
    return label_smoothed_kl(class_logits.float(), tgt), focal_neg_loss(tag_logit, batch['label']), class_logits


def param_groups(names, base_lr=1e-4, weight_decay=0.05, lr_multiplier=0.1):
    """name -> (lr, weight_decay) or None when the optimizer never sees the parameter (..._bertemb.py:280-356):
    ten sub-module groups; lr x multiplier for tag_blocks, blocks[:-4], pooler, tag_logit; no decay for names
    containing 'bias' or 'LayerNorm.weight'; `module.cls.*` is not in any group (decoder.weight only via its tie)."""
    out = {}
    for n in names:
        if n.startswith('module.cls.'):
            out[n] = None
            continue
        low = False
        if n.startswith('module.bert.encoder.tag_blocks.') or n.startswith('module.bert.pooler.') \
                or n.startswith('module.bert.tag_logit.'):
            low = True
        if n.startswith('module.bert.encoder.blocks.'):
            low = int(n.split('.')[4]) < 8
        wd = weight_decay
        if 'bias' in n or 'LayerNorm.weight' in n:
            wd = 0.0
        out[n] = (base_lr * (lr_multiplier if low else 1.0), wd)
    return out


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """solver.AdamW.step for one tensor (optimization.py:187-208): decay applied AFTER the Adam update."""
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


def train_step_as_written(sd, image, batch, step=1, max_iter=10, base_lr=1e-4, clip=1.0, state=None, layer_seeds=None,
                          p_drop=0.0, hseed=None, p_hid=0.0):
    """One do_train_dict iteration (trainer.py:95-142) on a dict of leaf tensors: forward, backward, global-norm clip over
    ALL parameters, AdamW over the optimizer's groups, linear LR decay.  Returns dict(loss, tag_loss, grad_norm, grads)."""
    leaves = {}
    seen = {}
    for k, t in sd.items():
        if id(t) not in seen:
            seen[id(t)] = t.detach().clone().requires_grad_(True)
        leaves[k] = seen[id(t)]
    loss, tag_loss, _ = train_losses_as_written(leaves, image, batch, layer_seeds=layer_seeds, p_drop=p_drop, hseed=hseed, p_hid=p_hid)
    loss.backward()
    uniq = {}
    for k, t in leaves.items():
        uniq.setdefault(id(t), (k, t))
    grads = {k: t.grad for k, t in uniq.values() if t.grad is not None}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    coef = clip / (total + 1e-6)
    if coef < 1:
        for g in grads.values():
            g.mul_(coef)
    pg = param_groups(leaves.keys(), base_lr)
    lr_scale = 1.0 if state is None else state.get('lr_scale', 1.0)
    state = state if state is not None else {'m': {}, 'v': {}}
    new = {}
    with torch.no_grad():
        for k, t in uniq.values():
            if t.grad is None or pg[k] is None:
                continue
            m = state['m'].setdefault(k, torch.zeros_like(t))
            v = state['v'].setdefault(k, torch.zeros_like(t))
            lr, wd = pg[k]
            adamw_step(t, t.grad, m, v, step, lr * lr_scale, wd)
    for k, t in leaves.items():
        new[k] = t.detach()
    state['lr_scale'] = max(0.0, float(max_iter - step) / float(max(1.0, max_iter)))     # WarmupLinearSchedule, warmup 0
    return {'loss': float(loss), 'tag_loss': float(tag_loss), 'grad_norm': total, 'grads': grads, 'params': new,
            'state': state}


# --------------------------------------------------------------------------------------------
# SCST (BASELINE config 5): spec at src/pipelines/tagger_caption_uni_pipeline_expanding.py:404-478 (dead code as shipped),
# generation = modeling_utils.py:768-886 with do_sample, criterion = utils_caption_evaluate.py:162-202.
def sequence_logprob_as_written(sd, image, sample_ids, tagemb='cls', max_length=MAX_LEN, reuse_encoder=True):
    """Differentiable mean log-probability of GIVEN sampled sequences, computed the way the generator produced them: one
    full forward per generated position on [tokens so far, MASK, tag slots] (greedy_as_written's loop with the sampled
    token fed back instead of the argmax), log_softmax gathered at the sampled token, averaged over the positions at
    which the sequence was still unfinished (modeling_utils.py:850-877).  sample_ids (B,20) starting with [CLS]."""
    B = image.shape[0]
    img_feats = patch_embed(sd, image)
    input_ids0, am = test_text_inputs(B, max_length)
    full_mask = construct_attn_mask(am, img_feats.shape[1])
    od_label_ids = input_ids0[:, max_length:]
    od_len = od_label_ids.shape[1]
    pos = torch.cat([torch.arange(max_length), torch.arange(max_length, max_length + od_len)])
    full_pos = pos.unsqueeze(0).expand(B, -1)
    full_tt = torch.zeros(B, max_length + od_len, dtype=torch.long)
    enc = split_encoder(sd, img_feats) if reuse_encoder else None
    unfinished = torch.ones(B, dtype=torch.long)
    lps, unfs = [], []
    for cur_len in range(1, max_length):
        step_ids = torch.cat([sample_ids[:, :cur_len], torch.full((B, 1), MASK, dtype=torch.long)], dim=1)
        curr = step_ids.shape[1]
        mask = _remove_rows_cols(full_mask, curr, max_length, curr, max_length)
        tt = torch.cat([full_tt[:, :curr], full_tt[:, max_length:]], dim=1)
        pp = torch.cat([full_pos[:, :curr], full_pos[:, max_length:]], dim=1)
        logits = encode_forward_infer(sd, torch.cat([step_ids, od_label_ids], dim=1), img_feats, mask, pp, tt, tagemb, enc=enc)
        tok = sample_ids[:, cur_len]
        lps.append(torch.gather(F.log_softmax(logits[:, cur_len, :], dim=-1), -1, tok.unsqueeze(-1)).squeeze(-1))
        unfs.append(unfinished)
        unfinished = unfinished * tok.ne(EOS).long() * tok.ne(PAD).long() if False else unfinished * tok.ne(EOS).long()
    lp = torch.stack(lps, 1)
    uh = torch.stack(unfs, 1).float()
    return (lp * uh).sum(1) / uh.sum(1)


def scst_loss_as_written(sd, image, sample_ids, reward):
    """ScstRewardCriterion.forward: mean over samples of -(logprob * (score - baseline)); reward = score - baseline."""
    return -(sequence_logprob_as_written(sd, image, sample_ids) * reward).mean()
