"""Host-side mirror of the reference's model surface for the captioning hot path.

``ImageCaptioning`` here plays the role of the reference's
``src/pipelines/tagger_caption_uni_pipeline_expanding_bertemb.py:23-189 ImageCaptioning`` wrapping
``src/layers/bert/modeling_bert.py:695 ViTCAP`` and the ``InputAsDict``-wrapped timm ViT patch embedder:

* same ``state_dict()`` key names and shapes, so a reference checkpoint loads as is;
* same call contract: ``forward(data: dict)`` -> at test time ``(ids int64 (B,1,20), logprobs fp32 (B,1))``.

All arithmetic runs in the hand-written HIP kernels of libvitcap_hip.so through one C-ABI engine call
per batch; PyTorch only owns the memory.  There is no CPU/eager fallback.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import weights as W
from ._lib import lib, check


class _Node(nn.Module):
    """Anonymous container: the parameter tree only has to reproduce the reference's key names."""


def _build_tree(root, spec, tie_weights):
    params = {}
    for key, (shape, _kind) in spec.items():
        parts = key.split('.')
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        if tie_weights and key == W.TIED_DST:
            prm = params[W.TIED_SRC]           # same Parameter object, as _tie_or_clone_weights does
        else:
            prm = nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False)
        mod.register_parameter(parts[-1], prm)
        params[key] = prm
    return params


_ROLE_STREAMS = {}


def role_stream(dev, role, make):
    """One stream per (device, role) for the life of the PROCESS.  torch hands out its pool streams round-robin and HIP maps streams onto
    a few hardware queues (GPU_MAX_HW_QUEUES, default 4): with a fresh pair of streams per model object, every new model of a process drew
    another arrangement, and the ones that put the encoder chain and the decode chain (or the loader's copies) on one queue ran the 2-slot
    pipeline at half its rate (profiles/r05_hw_queue_aliasing.txt).  Streams are only ever added, never handed back."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    s = _ROLE_STREAMS.get(key)
    if s is None:
        s = _ROLE_STREAMS[key] = make()
    return s


def _encode_stream(dev):
    """The pipeline's encoder stream.  VITCAP_ENC_CUS = n < 256 (experiment, docs/LAB_r01_r04.md 4.2 vi) confines it to the CUs of the
    low n mask bits (hipExtStreamCreateWithCUMask: bit i -> XCD i % 8), which leaves 256 - n CUs that the encoder's GEMM
    workgroups never occupy for the decode chain's dependent launches to start on."""
    import os
    n = int(os.environ.get('VITCAP_ENC_CUS', '256'))
    if n >= 256:
        return torch.cuda.Stream(dev)
    words = (C.c_uint32 * 8)()
    for cu in range(n):
        words[cu // 32] |= 1 << (cu % 32)
    hip = C.CDLL('libamdhip64.so')
    h = C.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


class _Pending(object):
    """Handle of one batch in flight in ImageCaptioning.generate_async."""

    def __init__(self, ids, lp, event, keep):
        self.ids, self.lp, self.event, self._keep = ids, lp, event, keep

    def wait(self, stream=None):
        """Makes `stream` (default: the current stream) wait for this batch, without blocking the host."""
        (stream or torch.cuda.current_stream(self.ids.device)).wait_event(self.event)
        return self.ids, self.lp

    def result(self):
        self.event.synchronize()
        self._keep = None
        return self.ids, self.lp


class ImageCaptioning(nn.Module):
    """ViT-B/16-384 + tag head + 4-layer BERT caption decoder, greedy decode on MI355X."""

    def __init__(self, tie_weights=True, tagemb='cls', test_extra_input=None, cfg=None):
        super().__init__()
        self.tie_weights = tie_weights
        self.tagemb = tagemb
        self.cfg = cfg
        # decode kwargs the reference passes at test time (..._bertemb.py:588-608)
        self.test_extra_input = dict(is_decode=True, do_sample=False, bos_token_id=101, pad_token_id=0,
                                     eos_token_ids=[102], mask_token_id=103, add_od_labels=True,
                                     od_labels_start_posid=20, max_length=20, num_beams=1, temperature=1,
                                     top_k=0, top_p=1, repetition_penalty=1, length_penalty=1,
                                     num_return_sequences=1, num_keep_best=1)
        if test_extra_input:
            self.test_extra_input.update(test_extra_input)
        self._params = _build_tree(self, W.state_dict_spec(), tie_weights)
        self._engine = None
        self._packed = None
        self._ws = {}
        self.last_tags = None

    # ---------------------------------------------------------------- weights
    def load_recipe(self, seed=0, vbias_std=None, bf16_exact=True):
        sd = W.make_state_dict(seed=seed, tie_weights=self.tie_weights, vbias_std=vbias_std, bf16_exact=bf16_exact)
        with torch.no_grad():
            for k, v in sd.items():
                self._params[k].copy_(torch.from_numpy(v))
        self._packed = None
        return self

    def load_state_dict(self, state_dict, strict=True):
        res = super().load_state_dict(state_dict, strict=strict)
        self._packed = None
        return res

    def _t(self, key):
        return self._params[key].detach()

    def pack(self, device='cuda'):
        """Re-lay the checkpoint tensors for the kernels (bf16 matrices, fused decoder QKV, padded vocab)."""
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        keep = []

        def bf(t):
            o = t.to(device=dev, dtype=torch.float32).to(torch.bfloat16).contiguous()
            keep.append(o)
            return o

        def f32(t):
            o = t.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(o)
            return o

        def ptr(t):
            return C.c_void_p(t.data_ptr())

        def pad_vocab(t):
            o = torch.zeros((L.VOCAB_PAD,) + tuple(t.shape[1:]), dtype=torch.float32)
            o[:t.shape[0]] = t
            return o

        w = L.Weights()
        ie = 'image_encoder.module.'
        w.patch_w = ptr(bf(self._t(ie + 'patch_embed.proj.weight').reshape(768, 768)))
        w.patch_b = ptr(f32(self._t(ie + 'patch_embed.proj.bias')))
        w.cls_token = ptr(f32(self._t(ie + 'cls_token').reshape(768)))
        w.pos_embed = ptr(f32(self._t(ie + 'pos_embed').reshape(577, 768)))

        def vit_block(dst, p):
            dst.qkv_w = ptr(bf(self._t(p + '.attn.qkv.weight')))
            dst.qkv_b = ptr(f32(self._t(p + '.attn.qkv.bias')))
            dst.proj_w = ptr(bf(self._t(p + '.attn.proj.weight')))
            dst.proj_b = ptr(f32(self._t(p + '.attn.proj.bias')))
            dst.fc1_w = ptr(bf(self._t(p + '.mlp.fc1.weight')))
            dst.fc1_b = ptr(f32(self._t(p + '.mlp.fc1.bias')))
            dst.fc2_w = ptr(bf(self._t(p + '.mlp.fc2.weight')))
            dst.fc2_b = ptr(f32(self._t(p + '.mlp.fc2.bias')))
            dst.n1_g = ptr(f32(self._t(p + '.norm1.weight')))
            dst.n1_b = ptr(f32(self._t(p + '.norm1.bias')))
            dst.n2_g = ptr(f32(self._t(p + '.norm2.weight')))
            dst.n2_b = ptr(f32(self._t(p + '.norm2.bias')))

        for i in range(12):
            vit_block(w.blocks[i], 'module.bert.encoder.blocks.%d' % i)
        for i in range(4):
            vit_block(w.tag_blocks[i], 'module.bert.encoder.tag_blocks.%d' % i)
        w.pooler_w = ptr(bf(self._t('module.bert.pooler.dense.weight')))
        w.pooler_b = ptr(f32(self._t('module.bert.pooler.dense.bias')))

        def lm_head(dst, p, dec_w_packed=None):
            dst.dense_w = ptr(bf(self._t(p + '.predictions.transform.dense.weight')))
            dst.dense_b = ptr(f32(self._t(p + '.predictions.transform.dense.bias')))
            dst.ln_g = ptr(f32(self._t(p + '.predictions.transform.LayerNorm.weight')))
            dst.ln_b = ptr(f32(self._t(p + '.predictions.transform.LayerNorm.bias')))
            dw = dec_w_packed if dec_w_packed is not None else bf(pad_vocab(self._t(p + '.predictions.decoder.weight')))
            dst.dec_w = ptr(dw)
            dst.dec_b = ptr(f32(pad_vocab(self._t(p + '.predictions.bias'))))
            return dw

        lm_head(w.tag_logit, 'module.bert.tag_logit')
        e = 'module.bert.embeddings'
        word = bf(pad_vocab(self._t(e + '.word_embeddings.weight')))
        w.word_emb = ptr(word)
        w.pos_emb = ptr(bf(self._t(e + '.position_embeddings.weight')))
        w.type_emb = ptr(bf(self._t(e + '.token_type_embeddings.weight')))
        w.emb_ln_g = ptr(f32(self._t(e + '.LayerNorm.weight')))
        w.emb_ln_b = ptr(f32(self._t(e + '.LayerNorm.bias')))
        for i in range(4):
            p = 'module.bert.decoder.layer.%d' % i
            d = w.dec[i]
            d.qkv_w = ptr(bf(torch.cat([self._t('%s.attention.self.%s.weight' % (p, n))
                                        for n in ('query', 'key', 'value')], 0)))
            d.qkv_b = ptr(f32(torch.cat([self._t('%s.attention.self.%s.bias' % (p, n))
                                         for n in ('query', 'key', 'value')], 0)))
            d.ao_w = ptr(bf(self._t(p + '.attention.output.dense.weight')))
            d.ao_b = ptr(f32(self._t(p + '.attention.output.dense.bias')))
            d.ao_g = ptr(f32(self._t(p + '.attention.output.LayerNorm.weight')))
            d.ao_beta = ptr(f32(self._t(p + '.attention.output.LayerNorm.bias')))
            d.i_w = ptr(bf(self._t(p + '.intermediate.dense.weight')))
            d.i_b = ptr(f32(self._t(p + '.intermediate.dense.bias')))
            d.o_w = ptr(bf(self._t(p + '.output.dense.weight')))
            d.o_b = ptr(f32(self._t(p + '.output.dense.bias')))
            d.o_g = ptr(f32(self._t(p + '.output.LayerNorm.weight')))
            d.o_beta = ptr(f32(self._t(p + '.output.LayerNorm.bias')))
        tied = self._params[W.TIED_DST] is self._params[W.TIED_SRC]
        lm_head(w.cls, 'module.cls', dec_w_packed=word if tied else None)
        if self.tagemb != 'cls':
            # bert.extra_embeddings: the tag rows' embedding under branch B of modeling_bert.py:1484-1485 (only read when the tag
            # tokens are visible to the caption, vitcap_gen_opts.tag_visible > 0)
            x = 'module.bert.extra_embeddings'
            w.xword_emb = ptr(bf(pad_vocab(self._t(x + '.word_embeddings.weight'))))
            w.xpos_emb = ptr(bf(self._t(x + '.position_embeddings.weight')))
            w.xtype_emb = ptr(bf(self._t(x + '.token_type_embeddings.weight')))
            w.xemb_ln_g = ptr(f32(self._t(x + '.LayerNorm.weight')))
            w.xemb_ln_b = ptr(f32(self._t(x + '.LayerNorm.bias')))

        if self._engine is None:
            h = C.c_void_p()
            check(lib.vitcap_engine_create(C.byref(h)), 'engine_create')
            self._engine = h
        check(lib.vitcap_engine_bind_weights(self._engine, C.byref(w)), 'bind_weights')
        self._packed = (w, keep, dev)
        return self

    def __del__(self):
        try:
            if self._engine is not None:
                lib.vitcap_engine_destroy(self._engine)
        except Exception:
            pass

    # ---------------------------------------------------------------- forward
    def _workspace(self, B, dev, slot=0, opts=None, wait=None):
        """One workspace per slot: concurrent generate() calls on different HIP streams use different slots.  `wait`: event of
        the slot's in-flight work, waited for before a larger buffer replaces the old one (the caching allocator may hand the
        old block to another stream at once)."""
        need = lib.vitcap_engine_workspace_bytes(B, C.byref(opts) if opts is not None else None)
        if need == 0:
            check(lib.vitcap_gen_opts_check(C.byref(opts)), 'gen_opts')
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != dev:
            if ws is not None and wait is not None:
                wait.synchronize()
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[slot] = ws
        return ws, need

    # ---- options: test_extra_input (the kwargs ViTCAP.generate receives, ..._bertemb.py:588-608) -> vitcap_gen_opts
    def gen_options(self, gemm_mode=L.GEMM_AUTO, use_graph=None, **over):
        """Validated vitcap_gen_opts for the current test_extra_input (+ overrides).  Raises for generate() options this build
        does not implement instead of silently ignoring them; used by forward() AND by the pipeline's predict loop."""
        te = dict(self.test_extra_input)
        te.update(over)
        cbs = None
        if te.get('use_cbs', False):
            # ViTCAP.generate(use_cbs=True, fsm=, num_constraints=, min_constraints_to_satisfy=) (modeling_bert.py:932-933, 949-953,
            # 1035-1057).  The reference's pipeline sets use_cbs / min_constraints_to_satisfy (..._bertemb.py:175-179) but nothing in
            # it ever supplies `fsm`: there generate() dies on `fsm.shape`; here the missing tensors are named.
            fsm, ncons = te.get('fsm'), te.get('num_constraints')
            if fsm is None or ncons is None:
                raise ValueError('use_cbs needs `fsm` (B, S, S, %d) uint8 and `num_constraints` (B,) next to it (vitcap_amd.cbs.batch_fsm); '
                                 'the reference reads fsm.shape at modeling_bert.py:952' % L.VOCAB)
            if fsm.dim() != 4 or fsm.shape[1] != fsm.shape[2] or fsm.shape[3] != L.VOCAB:
                raise AssertionError('fsm must be (B, S, S, %d), got %s (modeling_bert.py:953)' % (L.VOCAB, tuple(fsm.shape)))
            if not (fsm.is_cuda and ncons.is_cuda and fsm.is_contiguous() and fsm.dtype in (torch.uint8, torch.bool)):
                raise ValueError('fsm / num_constraints must be contiguous device tensors (uint8 / int64)')
            ncons = ncons.to(torch.int64).contiguous()
            cbs = (fsm, ncons)
            self._cbs_keep = cbs                       # the option struct holds raw pointers into these
        if not te.get('add_od_labels', True):
            # without the 50 od slots ViTSplitCLSEmbModel.forward has no rows to write its tag embeddings over
            # (`embedding_output[:, -pred_topk.shape[1]:] = tag_embedding`, modeling_bert.py:1467 / 1489) and the reference fails
            raise NotImplementedError('add_od_labels=False: the reference model needs the 50 tag slots behind the caption')
        eos = te.get('eos_token_ids', [102])
        eos = [int(x) for x in eos] if isinstance(eos, (list, tuple)) else [int(eos)]
        if not 1 <= len(eos) <= 4:
            raise NotImplementedError('eos_token_ids must hold 1..4 ids in this build (got %r)' % (eos,))
        nb, keep = int(te.get('num_beams', 1) or 1), int(te.get('num_keep_best', 1) or 1)
        nret = int(te.get('num_return_sequences', 1) or 1)
        do_sample = bool(te.get('do_sample', False))
        if keep != 1 and nb == 1:
            # modeling_utils.py:790: "cannot generate >1 sentences in greedy search"
            raise AssertionError('cannot generate >1 sentences in greedy search (num_keep_best > 1 needs num_beams > 1)')
        if len(eos) > 1 and nb > 1:
            # with several EOS ids more than num_beams of the 2*num_beams candidates can be EOS words and the reference's own
            # `assert len(next_sent_beam) == num_beams` fires (modeling_utils.py:1037): only the greedy / sampling loop takes a list
            raise NotImplementedError('several eos_token_ids need num_beams == 1 (the reference\'s beam search asserts with them)')
        if nret > 1 and (not do_sample or nb > 1):
            raise NotImplementedError('num_return_sequences > 1 needs do_sample and num_beams == 1 (the reference asserts the same)')
        ml = int(te.get('max_length', L.MAXLEN))
        if not 2 <= ml <= L.MAXLEN_CAP:
            raise NotImplementedError('max_length must be 2..%d in this build (got %d)' % (L.MAXLEN_CAP, ml))
        sp = L.SampleParams(int(do_sample), float(te.get('temperature', 1) or 1), int(te.get('top_k', 0) or 0),
                            float(te.get('top_p', 1) or 1), int(te.get('seed', 0)) & 0xffffffff)
        if use_graph is None:
            use_graph = bool(te.get('use_graph', False))
        o = L.gen_opts(num_beams=nb, seqs_per_image=min(nret, 8) if nret > 1 else 1, num_keep_best=keep, max_length=ml,
                       bos_token_id=int(te.get('bos_token_id', 101)), eos_token_id=int(eos[0]),
                       pad_token_id=int(te.get('pad_token_id', 0)), mask_token_id=int(te.get('mask_token_id', 103)),
                       length_penalty=float(te.get('length_penalty', 1) or 1),
                       repetition_penalty=float(te.get('repetition_penalty', 1) or 1), sampling=sp, gemm_mode=int(gemm_mode),
                       early_exit=int(bool(te.get('early_exit', True))), use_graph=int(bool(use_graph)),
                       tag_visible=int(te.get('tag_visible', 0) or 0), tagemb_cls=int(self.tagemb == 'cls'),
                       decode_streams=int(te.get('decode_streams', 0) or 0), encode_parts=int(te.get('encode_parts', 0) or 0),
                       eos_extra=eos[1:], tag_pos0=max(int(te.get('od_labels_start_posid', 20) or 0), ml, 20))
        if cbs is not None:
            o.use_cbs, o.cbs_states = 1, int(cbs[0].shape[1])
            o.min_constraints_to_satisfy = int(te.get('min_constraints_to_satisfy', 2))
            o.fsm, o.num_constraints = cbs[0].data_ptr(), cbs[1].data_ptr()
            # the search's optional rules (generate kwargs, modeling_bert.py:933, 1039-1042)
            if te.get('use_hypo', False):
                # upstream cannot run it either: utils_cbs.py:273 indexes a tensor with `beam_id / per_node_beam_size`, a float
                raise NotImplementedError('use_hypo: the reference\'s own hypothesis branch fails on a float tensor index (utils_cbs.py:273)')
            o.cbs_no_repeat = int(bool(te.get('decoding_constraint_flag', False)))
            bad = [int(x) for x in (te.get('bad_ending_ids') or [])]
            if len(bad) > 16:
                raise NotImplementedError('bad_ending_ids: at most 16 ids in this build (got %d)' % len(bad))
            o.cbs_bad_ending = (C.c_int32 * 16)(*(bad + [-1] * (16 - len(bad))))
        check(lib.vitcap_gen_opts_check(C.byref(o)), 'gen_opts')
        return o

    @staticmethod
    def _text_mask_pattern(T, max_length, n_tag, dtype, device):
        """The attention_mask the kernels implement for n_tag visible tag slots (dataset.py:377-390): tril on the caption slots,
        ones on [0:L, L:L+n] and [L:L+n, L:L+n], zeros elsewhere."""
        want = torch.zeros((T, T), dtype=dtype, device=device)
        want[:max_length, :max_length] = torch.tril(torch.ones(max_length, max_length, dtype=dtype, device=device))
        if n_tag:
            want[max_length:max_length + n_tag, max_length:max_length + n_tag] = 1
            want[:max_length, max_length:max_length + n_tag] = 1
        return want

    def check_text_inputs(self, data, max_length=L.MAXLEN, expect_n_tag=None):
        """The text side of the test-time batch (CaptionTensorizer.tensorize_ab at test time, dataset.py:218-219, 326, 377-390;
        consumed by ImageCaptioning.construct_attn_mask ..._bertemb.py:57-85 and ViTCAP.generate modeling_bert.py:955-1001).
        The engine hard-wires the mask STRUCTURE those tensors describe -- caption row i attends caption rows <= i and the 578
        visual tokens, nothing attends the tag slots -- so what the caller hands over is verified against that structure and
        anything else is refused: a different mask can no longer yield a silently wrong caption.

        * attention_mask (B, T, T), T = max_length + od_len: lower-triangular ones on [0:max_length, 0:max_length], zeros
          elsewhere (the 210 ones of SURVEY 8d for max_length 20) -- or, additionally, the first n tag slots visible to every
          caption row and to each other (what tensorize_ab builds for a text_b of n tokens): returns n;
        * token_type_ids: zeros;  masked_pos: not read by generate() beyond slicing;
        * input_ids (B, T): only the od-label slots [max_length:] are read by generate() (modeling_bert.py:959); their
          embeddings are overwritten by the predicted tag tokens (1435-1489) and never attended, so any value is harmless.

        expect_n_tag=None: returns n, read off the mask (host tensors: no GPU involved; device tensors: one host synchronisation).
        expect_n_tag=n (the pipeline's steady state, pipeline.predict): tensors that live on the device are compared with the
        pattern for n ON THEIR STREAM and nothing is read back -- returns (n, flag) with flag a 0-d device bool (True = inputs
        acceptable) that the caller tests when it next synchronises anyway, or None when everything was checked on the host."""
        am = data.get('attention_mask')
        defer = expect_n_tag is not None
        n_tag = int(expect_n_tag) if defer else 0
        flag = None

        def _verdict(ok_tensor, make_error):
            nonlocal flag
            if defer and ok_tensor.is_cuda:
                flag = ok_tensor if flag is None else (flag & ok_tensor)
            elif not bool(ok_tensor):
                raise make_error()

        if am is not None:
            if am.dim() != 3 or am.shape[1] != am.shape[2] or am.shape[1] < max_length:
                raise ValueError('attention_mask must be (B, T, T) with T >= max_length=%d, got %s' % (max_length, tuple(am.shape)))
            T = am.shape[1]
            # tags visible to the caption (a text_b of n tokens, dataset.py:240-252, 387-390): ones on [L0:L0+n, L0:L0+n] and on
            # [0:L0, L0:L0+n]; n is read off the first caption row of the first sample and must describe the whole batch
            if not defer or not am.is_cuda:
                n_tag = int((am[0, 0, max_length:] != 0).sum())
                if defer and n_tag != int(expect_n_tag):
                    raise ValueError('attention_mask shows %d visible tag slots, the batches before it showed %d' % (n_tag, int(expect_n_tag)))
            want = self._text_mask_pattern(T, max_length, n_tag, am.dtype, am.device)
            _verdict((am == want.unsqueeze(0)).all(), lambda: NotImplementedError(
                'attention_mask is neither the test-time seq2seq pattern (tril on the first %d caption slots, zeros elsewhere: '
                'dataset.py:377-390) nor that pattern with the first n tag slots visible to the caption and to each other '
                '(dataset.py:387-390); the HIP engine implements these two mask structures only' % max_length))
            if n_tag > 50 or (n_tag and max_length != L.MAXLEN):
                raise NotImplementedError('tag tokens visible to the caption need max_length == 20 and at most 50 tag slots')
        tt = data.get('token_type_ids')
        if tt is not None:
            _verdict((tt == 0).all(), lambda: NotImplementedError(
                'token_type_ids must be all zero at test time (dataset.py:326); segment-1 text tokens are not built'))
        ii = data.get('input_ids')
        if ii is not None and (ii.dim() != 2 or ii.shape[1] < max_length):
            raise ValueError('input_ids must be (B, T) with T >= max_length=%d, got %s' % (max_length, tuple(ii.shape)))
        for k in ('input_ids', 'attention_mask', 'token_type_ids', 'masked_pos'):
            v = data.get(k)
            if v is not None and v.shape[0] != data['image'].shape[0]:
                raise ValueError('%s has batch %d but image has %d' % (k, v.shape[0], data['image'].shape[0]))
        return (n_tag, flag) if defer else n_tag

    def _check_image(self, image):
        if self._packed is None:
            self.pack(image.device)
        assert image.is_cuda and image.is_contiguous() and tuple(image.shape[1:]) == (3, 384, 384)
        assert image.dtype in (torch.float32, torch.bfloat16)
        return self._packed[2]

    def _out_buffers(self, B, o, dev):
        if o.use_cbs:
            shape = (B, 1)
        elif o.num_beams > 1:
            shape = (B, o.num_keep_best)
        else:
            shape = (B * o.seqs_per_image, 1)
        return (torch.empty(shape + (o.max_length,), dtype=torch.int64, device=dev),
                torch.empty(shape, dtype=torch.float32, device=dev))

    def run(self, image, opts, want_tags=False, slot=0, want_last=False):
        """One vitcap_engine_generate call on the current stream under `opts` (a vitcap_gen_opts from gen_options())."""
        dev = self._check_image(image)
        B = image.shape[0]
        ws, need = self._workspace(B, dev, slot, opts)
        ids, lp = self._out_buffers(B, opts, dev)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wp = C.c_void_p(ws.data_ptr())
        if want_last:
            last = torch.empty((B * opts.seqs_per_image,), dtype=torch.int64, device=dev)
            check(lib.vitcap_engine_encode(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16), B,
                                           C.byref(opts), wp, need, s), 'engine_encode')
            check(lib.vitcap_engine_prefill(self._engine, B, C.byref(opts), wp, need, s), 'engine_prefill')
            check(lib.vitcap_engine_decode(self._engine, B, C.byref(opts), wp, need, C.c_void_p(ids.data_ptr()),
                                           C.c_void_p(lp.data_ptr()), C.c_void_p(last.data_ptr()), s), 'engine_decode')
            return ids, lp, last
        tag_logits = torch.empty((B, L.VOCAB), dtype=torch.float32, device=dev) if want_tags else None
        tag_topk = torch.empty((B, 50), dtype=torch.int64, device=dev) if want_tags else None
        check(lib.vitcap_engine_generate(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16), B,
                                         C.byref(opts), wp, need, C.c_void_p(ids.data_ptr()), C.c_void_p(lp.data_ptr()),
                                         C.c_void_p(tag_logits.data_ptr()) if want_tags else None,
                                         C.c_void_p(tag_topk.data_ptr()) if want_tags else None, s), 'engine_generate')
        if want_tags:
            self.last_tags = (tag_logits, tag_topk)
        self._last = (slot, opts, B)
        return ids, lp

    def generate(self, image, want_tags=False, slot=0, **over):
        """Greedy captions.  image: (B,3,384,384) fp32 or bf16 on the GPU, normalised with mean=.5/std=.5."""
        return self.run(image, self.gen_options(num_beams=1, do_sample=False, num_return_sequences=1, num_keep_best=1, **over),
                        want_tags=want_tags, slot=slot)

    def generate_multi(self, image, seqs_per_image, slot=0, want_last=False, **over):
        """`seqs_per_image` sampled sequences per image (num_return_sequences of ViTCAP.generate): the encoder and the
        visual prefill run once per image, the sequences of an image share its visual K/V.  Returns (ids (B*n,1,L),
        logprobs (B*n,1)) image-major -- the same sequences generate() gives on the n-times repeated batch -- and, with
        want_last, the token chosen at the last position (before the forced [SEP]) of every sequence."""
        o = self.gen_options(num_beams=1, num_keep_best=1, num_return_sequences=int(seqs_per_image), do_sample=True, **over)
        return self.run(image, o, slot=slot, want_last=want_last)

    def generate_beam(self, image, num_beams, length_penalty=1.0, slot=0, num_keep_best=1, **over):
        """Beam search -> (ids (B,num_keep_best,L), logprobs (B,num_keep_best)), best hypothesis first, like
        ViTCAP._generate_beam_search (modeling_utils.py:888-1100)."""
        over.setdefault('do_sample', False)      # do_sample=True: beam sampling (modeling_utils.py:966-985), temperature / top_k / top_p / seed
        o = self.gen_options(num_beams=int(num_beams), length_penalty=float(length_penalty), num_keep_best=int(num_keep_best),
                             num_return_sequences=1, **over)
        return self.run(image, o, slot=slot)

    def generate_cbs(self, image, fsm, num_constraints, num_beams=1, min_constraints_to_satisfy=2, slot=0, **over):
        """Constrained beam search: ViTCAP.generate(use_cbs=True, fsm=fsm, num_constraints=..., min_constraints_to_satisfy=...)
        (modeling_bert.py:1035-1057 -> utils_cbs.py:26-443) -> (ids (B, 1, T), logprobs (B, 1)).  Like the reference's output the
        ids carry NO BOS column and T <= max_length - 1 is the number of words decoded before every one of the batch's
        B * S * num_beams sequences had ended (one host synchronisation reads it).  fsm / num_constraints: vitcap_amd.cbs.batch_fsm."""
        if fsm is None or num_constraints is None:
            self.gen_options(use_cbs=True, fsm=fsm, num_constraints=num_constraints)       # raises, naming what is missing
        if fsm.shape[0] != image.shape[0] or num_constraints.shape[0] != image.shape[0]:
            raise AssertionError('fsm / num_constraints are for %d images, the batch has %d (modeling_bert.py:953)'
                                 % (fsm.shape[0], image.shape[0]))
        o = self.gen_options(use_cbs=True, fsm=fsm, num_constraints=num_constraints, num_beams=int(num_beams), num_keep_best=1,
                             min_constraints_to_satisfy=int(min_constraints_to_satisfy), do_sample=False, num_return_sequences=1, **over)
        ids, lp = self.run(image, o, slot=slot)
        n_pred = int(self.tap('cbs_npred', image.shape[0], (1,), dtype=torch.int32, slot=slot, opts=o)[0])
        return ids[:, :, :n_pred].contiguous(), lp

    def generate_async(self, image, num_beams=1, length_penalty=1.0, lane=0, opts=None):
        """Captions through a two-slot software pipeline: the ViT encoder + decoder prefill of THIS batch run on one
        HIP stream while the decode steps of the PREVIOUS batch run on another.  The decode phase is a chain of small
        latency-bound kernels that leaves most of the chip idle; the encoder is MFMA-bound and has a tail at every
        GEMM -- interleaved they fill each other's gaps (tools/pipe_bench.py).  The large GEMMs run one tile per workgroup
        here (vitcap_gen_opts.gemm_mode = TILES, a per-call option).
        Returns a handle; `.result()` waits for this batch only and gives (ids, logprobs).  Results are identical to run()."""
        dev = self._check_image(image)
        if opts is None:
            opts = self.gen_options(gemm_mode=L.GEMM_TILES, num_beams=int(num_beams), length_penalty=float(length_penalty),
                                    num_keep_best=1, do_sample=False, num_return_sequences=1)
        pipes = self.__dict__.setdefault('_pipes', {})
        pipe = pipes.get(lane)
        if pipe is None:
            import os
            prio = int(os.environ.get('VITCAP_DECODE_PRIORITY', '-1'))     # -1 = high: the latency-bound chain goes first
            pipe = pipes[lane] = {'enc': role_stream(dev, 'enc%d' % lane, lambda: _encode_stream(dev)),
                                  'dec': role_stream(dev, 'dec%d' % lane, lambda: torch.cuda.Stream(dev, priority=prio)),
                                  'done': [None, None], 'n': 0}
        slot = pipe['n'] % 2
        pipe['n'] += 1
        B = image.shape[0]
        ws, need = self._workspace(B, dev, slot='pipe%d_%d' % (lane, slot), opts=opts, wait=pipe['done'][slot])
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        image.record_stream(pipe['enc'])
        wp = C.c_void_p(ws.data_ptr())
        with torch.cuda.stream(pipe['enc']):
            pipe['enc'].wait_event(ready)
            if pipe['done'][slot] is not None:
                pipe['enc'].wait_event(pipe['done'][slot])         # the slot's previous decode has drained
            h = C.c_void_p(pipe['enc'].cuda_stream)
            check(lib.vitcap_engine_encode(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16),
                                           B, C.byref(opts), wp, need, h), 'engine_encode')
            check(lib.vitcap_engine_prefill(self._engine, B, C.byref(opts), wp, need, h), 'engine_prefill')
            filled = torch.cuda.Event()
            filled.record(pipe['enc'])
        with torch.cuda.stream(pipe['dec']):
            pipe['dec'].wait_event(filled)
            ids, lp = self._out_buffers(B, opts, dev)
            check(lib.vitcap_engine_decode(self._engine, B, C.byref(opts), wp, need, C.c_void_p(ids.data_ptr()),
                                           C.c_void_p(lp.data_ptr()), None, C.c_void_p(pipe['dec'].cuda_stream)), 'engine_decode')
            # blocking: a host thread that waits for this batch SLEEPS (hipEventBlockingSync) instead of spinning on a core -- under a
            # CPU quota (the GPU pool gives 16 cores) spinning waiters starve the JPEG decode workers of the loader
            done = torch.cuda.Event(blocking=True)
            done.record(pipe['dec'])
        pipe['done'][slot] = done
        return _Pending(ids, lp, done, image)

    def prime_pipeline(self, B, device, num_beams=1, lane=0, opts=None):
        """Allocate (and touch) both workspaces of generate_async's two-slot pipeline for batches of B images, so that no
        allocation lands inside a caller's timed or latency-critical region.  No kernels of the captioning path run."""
        if self._packed is None:
            self.pack(device)
        if opts is None:
            opts = self.gen_options(gemm_mode=L.GEMM_TILES, num_beams=int(num_beams), num_keep_best=1, do_sample=False,
                                    num_return_sequences=1)
        for slot in range(2):
            ws, _ = self._workspace(B, self._packed[2], slot='pipe%d_%d' % (lane, slot), opts=opts)
            ws.zero_()

    def tap(self, name, B, shape, dtype=torch.float32, slot=0, opts=None):
        """Copy of an engine workspace buffer after the last run() on `slot` (parity taps)."""
        if opts is None:
            last = getattr(self, '_last', None)
            opts = last[1] if last is not None and last[0] == slot and last[2] == B else self.gen_options()
        ws, _ = self._workspace(B, self._packed[2], slot, opts)
        p = lib.vitcap_engine_tap(self._engine, name.encode(), C.c_void_p(ws.data_ptr()), B, C.byref(opts))
        if not p:
            raise KeyError(name)
        off = p - ws.data_ptr()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return ws[off:off + n].view(dtype).view(shape).clone()

    def graph_count(self):
        return int(lib.vitcap_engine_graph_count(self._engine)) if self._engine is not None else 0

    def forward(self, data):
        """Test-time contract of the reference wrapper (..._bertemb.py:87-184)."""
        data = dict(data.items())
        data.pop('key', None)
        if self.training:
            # a16 train branch (..._bertemb.py:93-171): returns {'masked_loss': loss}.  Forward and backward are one fused
            # pass of the HIP training engine; loss.backward() (what do_train_dict calls, trainer.py:119) finalises the
            # gradients already sitting in the engine's flat buffer, engine.optimizer_step() is clip + AdamW + schedule.
            eng = getattr(self, 'train_engine', None)
            if eng is None:
                raise RuntimeError('training-mode forward needs a vitcap_amd.train.TrainEngine(model, ...) attached')
            return eng.loss_dict(data)
        te = self.test_extra_input
        over = {}
        if te.get('do_sample', False):
            # every forward() call advances the stream of draws, like consecutive torch.multinomial calls would
            self._sample_calls = getattr(self, '_sample_calls', 0) + 1
            over['seed'] = int(te.get('seed', 0)) + 0x632be5ab * (self._sample_calls - 1)
        nret = int(te.get('num_return_sequences', 1) or 1)
        image = data['image']
        if nret > 8 and te.get('do_sample', False) and int(te.get('num_beams', 1) or 1) == 1:
            # beyond the engine's sequences-per-image limit: expand the inputs like the reference does (modeling_bert.py:976-994)
            image = image.repeat_interleave(nret, 0).contiguous()
            over['num_return_sequences'] = 1
        if te.get('use_cbs', False):
            # the batch carries the machines (modeling_bert.py:932: generate's own kwargs)
            n_tag = self.check_text_inputs(data, int(te.get('max_length', L.MAXLEN)))
            if n_tag:
                raise NotImplementedError('use_cbs with tag tokens visible to the caption is not built')
            return self.generate_cbs(image, data.get('fsm', te.get('fsm')), data.get('num_constraints', te.get('num_constraints')),
                                     num_beams=int(te.get('num_beams', 1) or 1),
                                     min_constraints_to_satisfy=int(data.get('min_constraints_to_satisfy',
                                                                             te.get('min_constraints_to_satisfy', 2))))
        opts = self.gen_options(**over)
        n_tag = self.check_text_inputs(data, opts.max_length)
        if n_tag != opts.tag_visible:
            if opts.tag_visible:
                raise ValueError('test_extra_input tag_visible=%d but the attention_mask shows %d tag slots' % (opts.tag_visible, n_tag))
            opts = self.gen_options(tag_visible=n_tag, **over)      # the caller's mask decides
        return self.run(image, opts)
