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
from collections import OrderedDict

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
    def load_recipe(self, seed=0):
        sd = W.make_state_dict(seed=seed, tie_weights=self.tie_weights)
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
    def _workspace(self, B, dev, slot=0, beams=0):
        """One workspace per slot: concurrent generate() calls on different HIP streams use different slots."""
        need = lib.vitcap_engine_workspace_bytes_beam(B, beams) if beams else lib.vitcap_engine_workspace_bytes(B)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[slot] = ws
        return ws, need

    def set_sampling(self, do_sample=False, temperature=1.0, top_k=0, top_p=1.0, seed=0):
        """Token choice of generate(): greedy, or one draw from the temperature / top-k / top-p filtered distribution
        (modeling_utils.py:839-846).  Draws are a pure function of (seed, sequence, step, token)."""
        from ._lib import SampleParams
        sp = SampleParams(int(bool(do_sample)), float(temperature), int(top_k), float(top_p), int(seed) & 0xffffffff)
        if self._packed is None:
            raise RuntimeError('pack() the model before set_sampling()')
        check(lib.vitcap_engine_set_sampling(self._engine, C.byref(sp)), 'set_sampling')

    def generate(self, image, want_tags=False, slot=0):
        """image: (B,3,384,384) fp32 or bf16 on the GPU, normalised with mean=.5/std=.5."""
        if self._packed is None:
            self.pack(image.device)
        dev = self._packed[2]
        assert image.is_cuda and image.is_contiguous() and tuple(image.shape[1:]) == (3, 384, 384)
        assert image.dtype in (torch.float32, torch.bfloat16)
        B = image.shape[0]
        ws, need = self._workspace(B, dev, slot)
        ids = torch.empty((B, 1, L.MAXLEN), dtype=torch.int64, device=dev)
        lp = torch.empty((B, 1), dtype=torch.float32, device=dev)
        tag_logits = torch.empty((B, L.VOCAB), dtype=torch.float32, device=dev) if want_tags else None
        tag_topk = torch.empty((B, 50), dtype=torch.int64, device=dev) if want_tags else None
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.vitcap_gemm_set_persistent(1)
        check(lib.vitcap_engine_greedy(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16),
                                       B, C.c_void_p(ws.data_ptr()), need, C.c_void_p(ids.data_ptr()),
                                       C.c_void_p(lp.data_ptr()),
                                       C.c_void_p(tag_logits.data_ptr()) if want_tags else None,
                                       C.c_void_p(tag_topk.data_ptr()) if want_tags else None, s), 'engine_greedy')
        if want_tags:
            self.last_tags = (tag_logits, tag_topk)
        return ids, lp

    def generate_multi(self, image, seqs_per_image, slot=0, want_last=False):
        """`seqs_per_image` greedy / sampled sequences per image (num_return_sequences of ViTCAP.generate): the encoder and the
        visual prefill run once per image, the sequences of an image share its visual K/V.  Returns (ids (B*n,1,20),
        logprobs (B*n,1)) image-major -- the same sequences generate() gives on the n-times repeated batch -- and, with
        want_last, the token chosen at the last position (before the forced [SEP]) of every sequence."""
        if self._packed is None:
            self.pack(image.device)
        dev = self._packed[2]
        assert image.is_cuda and image.is_contiguous() and tuple(image.shape[1:]) == (3, 384, 384)
        B, n = image.shape[0], int(seqs_per_image)
        ws, need = self._workspace(B, dev, slot, beams=n)
        ids = torch.empty((B * n, 1, L.MAXLEN), dtype=torch.int64, device=dev)
        lp = torch.empty((B * n, 1), dtype=torch.float32, device=dev)
        last = torch.empty((B * n,), dtype=torch.int64, device=dev) if want_last else None
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.vitcap_gemm_set_persistent(1)
        wp = C.c_void_p(ws.data_ptr())
        check(lib.vitcap_engine_encode(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16), B, wp, need, s),
              'engine_encode')
        check(lib.vitcap_engine_prefill(self._engine, B, wp, need, s), 'engine_prefill')
        check(lib.vitcap_engine_decode_multi(self._engine, B, n, wp, need, C.c_void_p(ids.data_ptr()), C.c_void_p(lp.data_ptr()),
                                             C.c_void_p(last.data_ptr()) if want_last else None, s), 'engine_decode_multi')
        return (ids, lp, last) if want_last else (ids, lp)

    def generate_async(self, image, num_beams=1, length_penalty=1.0, lane=0):
        """Greedy captions through a two-slot software pipeline: the ViT encoder + decoder prefill of THIS batch run on one
        HIP stream while the 19 decode steps of the PREVIOUS batch run on another.  The decode phase is a chain of ~630
        small latency-bound kernels that leaves most of the chip idle; the encoder is MFMA-bound and has a tail at every
        GEMM -- interleaved they fill each other's gaps (B=64: 22.2 -> 19.8 ms per batch, tools/pipe_bench.py).
        Returns a handle; `.result()` waits for this batch only and gives (ids (B,1,20), logprobs (B,1)).  Results are
        identical to generate()."""
        if self._packed is None:
            self.pack(image.device)
        dev = self._packed[2]
        assert image.is_cuda and image.is_contiguous() and tuple(image.shape[1:]) == (3, 384, 384)
        assert image.dtype in (torch.float32, torch.bfloat16)
        pipes = getattr(self, '_pipes', None)
        if pipes is None:
            pipes = self._pipes = {}
        pipe = pipes.get(lane)
        if pipe is None:
            import os
            prio = int(os.environ.get('VITCAP_DECODE_PRIORITY', '-1'))     # -1 = high: the latency-bound chain goes first
            pipe = pipes[lane] = {'enc': torch.cuda.Stream(dev), 'dec': torch.cuda.Stream(dev, priority=prio),
                                  'done': [None, None], 'n': 0}
        slot = pipe['n'] % 2
        pipe['n'] += 1
        lib.vitcap_gemm_set_persistent(0)       # let the other slot's decode kernels in between GEMM tiles
        B = image.shape[0]
        ws, need = self._workspace(B, dev, slot='pipe%d_%d' % (lane, slot), beams=num_beams if num_beams > 1 else 0)
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        image.record_stream(pipe['enc'])
        with torch.cuda.stream(pipe['enc']):
            pipe['enc'].wait_event(ready)
            if pipe['done'][slot] is not None:
                pipe['enc'].wait_event(pipe['done'][slot])         # the slot's previous decode has drained
            h = C.c_void_p(pipe['enc'].cuda_stream)
            check(lib.vitcap_engine_encode(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16),
                                           B, C.c_void_p(ws.data_ptr()), need, h), 'engine_encode')
            check(lib.vitcap_engine_prefill(self._engine, B, C.c_void_p(ws.data_ptr()), need, h), 'engine_prefill')
            filled = torch.cuda.Event()
            filled.record(pipe['enc'])
        with torch.cuda.stream(pipe['dec']):
            pipe['dec'].wait_event(filled)
            ids = torch.empty((B, 1, L.MAXLEN), dtype=torch.int64, device=dev)
            lp = torch.empty((B, 1), dtype=torch.float32, device=dev)
            if num_beams > 1:
                check(lib.vitcap_engine_set_num_keep_best(self._engine, 1), 'set_num_keep_best')
                check(lib.vitcap_engine_beam_decode(self._engine, B, num_beams, length_penalty, C.c_void_p(ws.data_ptr()), need,
                                                    C.c_void_p(ids.data_ptr()), C.c_void_p(lp.data_ptr()),
                                                    C.c_void_p(pipe['dec'].cuda_stream)), 'engine_beam_decode')
            else:
                check(lib.vitcap_engine_decode(self._engine, B, C.c_void_p(ws.data_ptr()), need, C.c_void_p(ids.data_ptr()),
                                               C.c_void_p(lp.data_ptr()), C.c_void_p(pipe['dec'].cuda_stream)), 'engine_decode')
            done = torch.cuda.Event()
            done.record(pipe['dec'])
        pipe['done'][slot] = done
        return _Pending(ids, lp, done, image)

    def prime_pipeline(self, B, device, num_beams=1, lane=0):
        """Allocate (and touch) both workspaces of generate_async's two-slot pipeline for batches of B images, so that no
        allocation lands inside a caller's timed or latency-critical region.  No kernels of the captioning path run."""
        if self._packed is None:
            self.pack(device)
        for slot in range(2):
            ws, _ = self._workspace(B, self._packed[2], slot='pipe%d_%d' % (lane, slot), beams=num_beams if num_beams > 1 else 0)
            ws.zero_()

    def generate_beam(self, image, num_beams, length_penalty=1.0, slot=0, num_keep_best=1):
        """Beam search -> (ids (B,num_keep_best,20), logprobs (B,num_keep_best)), best hypothesis first, like
        ViTCAP._generate_beam_search (modeling_utils.py:888-1100)."""
        if self._packed is None:
            self.pack(image.device)
        dev = self._packed[2]
        assert image.is_cuda and image.is_contiguous() and tuple(image.shape[1:]) == (3, 384, 384)
        B = image.shape[0]
        ws, need = self._workspace(B, dev, slot, beams=num_beams)
        keep = int(num_keep_best)
        ids = torch.empty((B, keep, L.MAXLEN), dtype=torch.int64, device=dev)
        lp = torch.empty((B, keep), dtype=torch.float32, device=dev)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vitcap_engine_set_num_keep_best(self._engine, keep), 'set_num_keep_best')
        check(lib.vitcap_engine_beam(self._engine, C.c_void_p(image.data_ptr()), int(image.dtype == torch.bfloat16), B,
                                     num_beams, length_penalty, C.c_void_p(ws.data_ptr()), need,
                                     C.c_void_p(ids.data_ptr()), C.c_void_p(lp.data_ptr()), s), 'engine_beam')
        return ids, lp

    def tap(self, name, B, shape, dtype=torch.float32):
        """Copy of an engine workspace buffer after the last generate() (parity taps)."""
        ws, _ = self._workspace(B, self._packed[2])
        p = lib.vitcap_engine_tap(self._engine, name.encode(), C.c_void_p(ws.data_ptr()), B)
        if not p:
            raise KeyError(name)
        off = p - ws.data_ptr()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return ws[off:off + n].view(dtype).view(shape).clone()

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
        keep = int(te.get('num_keep_best', 1))
        if keep != 1 and te.get('num_beams', 1) == 1:
            # modeling_utils.py:790: "cannot generate >1 sentences in greedy search"
            raise AssertionError('cannot generate >1 sentences in greedy search (num_keep_best > 1 needs num_beams > 1)')
        nret = int(te.get('num_return_sequences', 1))
        if nret > 1:
            # ViTCAP.generate expands every input num_return_sequences times (modeling_bert.py:976-979, 994,
            # _expand_for_beams) and returns (B * n, 1, 20): n independent draws per image, image-major
            if not te.get('do_sample', False) or te.get('num_beams', 1) > 1:
                raise NotImplementedError('num_return_sequences > 1 needs do_sample and num_beams == 1 (the reference asserts the same)')
            if nret > 8:       # beyond the engine's sequences-per-image limit: expand the inputs like the reference does
                data = dict(data)
                data['image'] = data['image'].repeat_interleave(nret, 0).contiguous()
                nret = 1
        if te.get('max_length', 20) != L.MAXLEN:
            raise NotImplementedError('max_length is fixed to 20 in this build')
        # fail loudly on generate() options this build does not implement instead of silently ignoring them
        fixed = {'bos_token_id': 101, 'pad_token_id': 0, 'mask_token_id': 103, 'eos_token_ids': [102]}
        for k, v in fixed.items():
            got = te.get(k, v)
            if (list(got) if isinstance(got, (list, tuple)) else got) != v:
                raise NotImplementedError('%s is fixed to %r in this build (got %r)' % (k, v, got))
        if te.get('num_beams', 1) > 1 and te.get('do_sample', False):
            raise NotImplementedError('beam sampling (num_beams > 1 with do_sample, modeling_utils.py:966-985) is not built')
        if te.get('use_cbs', False):
            raise NotImplementedError('constrained beam search (use_cbs, src/tools/captioning/utils_cbs.py) is not built')
        rp = float(te.get('repetition_penalty', 1) or 1)
        if rp != getattr(self, '_rep_penalty', 1.0):
            if self._packed is None:
                self.pack(data['image'].device)
            check(lib.vitcap_engine_set_repetition_penalty(self._engine, rp), 'set_repetition_penalty')
            self._rep_penalty = rp
        if te.get('num_beams', 1) > 1:
            return self.generate_beam(data['image'], te['num_beams'], float(te.get('length_penalty', 1)), num_keep_best=keep)
        if te.get('do_sample', False):
            # every forward() call advances the stream of draws, like consecutive torch.multinomial calls would
            self._sample_calls = getattr(self, '_sample_calls', 0) + 1
            if self._packed is None:
                self.pack(data['image'].device)
            self.set_sampling(True, te.get('temperature', 1), te.get('top_k', 0), te.get('top_p', 1),
                              int(te.get('seed', 0)) + 0x632be5ab * (self._sample_calls - 1))
            try:
                return self.generate_multi(data['image'], nret) if nret > 1 else self.generate(data['image'])
            finally:
                self.set_sampling(False)
        return self.generate(data['image'])
