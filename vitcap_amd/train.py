"""Cross-entropy training step of ViTCAP on MI355X.

Mirrors ``do_train_dict`` (src/tools/opt/trainer.py:95-142) around ``ImageCaptioning.forward`` in training mode
(..._bertemb.py:87-171 -> ViTCAP.encode_forward(is_training=True), modeling_bert.py:751-807): forward, masked-token
label-smoothed loss, backward, global-norm clip over ALL parameters, the reference's AdamW (decay after the update)
on its ten parameter groups, linear LR decay.  Every FLOP runs in the hand-written HIP kernels; this module only
sequences launches and owns the buffers (torch is used for allocation, views, row gathers/copies and the RCCL
all-reduce).

What is computed (identical loss and gradients to the reference, less dead work):
* the decoder runs on [578 visual | 20 caption] rows per image.  The reference's 50 tag slots and the padding rows
  are attended by nothing that reaches the loss (seq2seq mask with text_b='', dataset.py:377-390), so their rows are
  not materialised; caption row r attends all visual rows and caption rows <= r.
* attention dropout of the decoder (attention_probs_dropout_prob = 0.1, active in the reference's training mode,
  modeling_bert.py:330-333) is applied inside the attention kernels with a counter-based keep decision
  (csrc/rng.h) that the forward and both backward passes recompute -- no mask is stored.  `attn_dropout=0` gives the
  deterministic step the goldens of tests/golden/make_golden_train.py are defined on.
* the tag head runs forward only: `tag_loss` is reported, never back-propagated by this pipeline
  (..._bertemb.py:170-171), and bert.pooler / bert.tag_logit / bert.extra_embeddings / caption_pooler / image head
  receive no gradient.

Gradients of a data-parallel job are summed with ONE RCCL all-reduce per bucket over the flat fp32 gradient buffer
(layout = reverse execution order friendly), launched on a side stream as soon as a bucket's last producer ran.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L
from . import ops
from . import weights as W
from ._lib import check, lib

CH = 1024           # optimizer chunk (elements)
_FUSE_BIAS = os.environ.get('VITCAP_TRAIN_FUSED_BIAS', '1') != '0'      # A/B switch of the measurement in docs/LAB_r01_r04.md 7
_TN_SUM = os.environ.get('VITCAP_TRAIN_TN_SUM', '0') != '0'             # 1: the splits of a weight gradient add themselves up inside the launch
# (vitcap_gemm_tn_sum, bit-identical) instead of a reduce_slabs launch each -- measured SLOWER inside the graph-replayed step (48.85 vs 49.3 ms:
# profiles/r05_train_attn_bwd_ab.txt): the ticket wait and the write-through slabs cost more than 80 cheap launches save
NV = 577
SV = 578
T = 20
LR = SV + T         # decoder rows per image


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lowbias32(x):
    x &= 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def mix32(h, v):
    """vc_mix of csrc/rng.h on Python ints."""
    return _lowbias32(h ^ ((v + 0x9e3779b9 + (h << 6) + (h >> 2)) & 0xffffffff))


def param_groups(names, base_lr, weight_decay, lr_multiplier):
    """name -> (lr, wd) or None (not owned by the optimizer); ..._bertemb.py:280-356."""
    out = {}
    for n in names:
        if n.startswith('module.cls.'):
            out[n] = None
            continue
        low = n.startswith('module.bert.encoder.tag_blocks.') or n.startswith('module.bert.pooler.') \
            or n.startswith('module.bert.tag_logit.')
        if n.startswith('module.bert.encoder.blocks.'):
            low = int(n.split('.')[4]) < 8
        wd = 0.0 if ('bias' in n or 'LayerNorm.weight' in n) else weight_decay
        out[n] = (base_lr * (lr_multiplier if low else 1.0), wd)
    return out


def flat_layout(tied):
    """Flat fp32 parameter / gradient layout: (order, offset, shape, total).  Every unique tensor is padded to CH
    elements (optimizer chunks carry lr / weight decay); the decoder's q|k|v weights, then their biases, are contiguous
    so one GEMM / one bias vector covers the three projections."""
    spec = W.state_dict_spec()
    off, shape, order = {}, {}, []
    for k in spec:
        if tied and k == W.TIED_DST:
            continue
        if '.attention.self.' in k:
            if k.endswith('query.weight'):          # q|k|v weights, then q|k|v biases, contiguous
                pre = k[:-len('query.weight')]
                order += [pre + n + '.weight' for n in ('query', 'key', 'value')]
                order += [pre + n + '.bias' for n in ('query', 'key', 'value')]
            continue
        order.append(k)
    cur = 0
    for k in order:
        shp = spec[k][0]
        n = int(torch.Size(shp).numel())
        off[k], shape[k] = cur, shp
        is_qk_bias = k.endswith('attention.self.query.bias') or k.endswith('attention.self.key.bias')
        cur = cur + n if is_qk_bias else (cur + n + CH - 1) // CH * CH      # q|k|v biases stay contiguous (2304 -> 3 chunks)
    nflat = (cur + CH - 1) // CH * CH
    if tied:
        off[W.TIED_DST], shape[W.TIED_DST] = off[W.TIED_SRC], shape[W.TIED_SRC]
    return order, off, shape, nflat


def chunk_hparams(order, off, shape, nflat, pg):
    """Per-chunk (lr, weight decay) of the fused AdamW; lr 0 = the tensor is not stepped (no gradient in the reference, or
    not owned by its optimizer).  A chunk never mixes tensors with different hyper-parameters."""
    lr = torch.zeros(nflat // CH)
    wd = torch.zeros(nflat // CH)
    owner = {}
    for k in order:
        n = int(torch.Size(shape[k]).numel())
        c0, c1 = off[k] // CH, (off[k] + n + CH - 1) // CH
        hp = (0.0, 0.0) if (pg[k] is None or k.startswith(NO_GRAD_PREFIXES)) else (float(pg[k][0]), float(pg[k][1]))
        for c in (c0, c1 - 1):
            if owner.setdefault(c, hp) != hp:
                raise AssertionError('optimizer chunk %d shared by tensors with different lr / weight decay (%s)' % (c, k))
        lr[c0:c1], wd[c0:c1] = hp
    return lr, wd


NO_GRAD_PREFIXES = ('module.bert.extra_embeddings.', 'module.bert.tag_logit.', 'module.bert.pooler.',
                    'module.bert.caption_pooler.', 'image_encoder.module.head.')


def grad_stage(key):
    """Backward stage after which the gradient of `key` is final, or None if it never receives one.  Stages complete in
    the order of GRAD_STAGES; two consecutive layers share a stage so a bucket is ~57 MB (ring all-reduce over xGMI is
    per-link bound: few large messages)."""
    if key.startswith(NO_GRAD_PREFIXES):
        return None
    if key.startswith('module.cls.'):
        return 'cls'
    if key.startswith('module.bert.decoder.layer.'):
        return 'dec%d' % (int(key.split('.')[4]) // 2)
    if key.startswith('module.bert.embeddings.'):
        return 'emb'
    if key.startswith('module.bert.encoder.tag_blocks.'):
        return 'tag%d' % (int(key.split('.')[4]) // 2)
    if key.startswith('module.bert.encoder.blocks.'):
        return 'blk%d' % (int(key.split('.')[4]) // 2)
    if key.startswith('image_encoder.module.'):
        return 'patch'
    raise KeyError(key)


GRAD_STAGES = ['cls', 'dec1', 'dec0', 'emb', 'tag1', 'tag0', 'blk5', 'blk4', 'blk3', 'blk2', 'blk1', 'blk0', 'patch']


def grad_buckets(order, off, shape):
    """stage -> list of (start, end) element ranges of the flat gradient (adjacent tensors of a stage merged)."""
    out = {st: [] for st in GRAD_STAGES}
    for k in order:
        st = grad_stage(k)
        if st is None:
            continue
        n = int(torch.Size(shape[k]).numel())
        a, b = off[k], off[k] + n
        runs = out[st]
        if runs and a - runs[-1][1] < CH:       # only chunk padding in between
            runs[-1] = (runs[-1][0], b)
        else:
            runs.append((a, b))
    return out


def train_text_inputs_ok(batch):
    """0-d bool tensor (on the device the batch's attention_mask lives on): True when `attention_mask` (B, L, L) and
    `masked_pos` describe the training mask the kernels implement -- tril on the real caption rows ([CLS] .. [SEP]) of the
    first 20 slots and zeros elsewhere, masked positions on real rows only (TrainEngine.check_text_inputs).  None when the
    batch carries no attention_mask (or is a self-critical batch, which has no text inputs)."""
    am = batch.get('attention_mask')
    if am is None or 'sample_ids' in batch:
        return None
    if am.dim() != 3 or am.shape[1] != am.shape[2] or am.shape[1] < T:
        raise NotImplementedError('training attention_mask must be the (B, L, L) seq2seq mask with L >= %d, got %s '
                                  '(mask_type bidirectional is not built)' % (T, tuple(am.shape)))
    Lm = am.shape[1]
    a = am != 0
    n_real = a[:, :T, :T].diagonal(dim1=1, dim2=2).sum(1)                         # real caption rows: [CLS] .. [SEP]
    idx = torch.arange(Lm, device=am.device)
    real = idx.view(1, Lm) < n_real.view(-1, 1)                                   # (B, L)
    want = (idx.view(1, 1, Lm) <= idx.view(1, Lm, 1)) & real.unsqueeze(2) & real.unsqueeze(1)
    ok = (a == want).all()
    mp = batch.get('masked_pos')
    if mp is not None:
        ok = ok & ((mp != 0).to(am.device) <= real[:, :mp.shape[1]]).all()
    return ok


class _FusedLoss(torch.autograd.Function):
    """Loss tensor of a fused forward+backward: .backward() scales the gradients the engine already holds."""
    @staticmethod
    def forward(ctx, anchor, engine, batch):
        ctx.engine = engine
        loss, _ = engine.forward_backward(batch)
        engine._pending = True
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, g):
        eng = ctx.engine
        if not eng._pending:
            raise RuntimeError('backward() called twice on the loss of one fused training forward')
        eng._pending = False
        # forward_backward() already launched the per-bucket all-reduce (+ 1/world scaling) of this same flat buffer on the
        # communication stream: join it BEFORE touching G on the compute stream, or the multiply races with the reduce and the
        # ranks end up with different gradients
        eng.all_reduce_grads()
        # d(sum of losses)/d(masked_loss): 1 in do_train_dict (trainer.py:117-119).  Applied on the device whatever its value --
        # testing it on the host would be a device-to-host synchronisation per step right behind the gradient exchange
        eng.G.mul_(g.to(eng.G.dtype))
        return None, None, None


class TrainEngine(object):
    def __init__(self, model, device='cuda', base_lr=1e-4, weight_decay=0.05, lr_multiplier=0.1, clip=1.0, max_iter=1000,
                 label_smoothing=0.1, dist=None, attn_dropout=0.1, dropout_seed=0, tag_loss='focal', hidden_dropout=0.0):
        """tag_loss: 'focal' (the shipped YAML's `loss: focal`: summed focal loss) or 'bce' (any other `loss`: BCEWithLogitsLoss mean,
        modeling_bert.py:713-717) -- the reported `tag_loss`, never back-propagated by this pipeline."""
        if tag_loss not in ('focal', 'bce'):
            raise ValueError("tag_loss must be 'focal' or 'bce', got %r" % (tag_loss,))
        self.tag_loss = tag_loss
        self.model = model
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device('cuda', torch.cuda.current_device())
        self.clip, self.max_iter, self.eps_ls = clip, max_iter, label_smoothing
        self.dist = dist
        self.attn_dropout = float(attn_dropout)
        self.hidden_dropout = float(hidden_dropout)     # BertConfig.hidden_dropout_prob (the pipeline's drop_out): embeddings, both dense outputs per layer
        # outputs nobody reads are not computed: rows 1..576 of the last tag block, the visual rows of the last decoder layer
        # (forward and backward; same loss and gradients).  VITCAP_TRAIN_FULL_ROWS=1 computes them anyway (A/B measurements).
        self.prune_dead_rows = os.environ.get('VITCAP_TRAIN_FULL_ROWS', '0') != '1'
        rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
        self.dropout_seed = mix32(int(dropout_seed) & 0xffffffff, rank)      # every rank drops differently
        self.step_no = 0
        self.lr_scale = 1.0
        tied = model.tie_weights
        order, self.off, self.shape, self.nflat = flat_layout(tied)
        self.P = torch.zeros(self.nflat, device=self.dev)
        self.G = torch.zeros(self.nflat, device=self.dev)
        self.M = torch.zeros(self.nflat, device=self.dev)
        self.V = torch.zeros(self.nflat, device=self.dev)
        sd = model.state_dict()
        for k in order:
            self.p(k).copy_(sd[k])
        pg = param_groups(order, base_lr, weight_decay, lr_multiplier)
        lr, wd = chunk_hparams(order, self.off, self.shape, self.nflat, pg)
        self.chunk_lr, self.chunk_wd = lr.to(self.dev), wd.to(self.dev)
        self.gsumsq = torch.zeros(1, device=self.dev)
        self.loss_buf = torch.zeros(2, device=self.dev)      # [masked_loss, tag_loss]
        self._gemm_w = {}
        self._ct_table = None
        # DDP's wrap broadcasts rank 0's parameters and buffers to every replica (uni_pipeline.py:497-505): ranks that loaded different
        # files (or none) must not train as one model.  The Adam moments follow after a resume (sync_from_rank0).
        from .dist_util import BucketedAllReduce, broadcast_from_rank0
        self.synced_bytes = broadcast_from_rank0([self.P], dist)
        # device-side NaN / Inf watch on the loss and the gradient norm (trainer.py:134-137 raises on `losses != losses` every step;
        # here the flag is read at the host synchronisations that exist anyway: every 50 steps, before a snapshot, after the last step)
        self._nan_bad = torch.zeros((), dtype=torch.int32, device=self.dev)
        self._nan_first = torch.full((), -1, dtype=torch.int32, device=self.dev)
        self.nan_dump = None              # callable(name) that saves the context (the pipeline passes its Checkpointer.save); default: torch.save here
        self.nan_dump_dir = '.'
        self.refresh_weights()
        self.reducer = BucketedAllReduce(self.G, grad_buckets(order, self.off, self.shape), GRAD_STAGES, dist)
        self._anchor = torch.zeros(1, device=self.dev, requires_grad=True)
        self._pending = False
        self.use_graphs = os.environ.get('VITCAP_TRAIN_GRAPH', '0') == '1'      # train_step() replays captured segments (graph mode below)
        self._graphs = {}
        model.__dict__['train_engine'] = self          # ImageCaptioning.forward (training mode) routes here

    def loss_dict(self, batch):
        """What ImageCaptioning.forward returns in training mode (..._bertemb.py:170-171)."""
        return {'masked_loss': _FusedLoss.apply(self._anchor, self, batch)}

    def check_text_inputs(self, batch):
        """The training kernels hard-wire the mask CaptionTensorizer.tensorize_ab builds for a caption without text_b
        (dataset.py:377-390, mask_type seq2seq): token row i attends the visual rows and token rows <= i; rows behind the
        caption's [SEP] are padding whose outputs the loss never reads.  A caller's `attention_mask` (B, L, L) is verified
        against exactly that -- tril on the real caption rows, zeros elsewhere (no text_b columns, no seq2seq_off diagonal,
        no bidirectional vector) -- and `masked_pos` must select real caption rows only; anything else is refused instead of
        being trained with a different mask than the caller asked for.
        Host tensors (what the loader yields) are checked on the host.  Tensors that already live on the device are checked
        with a host read on the first step only; afterwards the comparison runs on their stream into a device counter that is
        read every 50 steps (a per-step read would be a device-to-host synchronisation in front of every step)."""
        ok = train_text_inputs_ok(batch)
        if ok is None:
            return
        if not ok.is_cuda or not getattr(self, '_text_checked', False):
            self._text_checked = True
            if not bool(ok):
                raise NotImplementedError(
                    'training attention_mask / masked_pos do not describe the mask the HIP training kernels implement (tril on the '
                    'real caption rows of the first %d slots, zeros elsewhere: tensorize_ab without text_b, mask_type seq2seq, '
                    'dataset.py:377-390)' % T)
            return
        if getattr(self, '_text_bad', None) is None:
            self._text_bad = torch.zeros((), dtype=torch.int32, device=ok.device)
        self._text_bad += (~ok).to(torch.int32)
        if self.step_no % 50 == 49:
            self.flush_text_check()

    def sync_from_rank0(self):
        """Parameters AND Adam moments of every rank := rank 0's (after a resume: a rank that read a stale or different snapshot would
        otherwise keep its own moments); bf16 operand copies rebuilt.  No-op without a process group."""
        from .dist_util import broadcast_from_rank0
        n = broadcast_from_rank0([self.P, self.M, self.V], self.dist)
        if n:
            self.refresh_weights()
        return n

    def _watch_nan(self):
        """Enqueued once per step behind vitcap_sumsq: counts the steps whose loss or gradient norm is not finite (no host read)."""
        bad = ~(torch.isfinite(self.loss_buf[0]) & torch.isfinite(self.gsumsq[0]))
        self._nan_first.copy_(torch.where(bad & (self._nan_first < 0), torch.full_like(self._nan_first, self.step_no), self._nan_first))
        self._nan_bad += bad.to(torch.int32)

    def flush_nan_check(self):
        """trainer.py:134-137: `if losses != losses: checkpointer.save("NaN_context_{rank}"); raise RuntimeError('NaN encountered!')`.
        The reference tests the loss on the host after every step; here the device-side flag of _watch_nan (loss OR gradient norm not
        finite -- an Inf gradient reaches the parameters one step before the loss shows it) is read where the host waits anyway."""
        if int(self._nan_bad) == 0:
            return
        first, n = int(self._nan_first), int(self._nan_bad)
        self._nan_bad.zero_()
        self._nan_first.fill_(-1)
        rank = self.dist.get_rank() if (self.dist is not None and self.dist.is_initialized()) else 0
        name = 'NaN_context_{}'.format(rank)
        import logging
        logging.info('NaN encountered! (first at step %d, %d step(s) since the last check)', first, n)
        if self.nan_dump is not None:
            self.nan_dump(name)
        else:
            torch.save({'model': self.state_dict(), 'iteration': self.step_no, 'first_bad_step': first,
                        'optimizer': self.optimizer_state_dict(), 'scheduler': self.scheduler_state_dict()},
                       os.path.join(self.nan_dump_dir, name + '.pt'))
        raise RuntimeError('NaN encountered!')

    def flush_text_check(self):
        """Reads the device-side counter of batches whose text tensors the kernels do not implement (one host synchronisation) and
        the NaN watch (flush_nan_check).
        Called every 50 steps, and by the pipeline before every checkpoint it saves and after the last step (otherwise a
        snapshot could be written from up to 49 steps trained on the hard-wired mask, and the final steps were never reported)."""
        self.flush_nan_check()
        bad = getattr(self, '_text_bad', None)
        if bad is not None and int(bad) != 0:
            self._text_bad = None
            raise NotImplementedError('a training batch since the last check carried an attention_mask / masked_pos the HIP '
                                      'training kernels do not implement (TrainEngine.check_text_inputs)')

    # ------------------------------------------------------------------ views
    def p(self, k):
        n = int(torch.Size(self.shape[k]).numel())
        return self.P[self.off[k]:self.off[k] + n].view(self.shape[k])

    def g(self, k):
        n = int(torch.Size(self.shape[k]).numel())
        return self.G[self.off[k]:self.off[k] + n].view(self.shape[k])

    def _flat(self, buf, k, n, shape):
        return buf[self.off[k]:self.off[k] + n].view(shape)

    def state_dict(self):
        return {k: self.p(k).detach().clone() for k in W.state_dict_spec()}

    # optimizer / scheduler payloads of the reference checkpoint (src/tools/opt/checkpoint.py:60-75 saves
    # optimizer.state_dict() and scheduler.state_dict() next to the model; trainer.py:95 resumes from 'iteration')
    def optimizer_state_dict(self):
        return {'format': 'vitcap_amd.flat_adamw.v1', 'nflat': self.nflat, 'step': self.step_no,
                'exp_avg': self.M.detach().cpu(), 'exp_avg_sq': self.V.detach().cpu()}

    def load_optimizer_state_dict(self, sd):
        if sd.get('format') != 'vitcap_amd.flat_adamw.v1' or int(sd.get('nflat', -1)) != self.nflat:
            raise ValueError('optimizer state of another layout (%r, %r elements; this engine holds %d)'
                             % (sd.get('format'), sd.get('nflat'), self.nflat))
        self.M.copy_(sd['exp_avg'])
        self.V.copy_(sd['exp_avg_sq'])
        self.step_no = int(sd['step'])

    def scheduler_state_dict(self):
        return {'last_epoch': self.step_no, 'lr_scale': self.lr_scale, 't_total': self.max_iter}

    def load_scheduler_state_dict(self, sd):
        self.lr_scale = float(sd['lr_scale'])

    def load_model_state_dict(self, sd):
        """Parameters from a checkpoint's 'model' dict (resume): flat fp32 master copy + the bf16 GEMM operands."""
        for k in W.state_dict_spec():
            if k in sd:
                self.p(k).copy_(sd[k])
        self.refresh_weights()

    # ------------------------------------------------------------------ bf16 operands of the GEMMs
    def _matrices(self):
        """(name, first key, N, K, pad_N): GEMM weights; decoder q|k|v are one [2304,768] matrix in the flat layout."""
        out = [('patch', 'image_encoder.module.patch_embed.proj.weight', 768, 768, 768)]
        for pre in ['module.bert.encoder.blocks.%d' % i for i in range(12)] + \
                   ['module.bert.encoder.tag_blocks.%d' % i for i in range(4)]:
            out += [(pre + '.qkv', pre + '.attn.qkv.weight', 2304, 768, 2304), (pre + '.proj', pre + '.attn.proj.weight', 768, 768, 768),
                    (pre + '.fc1', pre + '.mlp.fc1.weight', 3072, 768, 3072), (pre + '.fc2', pre + '.mlp.fc2.weight', 768, 3072, 768)]
        for i in range(4):
            pre = 'module.bert.decoder.layer.%d' % i
            out += [(pre + '.qkv', pre + '.attention.self.query.weight', 2304, 768, 2304),
                    (pre + '.ao', pre + '.attention.output.dense.weight', 768, 768, 768),
                    (pre + '.i', pre + '.intermediate.dense.weight', 3072, 768, 3072),
                    (pre + '.o', pre + '.output.dense.weight', 768, 3072, 768)]
        out += [('pooler', 'module.bert.pooler.dense.weight', 768, 768, 768),
                ('tag.t', 'module.bert.tag_logit.predictions.transform.dense.weight', 768, 768, 768),
                ('tag.dec', 'module.bert.tag_logit.predictions.decoder.weight', 30522, 768, L.VOCAB_PAD),
                ('cls.t', 'module.cls.predictions.transform.dense.weight', 768, 768, 768),
                ('word', 'module.bert.embeddings.word_embeddings.weight', 30522, 768, L.VOCAB_PAD),
                ('pos', 'module.bert.embeddings.position_embeddings.weight', 512, 768, 512),
                ('type', 'module.bert.embeddings.token_type_embeddings.weight', 2, 768, 8)]
        if not self.model.tie_weights:
            out.append(('cls.dec', 'module.cls.predictions.decoder.weight', 30522, 768, L.VOCAB_PAD))
        return out

    def refresh_weights(self):
        """fp32 masters -> bf16 W[N][K] and W^T[K][N] (forward / dgrad operands); called after every optimizer step.  ONE launch for
        all matrices (vitcap_cast_transpose_multi over a device-resident table built once): as ~110 launches of a few microseconds
        each they were launch floor (0.74 ms of kernel time per step; same-box A/B 55.78 -> 55.44 ms)."""
        if self._ct_table is None:
            items, tile0 = [], 0
            for name, key, N, K, padN in self._matrices():
                self._gemm_w[name] = (torch.zeros(padN, K, device=self.dev, dtype=torch.bfloat16),
                                      torch.zeros(K, padN, device=self.dev, dtype=torch.bfloat16))
                wb, wt = self._gemm_w[name]
                assert K % 64 == 0 and padN >= N and padN % 8 == 0
                src = self.P[self.off[key]:self.off[key] + N * K]
                items.append(L.CtItem(src.data_ptr(), wb.data_ptr(), wt.data_ptr(), N, K, padN, tile0))
                tile0 += ((N + 63) // 64) * (K // 64)
            raw = bytes((L.CtItem * len(items))(*items))
            self._ct_table = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev), len(items), tile0)
            if self.model.tie_weights:
                self._gemm_w['cls.dec'] = self._gemm_w['word']
        tab, n, tiles = self._ct_table
        check(lib.vitcap_cast_transpose_multi(_p(tab), n, tiles, _s()), 'cast_transpose_multi')

    def bind_inference(self):
        """Points the model's INFERENCE engine at this engine's own device buffers (bf16 matrices refreshed after every
        optimizer step, fp32 vectors = views of the master buffer): generate() then always decodes with the current
        weights, with no copy -- what the self-critical step needs between its sampling and its gradient pass."""
        m, w = self.model, L.Weights()
        P = lambda t: C.c_void_p(t.data_ptr())       # noqa: E731
        vec = self.vec
        ie = 'image_encoder.module.'
        w.patch_w, w.patch_b = P(self.wb('patch')), P(vec(ie + 'patch_embed.proj.bias'))
        w.cls_token, w.pos_embed = P(vec(ie + 'cls_token')), P(vec(ie + 'pos_embed'))

        def vit_block(dst, pre):
            dst.qkv_w, dst.qkv_b = P(self.wb(pre + '.qkv')), P(vec(pre + '.attn.qkv.bias'))
            dst.proj_w, dst.proj_b = P(self.wb(pre + '.proj')), P(vec(pre + '.attn.proj.bias'))
            dst.fc1_w, dst.fc1_b = P(self.wb(pre + '.fc1')), P(vec(pre + '.mlp.fc1.bias'))
            dst.fc2_w, dst.fc2_b = P(self.wb(pre + '.fc2')), P(vec(pre + '.mlp.fc2.bias'))
            dst.n1_g, dst.n1_b = P(vec(pre + '.norm1.weight')), P(vec(pre + '.norm1.bias'))
            dst.n2_g, dst.n2_b = P(vec(pre + '.norm2.weight')), P(vec(pre + '.norm2.bias'))
        for i in range(12):
            vit_block(w.blocks[i], 'module.bert.encoder.blocks.%d' % i)
        for i in range(4):
            vit_block(w.tag_blocks[i], 'module.bert.encoder.tag_blocks.%d' % i)
        w.pooler_w, w.pooler_b = P(self.wb('pooler')), P(vec('module.bert.pooler.dense.bias'))

        def padded_bias(key):       # the master chunk behind a 30522-vector is zero-padded to 30720 >= VOCAB_PAD
            o = self.off[key]
            return self.P[o:o + L.VOCAB_PAD]

        def lm_head(dst, pre, tname, dname):
            dst.dense_w, dst.dense_b = P(self.wb(tname)), P(vec(pre + '.predictions.transform.dense.bias'))
            dst.ln_g = P(vec(pre + '.predictions.transform.LayerNorm.weight'))
            dst.ln_b = P(vec(pre + '.predictions.transform.LayerNorm.bias'))
            dst.dec_w, dst.dec_b = P(self.wb(dname)), P(padded_bias(pre + '.predictions.bias'))
        lm_head(w.tag_logit, 'module.bert.tag_logit', 'tag.t', 'tag.dec')
        lm_head(w.cls, 'module.cls', 'cls.t', 'cls.dec')
        e = 'module.bert.embeddings'
        w.word_emb, w.pos_emb, w.type_emb = P(self.wb('word')), P(self.wb('pos')), P(self.wb('type'))
        w.emb_ln_g, w.emb_ln_b = P(vec(e + '.LayerNorm.weight')), P(vec(e + '.LayerNorm.bias'))
        for i in range(4):
            pre, d = 'module.bert.decoder.layer.%d' % i, w.dec[i]
            d.qkv_w, d.qkv_b = P(self.wb(pre + '.qkv')), P(self.qkv_bias(pre))
            d.ao_w, d.ao_b = P(self.wb(pre + '.ao')), P(vec(pre + '.attention.output.dense.bias'))
            d.ao_g, d.ao_beta = P(vec(pre + '.attention.output.LayerNorm.weight')), P(vec(pre + '.attention.output.LayerNorm.bias'))
            d.i_w, d.i_b = P(self.wb(pre + '.i')), P(vec(pre + '.intermediate.dense.bias'))
            d.o_w, d.o_b = P(self.wb(pre + '.o')), P(vec(pre + '.output.dense.bias'))
            d.o_g, d.o_beta = P(vec(pre + '.output.LayerNorm.weight')), P(vec(pre + '.output.LayerNorm.bias'))
        if m._engine is None:
            h = C.c_void_p()
            check(lib.vitcap_engine_create(C.byref(h)), 'engine_create')
            m._engine = h
        check(lib.vitcap_engine_bind_weights(m._engine, C.byref(w)), 'bind_weights')
        m._packed = (w, [self], self.dev)
        return m

    def wb(self, name):
        return self._gemm_w[name][0]

    def wt(self, name):
        return self._gemm_w[name][1]

    def vec(self, k):
        return self.p(k).view(-1)

    def qkv_bias(self, pre):      # decoder q|k|v biases are contiguous in the flat layout
        o = self.off[pre + '.attention.self.query.bias']
        return self.P[o:o + 2304]

    def qkv_bias_grad(self, pre):
        o = self.off[pre + '.attention.self.query.bias']
        return self.G[o:o + 2304]

    def qkv_w_grad(self, pre):
        o = self.off[pre + '.attention.self.query.weight']
        return self.G[o:o + 2304 * 768].view(2304, 768)

    # ------------------------------------------------------------------ helpers
    def _wgrad_tn(self, dy, x, gview, bias_grad):
        """gview[N][K] = dy^T x and bias_grad[N] += column sums of dy, straight from the row-major backward operands
        (LDS transpose reads, no transposed copies); M is split so that about one wave of 256x256 tiles fills the chip.
        bias_grad=None: the kernel that produced dy has already added its column sums (ops.gemm_ex(colsum=),
        ops.layernorm_bwd(dxb_colsum=), ops.cast_bf16_colsum)."""
        N, Kin = gview.shape
        tiles = (N // 256) * (Kin // 256)
        stages = (dy.shape[0] + 63) // 64
        S = max(1, min(stages, 256 // tiles))
        if S == 1:
            ops.gemm_tn(dy, x, 1, slabs=gview.view(1, N, Kin))
        elif _TN_SUM and gview.is_contiguous():
            ops.gemm_tn_sum(dy, x, S, gview)                     # the splits add themselves up inside the launch (round 5)
        else:
            ops.reduce_slabs(ops.gemm_tn(dy, x, S), gview)
        if bias_grad is not None:
            ops.colsum_bf16(dy, bias_grad)

    def _wgrad(self, dyT, xT, gview, accumulate=False):
        """gview[N][K] (+)= dyT[N][Mp] @ xT[K][Mp]^T with split-K sized to fill the chip."""
        N, Kin = dyT.shape[0], xT.shape[0]
        tiles = ((N + 127) // 128) * ((Kin + 127) // 128)
        nk = dyT.shape[1] // 64
        S = max(1, min(32, nk, (640 + tiles - 1) // tiles))
        if S == 1:
            if accumulate:
                tmp = ops.gemm_ex(dyT, xT, out_dtype=torch.float32)
                ops.reduce_slabs(tmp.view(1, N, Kin), gview, accumulate=True)
            else:
                ops.gemm_ex(dyT, xT, out=gview)
            return
        slabs = ops.gemm_ex(dyT, xT, split_k=S)
        ops.reduce_slabs(slabs, gview, accumulate=accumulate)

    def _dgrad_gelu(self, dy, wt, z, bias_grad):
        """dz = (dy @ W) * gelu'(z) (bf16) with the bias gradient of the layer dz is the output gradient of: added by the GEMM's
        own epilogue where the 256x256 kernel runs (M >= 2048), by a separate pass over dz otherwise.  Returns (dz, pending) --
        pending is the bias gradient still to be added by _wgrad_tn, or None."""
        if _FUSE_BIAS and dy.shape[0] >= 2048:
            return ops.gemm_ex(dy, wt, aux=z, colsum=bias_grad), None
        return ops.gemm_ex(dy, wt, aux=z), bias_grad

    def _ln_bwd(self, x, dy, gkey, bkey, eps, dres=None, dxb_colsum=None):
        dxf, dxb = ops.layernorm_bwd(x, dy, self.vec(gkey), eps, self.g(gkey).view(-1), self.g(bkey).view(-1), dres=dres,
                                     dxb_colsum=dxb_colsum if _FUSE_BIAS else None)
        if dxb_colsum is not None and not _FUSE_BIAS:
            ops.colsum_bf16(dxb, dxb_colsum)
        return dxf, dxb

    # ------------------------------------------------------------------ forward + backward
    def forward_backward(self, batch):
        """One fused forward + backward (see _fb_gen): drives the stage generator and starts a bucket's gradient exchange after each
        backward stage."""
        self.reducer.begin()
        gen = self._fb_gen(batch)
        while True:
            try:
                stage = next(gen)
            except StopIteration as done:
                return done.value
            self.reducer.stage_done(stage)

    def _fb_gen(self, batch, capture=False):
        """Generator: yields the name of every backward stage right after that stage's last kernel was enqueued (the driver starts the
        stage's gradient exchange there, or -- graph mode -- ends a captured segment), returns (masked_loss, tag_loss).
        capture=True: the body is being recorded into hipGraph segments (train_step_graph): no host synchronisation, dropout seeds
        of step 0 (the per-step variation comes from the device-resident salt, vitcap_set_dropout_salt).

        batch: image (B,3,384,384) fp32/bf16 cuda; input_ids (B,70) int64; masked_pos (B,70) int; masked_ids (B,3) int64;
        label (B,30522) fp32 -- the dict the reference's collate feeds ImageCaptioning.forward in training.
        Accumulates gradients into self.G (zeroed here) and returns (masked_loss, tag_loss) device scalars.

        Self-critical mode (`sample_ids` (B,20) int64 + `sample_weight` (B,) fp32 in the batch instead of the masked-token
        fields): the loss is sum_b sample_weight[b] * mean_t(-log p(sample_ids[b,t] | sample_ids[b,:t], image)) over the
        generated positions, i.e. ScstRewardCriterion with sample_weight = (reward - baseline) / B.  The 19 per-step
        forwards the reference differentiates through (one full re-encode per generated token) collapse into ONE pass:
        the decoder runs on [578 visual | 20 token rows | 19 [MASK] probe rows], probe j sees tokens 0..j."""
        dev = self.dev
        if capture:       # the device-side comparison only (read by flush_text_check every 50 steps, outside the graph)
            ok = train_text_inputs_ok(batch)
            if ok is not None:
                if getattr(self, '_text_bad', None) is None:
                    self._text_bad = torch.zeros((), dtype=torch.int32, device=ok.device)
                self._text_bad += (~ok).to(torch.int32)
        else:
            self.check_text_inputs(batch)
        seed_step = 0 if capture else self.step_no
        img = batch['image']
        Be = img.shape[0]                       # images the encoder runs on
        KS = int(batch.get('seq_per_image', 1))   # decoder sequences per image (self-critical step: the samples of an image
        B = Be * KS                             # share its encoder pass; their visual-row gradients are summed before its backward)
        scst = 'sample_ids' in batch
        TP = (T - 1) if scst else 0          # probe rows per sequence
        TT = T + TP                          # text rows per sequence
        LR = SV + TT                         # decoder rows per image (shadows the module constant on purpose)
        M, Md = Be * NV, B * LR
        self.G.zero_()
        self.loss_buf.zero_()
        ie = 'image_encoder.module.'
        # ================= forward: patch embed
        patches = ops.patch_gather(img)
        x = torch.empty(M, 768, device=dev)
        pos = self.p(ie + 'pos_embed').view(NV, 768)
        ops.gemm_bias_act(patches, self.wb('patch'), self.vec(ie + 'patch_embed.proj.bias'), residual=pos[1:], out=x,
                          row_group=576, out_group_rows=NV, out_row_off=1, res_periodic=1)
        check(lib.vitcap_cls_rows(_p(self.vec(ie + 'cls_token')), _p(pos), _p(x), Be, NV, _s()), 'cls_rows')
        # ================= forward: 12 blocks + 4 tag blocks, activations kept
        saved = {}

        def block_fwd(pre, xin):
            h1, _ = ops.layernorm(xin, self.vec(pre + '.norm1.weight'), self.vec(pre + '.norm1.bias'), 1e-6)
            qkv = ops.gemm_bias_act(h1, self.wb(pre + '.qkv'), self.vec(pre + '.attn.qkv.bias'))
            ao, lse = ops.attn_dense_train(qkv, Be, NV)
            xmid = torch.empty(M, 768, device=dev)
            ops.gemm_bias_act(ao, self.wb(pre + '.proj'), self.vec(pre + '.attn.proj.bias'), residual=xin, out=xmid)
            h2, _ = ops.layernorm(xmid, self.vec(pre + '.norm2.weight'), self.vec(pre + '.norm2.bias'), 1e-6)
            z = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
            g = ops.gemm_ex(h2, self.wb(pre + '.fc1'), bias=self.vec(pre + '.mlp.fc1.bias'), act=L.ACT_GELU_ERF, zout=z)
            xout = torch.empty(M, 768, device=dev)
            ops.gemm_bias_act(g, self.wb(pre + '.fc2'), self.vec(pre + '.mlp.fc2.bias'), residual=xmid, out=xout)
            saved[pre] = (xin, h1, qkv, ao, lse, xmid, h2, z, g)
            return xout

        def block_fwd_cls(pre, xin):
            """The same block when only row 0 (CLS) of its output is ever read -- the last tag block: tag_hidden feeds the pooler
            and the joint sequence through tag_hidden[:, 0] alone (modeling_bert.py:1424, 1493).  LN1 and the qkv projection
            run on every row (the CLS query attends all keys, and K/V carry gradient to every row); attention, proj, LN2 and
            the MLP run for the B CLS rows.  Returns (B,768)."""
            h1, _ = ops.layernorm(xin, self.vec(pre + '.norm1.weight'), self.vec(pre + '.norm1.bias'), 1e-6)
            qkv = ops.gemm_bias_act(h1, self.wb(pre + '.qkv'), self.vec(pre + '.attn.qkv.bias'))
            ao, lse = ops.attn_dense_train(qkv, Be, NV, q_range=(0, 1))
            ao_c = ao.view(Be, NV, 768)[:, 0].contiguous()
            xin_c = xin.view(Be, NV, 768)[:, 0].contiguous()
            xmid_c = torch.empty(Be, 768, device=dev)
            ops.gemm_bias_act(ao_c, self.wb(pre + '.proj'), self.vec(pre + '.attn.proj.bias'), residual=xin_c, out=xmid_c)
            h2_c, _ = ops.layernorm(xmid_c, self.vec(pre + '.norm2.weight'), self.vec(pre + '.norm2.bias'), 1e-6)
            z_c = torch.empty(Be, 3072, device=dev, dtype=torch.bfloat16)
            g_c = ops.gemm_ex(h2_c, self.wb(pre + '.fc1'), bias=self.vec(pre + '.mlp.fc1.bias'), act=L.ACT_GELU_ERF, zout=z_c)
            xout_c = torch.empty(Be, 768, device=dev)
            ops.gemm_bias_act(g_c, self.wb(pre + '.fc2'), self.vec(pre + '.mlp.fc2.bias'), residual=xmid_c, out=xout_c)
            saved[pre] = (xin, h1, qkv, ao, lse, xmid_c, h2_c, z_c, g_c, ao_c)
            return xout_c

        prune = self.prune_dead_rows
        xt = None
        for i in range(12):
            if i == 8:
                xt = x
            x = block_fwd('module.bert.encoder.blocks.%d' % i, x)
        for i in range(3):
            xt = block_fwd('module.bert.encoder.tag_blocks.%d' % i, xt)
        if prune:
            xt_cls = block_fwd_cls('module.bert.encoder.tag_blocks.3', xt)
        else:
            xt_cls = block_fwd('module.bert.encoder.tag_blocks.3', xt).view(Be, NV, 768)[:, 0].contiguous()
        # ================= forward: tag head (value of tag_loss only)
        tg = 'module.bert.tag_logit.predictions'
        pin = ops.cast_bf16(xt_cls)
        pooled = ops.gemm_bias_act(pin, self.wb('pooler'), self.vec('module.bert.pooler.dense.bias'), act=L.ACT_TANH)
        tgf = ops.gemm_bias_act(pooled, self.wb('tag.t'), self.vec(tg + '.transform.dense.bias'), act=L.ACT_GELU_ERF,
                                out_dtype=torch.float32)
        tgb, _ = ops.layernorm(tgf, self.vec(tg + '.transform.LayerNorm.weight'), self.vec(tg + '.transform.LayerNorm.bias'), 1e-12)
        tbias = torch.zeros(L.VOCAB_PAD, device=dev)
        tbias[:L.VOCAB] = self.vec(tg + '.bias')
        tag_logits = ops.gemm_bias_act(tgb, self.wb('tag.dec'), tbias, out_dtype=torch.float32)
        if 'label' in batch:                 # reported only (never back-propagated); the self-critical step has no labels
            label = batch['label'].to(dev).contiguous()
            if self.tag_loss == 'focal':       # FocalLossWithLogitsNegLoss(alpha .5, gamma 1).sum(), modeling_bert.py:713-715, 789-791
                check(lib.vitcap_focal_loss_sum(_p(tag_logits), L.VOCAB_PAD, L.VOCAB, _p(label), 0.5, _p(self.loss_buf[1:]), Be, _s()),
                      'focal')
            else:                              # torch.nn.BCEWithLogitsLoss(), modeling_bert.py:716-717
                check(lib.vitcap_bce_logits_mean(_p(tag_logits), L.VOCAB_PAD, L.VOCAB, _p(label), _p(self.loss_buf[1:]), Be, _s()), 'bce')
        # ================= forward: decoder on [578 visual | 20 caption] rows per image
        e = 'module.bert.embeddings'
        if scst:
            sid = batch['sample_ids'].to(dev).view(B, T)
            ids20 = torch.cat([sid, torch.full((B, TP), 103, dtype=torch.int64, device=dev)], 1).contiguous()   # tokens | [MASK] x19
        else:
            ids20 = batch['input_ids'][:, :T].to(dev).contiguous()
        pre_emb = torch.empty(B * TT, 768, device=dev)
        xtext = torch.empty(B * TT, 768, device=dev)
        check(lib.vitcap_embed_rows(_p(ids20), TT, _p(self.wb('word')), _p(self.wb('pos')), _p(self.wb('type')),
                                    _p(self.vec(e + '.LayerNorm.weight')), _p(self.vec(e + '.LayerNorm.bias')), 1e-12,
                                    _p(pre_emb), _p(xtext), None, B * TT, T if scst else 0, _s()), 'embed_rows')
        ph = self.hidden_dropout

        def hseed(l, site):            # per step, per layer (4 = embeddings), per site (1 attention.output, 2 output, 3 embeddings)
            return mix32(mix32(mix32(self.dropout_seed, seed_step), 16 + l), site)
        if ph > 0:                     # BertEmbeddings: dropout(LayerNorm(...)) on the text rows (device rows SV .. SV + TT - 1 of a sequence)
            ops.hidden_dropout(xtext, None, TT, SV, hseed(4, 3), ph, out=xtext)
        dx = torch.empty(B, LR, 768, device=dev)
        dx[:, 0] = xt_cls if KS == 1 else xt_cls.repeat_interleave(KS, 0)
        dx[:, 1:SV] = x.view(Be, NV, 768) if KS == 1 else x.view(Be, NV, 768).repeat_interleave(KS, 0)
        dx[:, SV:] = xtext.view(B, TT, 768)
        xd = dx.view(Md, 768)
        dsaved = []
        pd = self.attn_dropout
        dseed = [mix32(mix32(self.dropout_seed, seed_step), l) for l in range(4)]      # per step, per layer
        QLO = (SV // 128) * 128                       # first 128-row query block that holds text rows
        for l in range(4):
            pre = 'module.bert.decoder.layer.%d' % l
            xb = ops.cast_bf16(xd)
            qkv = ops.gemm_bias_act(xb, self.wb(pre + '.qkv'), self.qkv_bias(pre))
            if l == 3 and prune:
                # Last layer: only the text rows' outputs are read (LM head on the masked / probe rows); the visual rows' K/V
                # are still needed (and carry gradient), their attention / output / MLP are not.  Text rows only from here.
                ctx, lse = ops.attn_dense_train(qkv, B, LR, p_drop=pd, drop_seed=dseed[l], causal_from=SV,
                                                mask_from=SV + T if scst else 0, q_range=(QLO, LR))
                ctx_t = ctx.view(B, LR, 768)[:, SV:].reshape(B * TT, 768).contiguous()
                xd_t = xd.view(B, LR, 768)[:, SV:].reshape(B * TT, 768).contiguous()
                t1 = torch.empty(B * TT, 768, device=dev)
                if ph > 0:      # BertSelfOutput: LayerNorm(dropout(dense(ctx)) + x)
                    ops.gemm_bias_act(ctx_t, self.wb(pre + '.ao'), self.vec(pre + '.attention.output.dense.bias'), out=t1)
                    ops.hidden_dropout(t1, xd_t, TT, SV, hseed(l, 1), ph, out=t1)
                else:
                    ops.gemm_bias_act(ctx_t, self.wb(pre + '.ao'), self.vec(pre + '.attention.output.dense.bias'), residual=xd_t, out=t1)
                ab, af = ops.layernorm(t1, self.vec(pre + '.attention.output.LayerNorm.weight'),
                                       self.vec(pre + '.attention.output.LayerNorm.bias'), 1e-12, want_f32=True)
                z = torch.empty(B * TT, 3072, device=dev, dtype=torch.bfloat16)
                it = ops.gemm_ex(ab, self.wb(pre + '.i'), bias=self.vec(pre + '.intermediate.dense.bias'), act=L.ACT_GELU_ERF, zout=z)
                t2 = torch.empty(B * TT, 768, device=dev)
                if ph > 0:      # BertOutput: LayerNorm(dropout(dense(intermediate)) + attention_output)
                    ops.gemm_bias_act(it, self.wb(pre + '.o'), self.vec(pre + '.output.dense.bias'), out=t2)
                    ops.hidden_dropout(t2, af, TT, SV, hseed(l, 2), ph, out=t2)
                else:
                    ops.gemm_bias_act(it, self.wb(pre + '.o'), self.vec(pre + '.output.dense.bias'), residual=af, out=t2)
                _, text_out = ops.layernorm(t2, self.vec(pre + '.output.LayerNorm.weight'), self.vec(pre + '.output.LayerNorm.bias'),
                                            1e-12, want_bf16=False, want_f32=True)
                dsaved.append((xb, qkv, ctx, lse, t1, ab, af, z, it, t2, ctx_t))
                break
            # visual rows attend visual rows; caption row q attends all visual rows and caption rows <= q (one kernel)
            ctx, lse = ops.attn_dense_train(qkv, B, LR, p_drop=pd, drop_seed=dseed[l], causal_from=SV,
                                            mask_from=SV + T if scst else 0)
            t1 = torch.empty(Md, 768, device=dev)
            if ph > 0:
                ops.gemm_bias_act(ctx, self.wb(pre + '.ao'), self.vec(pre + '.attention.output.dense.bias'), out=t1)
                ops.hidden_dropout(t1, xd, LR, 0, hseed(l, 1), ph, out=t1)
            else:
                ops.gemm_bias_act(ctx, self.wb(pre + '.ao'), self.vec(pre + '.attention.output.dense.bias'), residual=xd, out=t1)
            ab, af = ops.layernorm(t1, self.vec(pre + '.attention.output.LayerNorm.weight'),
                                   self.vec(pre + '.attention.output.LayerNorm.bias'), 1e-12, want_f32=True)
            z = torch.empty(Md, 3072, device=dev, dtype=torch.bfloat16)
            it = ops.gemm_ex(ab, self.wb(pre + '.i'), bias=self.vec(pre + '.intermediate.dense.bias'), act=L.ACT_GELU_ERF, zout=z)
            t2 = torch.empty(Md, 768, device=dev)
            if ph > 0:
                ops.gemm_bias_act(it, self.wb(pre + '.o'), self.vec(pre + '.output.dense.bias'), out=t2)
                ops.hidden_dropout(t2, af, LR, 0, hseed(l, 2), ph, out=t2)
            else:
                ops.gemm_bias_act(it, self.wb(pre + '.o'), self.vec(pre + '.output.dense.bias'), residual=af, out=t2)
            _, yf = ops.layernorm(t2, self.vec(pre + '.output.LayerNorm.weight'), self.vec(pre + '.output.LayerNorm.bias'), 1e-12,
                                  want_bf16=False, want_f32=True)
            dsaved.append((xb, qkv, ctx, lse, t1, ab, af, z, it, t2))
            xd = yf
        if not prune:
            text_out = xd.view(B, LR, 768)[:, SV:].reshape(B * TT, 768)
        # ================= loss on the masked caption positions
        row_w = None
        if scst:
            # probe j (grid column T + j) predicts position j+1; a position counts while the sequence is unfinished, i.e. up
            # to and including its first [SEP] (modeling_utils.py:853-877); coefficient = weight_b / (#counted positions)
            tok = sid[:, 1:]
            ended = ((sid[:, :-1] == 102).cumsum(1) > 0)                     # a [SEP] strictly before this position
            unf = ~ended
            mp = torch.zeros(B, TT, dtype=torch.bool, device=dev)
            mp[:, T:] = unf
            cnt = unf.sum(1).clamp(min=1).to(torch.float32)
            wrow = (batch['sample_weight'].to(dev).to(torch.float32) / cnt)[:, None].expand(B, TP)
            row_w = wrow[unf].contiguous()
            tgt = tok[unf].contiguous()
            sel = mp.view(-1).nonzero().view(-1)                              # rows of the (B*TT) text grid
            n = int(sel.numel())
            assert int(tgt.numel()) == n, 'sample_ids / unfinished mask disagree'
            sel_bwd = sel
        else:
            # Static form (no host synchronisation, same shapes every step: graph-capturable).  Slot (b, j) = the j-th masked position of
            # sample b; masked_ids[b, j] == 0 marks an unused slot (a caption with fewer than 3 masked tokens).  The used slots are moved
            # to the front in their original order (stable sort), so with every slot used -- the synthetic batches -- rows, targets and
            # weights are exactly what `nonzero()` / boolean indexing produced; an unused slot becomes a row with weight 0 (zero loss,
            # zero gradient rows: exact zeros in every sum) whose gradient lands in a dummy row behind the text grid.
            mpos = batch['masked_pos'][:, :T].to(dev) != 0                       # (B, T)
            mids = batch['masked_ids'].to(dev)                                   # (B, J)
            J = mids.shape[1]
            cum = mpos.to(torch.int32).cumsum(1)                                 # 1-based rank of every masked position
            want = torch.arange(1, J + 1, device=dev, dtype=torch.int32).view(1, J, 1)
            tpos = ((cum.unsqueeze(1) == want) & mpos.unsqueeze(1)).to(torch.int32).argmax(2)      # (B, J): position of the j-th masked token
            rows_all = (torch.arange(B, device=dev).view(B, 1) * TT + tpos).view(-1)
            used = (mids != 0).view(-1)
            # a batch whose masked_pos and masked_ids disagree is reported like a bad mask (device counter, TrainEngine.flush_text_check)
            # (counts equal AND the used slots are the leading ones: dataset.py:321-324 pads masked_ids behind the real ids; a zero in
            # front of a real id would pair targets with the wrong rows -- ADVICE r5)
            nm = mpos.sum(1, keepdim=True)
            bad = (nm.view(-1) != (mids != 0).sum(1)).any() | ((mids != 0) != (torch.arange(J, device=dev).view(1, J) < nm)).any()
            if getattr(self, '_text_bad', None) is None:
                self._text_bad = torch.zeros((), dtype=torch.int32, device=dev)
            self._text_bad += bad.to(torch.int32)
            order = torch.argsort((~used).to(torch.int8), stable=True)
            used_s = used[order]
            sel = rows_all[order].contiguous()                                   # forward: any valid row index (weight 0 where unused)
            sel_bwd = torch.where(used_s, sel, torch.full_like(sel, B * TT))     # backward: unused slots -> the dummy row
            tgt = mids.view(-1)[order].contiguous()
            n = B * J
            row_w = (used_s.to(torch.float32) / used.sum().clamp(min=1).to(torch.float32)).contiguous()     # 1 / #masked tokens: the mean
        hrows = ops.cast_bf16(text_out.index_select(0, sel).contiguous())
        c = 'module.cls.predictions'
        zt = torch.empty(n, 768, device=dev, dtype=torch.bfloat16)
        gt = ops.gemm_ex(hrows, self.wb('cls.t'), bias=self.vec(c + '.transform.dense.bias'), act=L.ACT_GELU_ERF, zout=zt,
                         out_dtype=torch.float32)
        h2b, _ = ops.layernorm(gt, self.vec(c + '.transform.LayerNorm.weight'), self.vec(c + '.transform.LayerNorm.bias'), 1e-12)
        cbias = torch.zeros(L.VOCAB_PAD, device=dev)
        cbias[:L.VOCAB] = self.vec(c + '.bias')
        logits = ops.gemm_bias_act(h2b, self.wb('cls.dec'), cbias, out_dtype=torch.float32)
        dlog = torch.empty(n, L.VOCAB_PAD, device=dev, dtype=torch.bfloat16)
        check(lib.vitcap_ls_kl_loss(_p(logits), L.VOCAB_PAD, L.VOCAB, _p(tgt), 0.0 if scst else self.eps_ls, n, _p(row_w),
                                    _p(self.loss_buf), _p(dlog), L.VOCAB_PAD, _s()), 'ls_kl')
        # ================= backward: LM head
        wkey = W.TIED_SRC if self.model.tie_weights else c + '.decoder.weight'
        bsum = torch.zeros(L.VOCAB_PAD, device=dev)
        dlT = ops.transpose_colsum(dlog, bsum)                                 # [VOCAB_PAD][n_pad]
        self.g(c + '.bias').copy_(bsum[:L.VOCAB])
        h2T = ops.transpose_colsum(h2b)
        gdec = torch.empty(L.VOCAB_PAD, 768, device=dev)
        ops.gemm_ex(dlT, h2T, out=gdec)
        self.g(wkey).copy_(gdec[:L.VOCAB])                                     # first writer of the (tied) embedding gradient
        dh2 = ops.gemm_ex(dlog, self.wt('cls.dec'))                            # [n,768] bf16
        dgt, _ = self._ln_bwd(gt, dh2, c + '.transform.LayerNorm.weight', c + '.transform.LayerNorm.bias', 1e-12)
        dzt = torch.empty(n, 768, device=dev, dtype=torch.bfloat16)
        check(lib.vitcap_gelu_bwd(_p(dgt), _p(zt), _p(dzt), n * 768, _s()), 'gelu_bwd')
        self._wgrad_tn(dzt, hrows, self.g(c + '.transform.dense.weight'), self.g(c + '.transform.dense.bias').view(-1))
        dh = ops.gemm_ex(dzt, self.wt('cls.t'), out_dtype=torch.float32)      # [n,768] fp32
        yield 'cls'
        dtext = torch.zeros(B * TT + 1, 768, device=dev)            # + the dummy row unused loss slots write to
        dtext.index_copy_(0, sel_bwd, dh)
        dtext = dtext[:B * TT]
        if prune:
            dy = dtext                               # the last layer's backward runs on the text rows (see the forward)
        else:
            dy = torch.zeros(B, LR, 768, device=dev)
            dy[:, SV:] = dtext.view(B, TT, 768)
            dy = dy.view(Md, 768)
        # ================= backward: decoder layers
        for l in (3, 2, 1, 0):
            pre = 'module.bert.decoder.layer.%d' % l
            if l == 3 and prune:
                xb, qkv, ctx, lse, t1, ab, af, z, it, t2, ctx_t = dsaved[l]
                dt2f, dt2b = self._ln_bwd(t2, dy, pre + '.output.LayerNorm.weight', pre + '.output.LayerNorm.bias', 1e-12)
                if ph > 0:      # the dense branch sees the masked gradient, the residual branch (dt2f) the whole one
                    dt2b = ops.cast_bf16(ops.hidden_dropout(dt2f, None, TT, SV, hseed(l, 2), ph))
                self._wgrad_tn(dt2b, it, self.g(pre + '.output.dense.weight'), self.g(pre + '.output.dense.bias').view(-1))
                dz = ops.gemm_ex(dt2b, self.wt(pre + '.o'), aux=z)
                self._wgrad_tn(dz, ab, self.g(pre + '.intermediate.dense.weight'), self.g(pre + '.intermediate.dense.bias').view(-1))
                da = ops.gemm_ex(dz, self.wt(pre + '.i'), residual=dt2f, out_dtype=torch.float32)
                dt1f, dt1b = self._ln_bwd(t1, da, pre + '.attention.output.LayerNorm.weight',
                                          pre + '.attention.output.LayerNorm.bias', 1e-12)
                if ph > 0:
                    dt1b = ops.cast_bf16(ops.hidden_dropout(dt1f, None, TT, SV, hseed(l, 1), ph))
                self._wgrad_tn(dt1b, ctx_t, self.g(pre + '.attention.output.dense.weight'),
                               self.g(pre + '.attention.output.dense.bias').view(-1))
                dctx_t = ops.gemm_ex(dt1b, self.wt(pre + '.ao'))
                dctx = torch.zeros(B, LR, 768, device=dev, dtype=torch.bfloat16)
                dctx[:, SV:] = dctx_t.view(B, TT, 768)
                dqkv = ops.attn_dense_bwd(qkv, ctx, dctx.view(Md, 768), lse, B, LR, p_drop=pd, drop_seed=dseed[l], causal_from=SV,
                                          mask_from=SV + T if scst else 0, q_range=(QLO, LR))
                self._wgrad_tn(dqkv, xb, self.qkv_w_grad(pre), self.qkv_bias_grad(pre))
                res = torch.zeros(B, LR, 768, device=dev)
                res[:, SV:] = dt1f.view(B, TT, 768)               # the residual path reaches the layer input on the text rows only
                dy = ops.gemm_ex(dqkv, self.wt(pre + '.qkv'), residual=res.view(Md, 768), out_dtype=torch.float32)
                continue
            xb, qkv, ctx, lse, t1, ab, af, z, it, t2 = dsaved[l]
            # bias gradients = column sums of dt2b / dz / dt1b, added by the kernels that write those operands
            if ph > 0:      # masked gradient for the dense branch: its bf16 copy + bias column sums from the masked values
                dt2f, _ = self._ln_bwd(t2, dy, pre + '.output.LayerNorm.weight', pre + '.output.LayerNorm.bias', 1e-12)
                dt2b = ops.cast_bf16_colsum(ops.hidden_dropout(dt2f, None, LR, 0, hseed(l, 2), ph), self.g(pre + '.output.dense.bias').view(-1))
            else:
                dt2f, dt2b = self._ln_bwd(t2, dy, pre + '.output.LayerNorm.weight', pre + '.output.LayerNorm.bias', 1e-12,
                                          dxb_colsum=self.g(pre + '.output.dense.bias').view(-1))
            self._wgrad_tn(dt2b, it, self.g(pre + '.output.dense.weight'), None)
            dz, pend = self._dgrad_gelu(dt2b, self.wt(pre + '.o'), z, self.g(pre + '.intermediate.dense.bias').view(-1))   # [Md,3072] bf16
            self._wgrad_tn(dz, ab, self.g(pre + '.intermediate.dense.weight'), pend)
            da = ops.gemm_ex(dz, self.wt(pre + '.i'), residual=dt2f, out_dtype=torch.float32)
            if ph > 0:
                dt1f, _ = self._ln_bwd(t1, da, pre + '.attention.output.LayerNorm.weight', pre + '.attention.output.LayerNorm.bias', 1e-12)
                dt1b = ops.cast_bf16_colsum(ops.hidden_dropout(dt1f, None, LR, 0, hseed(l, 1), ph),
                                            self.g(pre + '.attention.output.dense.bias').view(-1))
            else:
                dt1f, dt1b = self._ln_bwd(t1, da, pre + '.attention.output.LayerNorm.weight', pre + '.attention.output.LayerNorm.bias',
                                          1e-12, dxb_colsum=self.g(pre + '.attention.output.dense.bias').view(-1))
            self._wgrad_tn(dt1b, ctx, self.g(pre + '.attention.output.dense.weight'), None)
            dctx = ops.gemm_ex(dt1b, self.wt(pre + '.ao'))
            dqkv = ops.attn_dense_bwd(qkv, ctx, dctx, lse, B, LR, p_drop=pd, drop_seed=dseed[l], causal_from=SV,
                                      mask_from=SV + T if scst else 0)
            self._wgrad_tn(dqkv, xb, self.qkv_w_grad(pre), self.qkv_bias_grad(pre))
            dy = ops.gemm_ex(dqkv, self.wt(pre + '.qkv'), residual=dt1f, out_dtype=torch.float32)
            if l % 2 == 0:
                yield 'dec%d' % (l // 2)
        dyv = dy.view(B, LR, 768)
        # ================= backward: text embeddings
        dtext_in = dyv[:, SV:].reshape(B * TT, 768).contiguous()
        if ph > 0:
            ops.hidden_dropout(dtext_in, None, TT, SV, hseed(4, 3), ph, out=dtext_in)
        demb, _ = self._ln_bwd(pre_emb, dtext_in, e + '.LayerNorm.weight', e + '.LayerNorm.bias', 1e-12, dres=None)
        check(lib.vitcap_embed_bwd(_p(demb), _p(ids20.view(-1)), TT, _p(self.g(e + '.word_embeddings.weight')),
                                   _p(self.g(e + '.position_embeddings.weight')), _p(self.g(e + '.token_type_embeddings.weight')),
                                   B * TT, T if scst else 0, _s()), 'embed_bwd')
        yield 'emb'
        # ================= backward: encoder
        if KS > 1:      # the KS sequences of an image saw the same visual rows: their gradients add up
            dyv_vis = dyv[:, :SV].reshape(Be, KS, SV, 768).sum(1)
        else:
            dyv_vis = dyv[:, :SV]
        dhid = dyv_vis[:, 1:SV].reshape(M, 768).contiguous()
        if not prune:
            dtag = torch.zeros(Be, NV, 768, device=dev)
            dtag[:, 0] = dyv_vis[:, 0]
            dtag = dtag.view(M, 768)

        def block_bwd_cls(pre, dxo_c):
            """Backward of block_fwd_cls: dxo_c (B,768) = gradient of the block's CLS output; returns the gradient of the block's
            input on all rows (the CLS query's attention spreads it over every row through K and V)."""
            xin, h1, qkv, ao, lse, xmid_c, h2_c, z_c, g_c, ao_c = saved.pop(pre)
            dxb = ops.cast_bf16(dxo_c)
            self._wgrad_tn(dxb, g_c, self.g(pre + '.mlp.fc2.weight'), self.g(pre + '.mlp.fc2.bias').view(-1))
            dz = ops.gemm_ex(dxb, self.wt(pre + '.fc2'), aux=z_c)
            self._wgrad_tn(dz, h2_c, self.g(pre + '.mlp.fc1.weight'), self.g(pre + '.mlp.fc1.bias').view(-1))
            dh2 = ops.gemm_ex(dz, self.wt(pre + '.fc1'))
            dmf, dmb = self._ln_bwd(xmid_c, dh2, pre + '.norm2.weight', pre + '.norm2.bias', 1e-6, dres=dxo_c)
            self._wgrad_tn(dmb, ao_c, self.g(pre + '.attn.proj.weight'), self.g(pre + '.attn.proj.bias').view(-1))
            dao_c = ops.gemm_ex(dmb, self.wt(pre + '.proj'))
            dao = torch.zeros(Be, NV, 768, device=dev, dtype=torch.bfloat16)
            dao[:, 0] = dao_c
            dqkv = ops.attn_dense_bwd(qkv, ao, dao.view(M, 768), lse, Be, NV, q_range=(0, 1))
            self._wgrad_tn(dqkv, h1, self.g(pre + '.attn.qkv.weight'), self.g(pre + '.attn.qkv.bias').view(-1))
            dh1 = ops.gemm_ex(dqkv, self.wt(pre + '.qkv'))
            res = torch.zeros(Be, NV, 768, device=dev)
            res[:, 0] = dmf
            dif, _ = self._ln_bwd(xin, dh1, pre + '.norm1.weight', pre + '.norm1.bias', 1e-6, dres=res.view(M, 768))
            return dif

        def block_bwd(pre, dxo):
            xin, h1, qkv, ao, lse, xmid, h2, z, g = saved.pop(pre)
            # the three bias gradients below are column sums of a backward operand; each is added by the kernel that WRITES that
            # operand (cast, GEMM epilogue, LayerNorm backward) instead of a separate pass over it
            if _FUSE_BIAS:
                dxb = ops.cast_bf16_colsum(dxo, self.g(pre + '.mlp.fc2.bias').view(-1))
            else:
                dxb = ops.cast_bf16(dxo)
                ops.colsum_bf16(dxb, self.g(pre + '.mlp.fc2.bias').view(-1))
            self._wgrad_tn(dxb, g, self.g(pre + '.mlp.fc2.weight'), None)
            dz, pend = self._dgrad_gelu(dxb, self.wt(pre + '.fc2'), z, self.g(pre + '.mlp.fc1.bias').view(-1))
            self._wgrad_tn(dz, h2, self.g(pre + '.mlp.fc1.weight'), pend)
            dh2 = ops.gemm_ex(dz, self.wt(pre + '.fc1'))
            dmf, dmb = self._ln_bwd(xmid, dh2, pre + '.norm2.weight', pre + '.norm2.bias', 1e-6, dres=dxo,
                                    dxb_colsum=self.g(pre + '.attn.proj.bias').view(-1))
            self._wgrad_tn(dmb, ao, self.g(pre + '.attn.proj.weight'), None)
            dao = ops.gemm_ex(dmb, self.wt(pre + '.proj'))
            dqkv = ops.attn_dense_bwd(qkv, ao, dao, lse, Be, NV)
            self._wgrad_tn(dqkv, h1, self.g(pre + '.attn.qkv.weight'), self.g(pre + '.attn.qkv.bias').view(-1))
            dh1 = ops.gemm_ex(dqkv, self.wt(pre + '.qkv'))
            dif, _ = self._ln_bwd(xin, dh1, pre + '.norm1.weight', pre + '.norm1.bias', 1e-6, dres=dmf)
            return dif

        for i in (3, 2, 1, 0):
            if i == 3 and prune:
                dtag = block_bwd_cls('module.bert.encoder.tag_blocks.3', dyv_vis[:, 0].contiguous())
            else:
                dtag = block_bwd('module.bert.encoder.tag_blocks.%d' % i, dtag)
            if i % 2 == 0:
                yield 'tag%d' % (i // 2)
        for i in (11, 10, 9, 8):
            dhid = block_bwd('module.bert.encoder.blocks.%d' % i, dhid)
            if i % 2 == 0:
                yield 'blk%d' % (i // 2)
        dxe = dhid
        ops.reduce_slabs(dtag.view(1, M * 768), dxe.view(-1), accumulate=True)      # fork point: the two branches' gradients meet
        for i in range(7, -1, -1):
            dxe = block_bwd('module.bert.encoder.blocks.%d' % i, dxe)
            if i % 2 == 0:
                yield 'blk%d' % (i // 2)
        # ================= backward: patch embed, cls token, position embedding
        check(lib.vitcap_sum_over_batch(_p(dxe), NV * 768, Be, _p(self.g(ie + 'pos_embed')), NV * 768, _s()), 'sum_over_batch')
        self.g(ie + 'cls_token').view(-1).copy_(self.g(ie + 'pos_embed').view(NV, 768)[0])
        dpatch = ops.cast_bf16(dxe.view(Be, NV, 768)[:, 1:].reshape(Be * 576, 768).contiguous())
        self._wgrad_tn(dpatch, patches, self.g(ie + 'patch_embed.proj.weight').view(768, 768), self.g(ie + 'patch_embed.proj.bias').view(-1))
        yield 'patch'
        return self.loss_buf[0], self.loss_buf[1]

    # ------------------------------------------------------------------ optimizer
    def all_reduce_grads(self):
        """Joins the bucketed all-reduces that forward_backward launched behind the backward pass (mean over ranks)."""
        self.reducer.finish()

    def optimizer_step(self):
        self.step_no += 1
        self.gsumsq.zero_()
        check(lib.vitcap_sumsq(_p(self.G), self.nflat, _p(self.gsumsq), _s()), 'sumsq')
        self._watch_nan()
        check(lib.vitcap_adamw_multi(_p(self.P), _p(self.G), _p(self.M), _p(self.V), _p(self.chunk_lr), _p(self.chunk_wd),
                                     _p(self.gsumsq), self.clip, self.lr_scale, self.step_no, 0.9, 0.999, 1e-8,
                                     self.nflat // CH, _s()), 'adamw')
        self.lr_scale = max(0.0, float(self.max_iter - self.step_no) / float(max(1.0, self.max_iter)))   # WarmupLinearSchedule
        self.refresh_weights()

    def train_step(self, batch):
        if self.use_graphs and self._graphable(batch):
            return self.train_step_graph(batch)
        loss, tag_loss = self.forward_backward(batch)
        self.all_reduce_grads()
        self.optimizer_step()
        if self.step_no % 50 == 0:
            self.flush_nan_check()
        return {'masked_loss': loss, 'tag_loss': tag_loss}

    # ------------------------------------------------------------------ graph mode: the step as a few host calls
    # The eager step issues ~600 ctypes launches + ~150 torch calls from Python (with 8 ranks x (loader processes +
    # launcher) on one host that is the fragile part).  Here the whole forward + backward is recorded ONCE per batch shape into hipGraph
    # segments (torch.cuda.graph: the stream capture also records torch's own small kernels and keeps the step's activations in a
    # private pool) and a step becomes: copy the batch into the static input buffers (5 calls), rewrite the dropout salt (1), replay
    # the segments (1 without a gradient exchange; 4 with one, the buckets of a segment's stages start behind it), clip + AdamW +
    # weight refresh (4).  Same kernels, same arguments, same order as the eager step: bit-identical parameters with dropout off
    # (tests/test_hip_train_e2e.py::test_graph_step_equals_eager); with dropout on the keep decisions of step t are those of
    # seed(step 0) ^ salt(t) -- another stream than the eager step's, equally distributed.
    GRAPH_SEGMENT_ENDS = ('dec0', 'tag0', 'blk3', 'patch')       # with a gradient exchange: 4 segments (cls+dec | emb+tag | blk5-3 | blk2-0+patch)
    _GRAPH_KEYS = ('image', 'input_ids', 'attention_mask', 'masked_pos', 'masked_ids', 'label')

    def _graphable(self, batch):
        return ('sample_ids' not in batch and int(batch.get('seq_per_image', 1)) == 1 and 'masked_pos' in batch and 'masked_ids' in batch
                and all(torch.is_tensor(batch[k]) for k in self._GRAPH_KEYS if k in batch))

    def _graph_build(self, key, batch):
        dev = self.dev
        static = {k: batch[k].to(dev).clone().contiguous() for k in self._GRAPH_KEYS if k in batch}
        salt = torch.zeros(1, dtype=torch.int32, device=dev)
        # one eager pass on the static buffers first: every kernel's launch attributes exist and the caching allocator is warm.  It runs
        # WITHOUT the gradient exchange: a rank that meets a new batch shape -- a ragged last batch, or a batch that stays
        # eager on another rank -- must not issue one more round of collectives than its peers (RCCL would hang); the pass only
        # changes gradients and the loss buffer, both rewritten by the step that follows
        exchange = self.reducer.exchange
        self.reducer.exchange = False
        try:
            self.forward_backward(static)
            self.all_reduce_grads()
        finally:
            self.reducer.exchange = exchange
        torch.cuda.synchronize(dev)
        segs = []
        pool = torch.cuda.graph_pool_handle()
        gen = self._fb_gen(static, capture=True)
        done = False
        last = GRAD_STAGES[-1]
        check(lib.vitcap_set_dropout_salt(C.c_void_p(salt.data_ptr())), 'set_dropout_salt')
        # The grid of a persistent GEMM is fixed when its launch is RECORDED (vitcap_gemm_reserve_cus is read by the launcher, on the
        # host): a replayed segment keeps whatever reservation was set while it was captured.  The first bucket's collective starts
        # behind segment 0, so every later segment is captured with the reducer's CUs left free (captured after finish()
        # had set the reservation back to 0, the graphs froze all-CU grids and the reservation was a no-op in graph mode).
        reserve = self.reducer.reserve_cus if exchange else 0
        try:
            while not done:
                g = torch.cuda.CUDAGraph()
                stages = []
                if reserve and segs:
                    lib.vitcap_gemm_reserve_cus(reserve)
                with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):     # a loader thread may copy meanwhile
                    while True:
                        try:
                            st = next(gen)
                        except StopIteration:
                            done = True
                            break
                        stages.append(st)
                        # the last stage is followed by nothing but the generator's return: it stays in this segment
                        if exchange and st in self.GRAPH_SEGMENT_ENDS and st != last:
                            break
                segs.append((g, stages))
        finally:
            if reserve:
                lib.vitcap_gemm_reserve_cus(0)
            check(lib.vitcap_set_dropout_salt(None), 'set_dropout_salt')
        entry = {'static': static, 'salt': salt, 'segs': segs, 'reserved_cus': reserve}
        self._graphs[key] = entry
        return entry

    def train_step_graph(self, batch):
        key = tuple((k, tuple(batch[k].shape), str(batch[k].dtype)) for k in self._GRAPH_KEYS if k in batch)
        entry = self._graphs.get(key) or self._graph_build(key, batch)
        for k, t in entry['static'].items():
            t.copy_(batch[k], non_blocking=True)
        if self.attn_dropout > 0 or self.hidden_dropout > 0:
            entry['salt'].fill_(mix32(0x85ebca6b, self.step_no) & 0x7fffffff)
        self.reducer.begin()
        for g, stages in entry['segs']:
            g.replay()
            for st in stages:
                self.reducer.stage_done(st)
        self.all_reduce_grads()
        self.optimizer_step()
        if self.step_no % 50 == 0 or not entry.get('checked'):     # the first replay of a shape as well: a bad batch layout is reported at once
            entry['checked'] = True
            self.flush_text_check()
        return {'masked_loss': self.loss_buf[0], 'tag_loss': self.loss_buf[1]}

    def grad_norm(self):
        return float(self.gsumsq.sqrt())
