"""Generate golden vectors by running the REFERENCE itself (container only).

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz, *.json)

This script imports /root/reference (read-only) with the throw-away shims listed in SURVEY.md
section 8c, loads the seeded weights of ``vitcap_amd.weights`` into the reference's own
``ViTCAP`` / timm modules, runs them on CPU fp32 and stores small input/output vectors.
Nothing from the reference is copied into the repo -- the fixtures are data (inputs + expected
outputs); ``/root/reference`` does not exist on the GPU box and no test reads it.

Each fixture records ``torch.__version__`` because all dense arithmetic is ATen's.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)


def install_shims():
    # torch._six (timm/models/layers/helpers.py:6)
    import collections.abc
    six = types.ModuleType('torch._six')
    six.container_abcs = collections.abc
    six.int_classes = int
    six.string_classes = str
    sys.modules['torch._six'] = six
    torch._six = six
    # boto3 / botocore (file_utils.py:19-21)
    b3 = types.ModuleType('boto3')
    bc = types.ModuleType('botocore')
    bce = types.ModuleType('botocore.exceptions')
    bce.ClientError = type('ClientError', (Exception,), {})
    bc.exceptions = bce
    sys.modules.update({'boto3': b3, 'botocore': bc, 'botocore.exceptions': bce})
    # pip-timm facade: package path -> vendored tree, plus timm.data constants
    # (vision_transformer.py:34, hub.py:14)
    vend = os.path.join(REF, 'src', 'pytorch_image_models', 'timm')
    tm = types.ModuleType('timm')
    tm.__path__ = [vend]
    tm.__version__ = '0.4.1'
    sys.modules['timm'] = tm
    td = types.ModuleType('timm.data')
    td.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    td.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    td.IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
    td.IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
    td.IMAGENET_DPN_MEAN = tuple(x / 255 for x in (124, 117, 104))
    td.IMAGENET_DPN_STD = tuple(1 / (.0167 * 255) for _ in range(3))
    sys.modules['timm.data'] = td
    tm.data = td
    # modeling_bert.py:1415,1496 call .cuda() on CPU-created tensors
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


class _StubModule(types.ModuleType):
    """A module whose every attribute is an inert class: stands in for packages the reference's pipeline module imports at the
    top of the file and never touches on the path driven here (image transforms, tokenisers for tags, progress bars)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        v = type(name, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})
        setattr(self, name, v)
        return v


def install_pipeline_shims():
    """On top of install_shims(): whatever ``src.pipelines.tagger_caption_uni_pipeline_expanding_bertemb`` needs to IMPORT in
    this container, so that the reference's own ``ImageCaptioning`` wrapper (construct_attn_mask :57-85, forward :87-184), its
    ``InputAsDict`` and its ``CaptionTensorizer`` run as written.  Stubbed (absent here, unused on this path): torchvision, cv2,
    nltk, future, progressbar, anytree, pathos, deprecated, pycocotools; torch.utils.model_zoo lost the private names
    checkpoint.py:11 imports."""
    install_shims()
    for name in ('torchvision', 'torchvision.transforms', 'torchvision.transforms.transforms', 'torchvision.transforms.functional',
                 'torchvision.datasets', 'torchvision.datasets.folder', 'cv2', 'nltk', 'future', 'future.utils', 'progressbar',
                 'anytree', 'pathos', 'pathos.multiprocessing', 'deprecated', 'pycocotools', 'pycocotools.coco',
                 'pycocotools.cocoeval'):
        parts = name.split('.')
        for i in range(1, len(parts) + 1):
            n = '.'.join(parts[:i])
            if n not in sys.modules:
                m = _StubModule(n)
                m.__path__ = []
                sys.modules[n] = m
                if i > 1:
                    setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
    import torch.utils.model_zoo as mz
    for n in ('_download_url_to_file', 'urlparse', 'HASH_REGEX'):
        if not hasattr(mz, n):
            setattr(mz, n, None)


def build_wrapper(model, enc, num_beams=1, **over):
    """The reference's own inference model as CaptionUniPipeline.get_raw_model(is_train=False) assembles it
    (..._bertemb.py:566-618, image encoder :750-778): ImageCaptioning(ViTCAP, test_extra_input, InputAsDict(timm ViT))."""
    from src.pipelines.tagger_caption_uni_pipeline_expanding_bertemb import ImageCaptioning
    from src.tools.torch_common import InputAsDict
    from src.layers.bert.tokenization_bert import BertTokenizer
    tok = BertTokenizer(os.path.join(REF, 'yaml', 'VILT-L12-H784-uncased_16_384', 'vocab.txt'), do_lower_case=True)
    cls_id, sep_id, pad_id, mask_id = tok.convert_tokens_to_ids([tok.cls_token, tok.sep_token, tok.pad_token, tok.mask_token])
    cfg = types.SimpleNamespace(mask_type='seq2seq', use_cbs=False, pert_img_prob=None, category='bert', gt_tag_train=False,
                                pred_tag_train=False, gen_tag_ratio=None, max_iter=1)
    extra = {'is_decode': True, 'do_sample': False, 'bos_token_id': cls_id, 'pad_token_id': pad_id, 'eos_token_ids': [sep_id],
             'mask_token_id': mask_id, 'add_od_labels': True, 'od_labels_start_posid': 20, 'max_length': 20,
             'num_beams': num_beams, 'temperature': 1, 'top_k': 0, 'top_p': 1, 'repetition_penalty': 1, 'length_penalty': 1,
             'num_return_sequences': 1, 'num_keep_best': 1}
    extra.update(over)
    wrap = ImageCaptioning(model, extra, tokenizer=tok, bert_tokenizer=tok, image_encoder=InputAsDict(enc), cfg=cfg)
    return wrap.eval(), tok          # the pipeline evaluates under .eval() (uni_pipeline.py:759-769)


def reference_tensorizer(tok, is_train, **kw):
    """The reference's CaptionTensorizer (src/data_layer/dataset.py:158-417) with the shipped YAML's lengths (max_seq_a_length 20 =
    max_gen_length at test time, od label span 50: ..._bertemb.py:423-470)."""
    from src.data_layer.dataset import CaptionTensorizer
    return CaptionTensorizer(tok, max_img_seq_length=0, max_seq_length=70, max_seq_a_length=20, is_train=is_train,
                             mask_type='seq2seq', **kw)


def build_reference(tagemb='cls', tie_weights=True):
    from src.layers.bert import BertConfig, ViTCAP
    from src.pytorch_image_models.timm.models import vision_transformer as vt
    from src.pytorch_image_models import timm as vtimm
    vt.load_pretrained = lambda *a, **k: None          # modeling_bert.py:449,456 hard-code pretrained=True
    cfg = BertConfig.from_pretrained(os.path.join(REF, 'yaml', 'VILT-L12-H784-uncased_16_384'),
                                     num_labels=2, finetuning_task='image_captioning')
    # ..._bertemb.py:520-564 get_fusion_config with the shipped YAML + pipeline defaults
    cfg.img_feature_type = 'frcnn'
    cfg.hidden_dropout_prob = 0
    cfg.loss_type = 'classification'
    cfg.tie_weights = tie_weights
    cfg.freeze_embedding = False
    cfg.label_smoothing = 0.1
    cfg.drop_worst_ratio = 0
    cfg.drop_worst_after = 0
    cfg.img_feature_dim = 2054
    cfg.use_img_layernorm = False
    cfg.img_layer_norm_eps = 1e-12
    cfg.net = 'vit_base_patch16_384'
    cfg.ignore_project_image = True
    cfg.later_captioning = None
    cfg.attn_token_sample = None
    cfg.vocab = None
    cfg.tokenizer = None
    cfg.loss = 'focal'
    cfg.split_blocks = 4
    cfg.topktagger = None
    cfg.tagemb = tagemb
    cfg.tagemb_gradient = None
    cfg.category = 'bert'
    cfg.tie_tag_weights = False
    cfg.topk = 50
    model = ViTCAP(cfg).eval()
    enc = vtimm.create_model('vit_base_patch16_384', output_grid=True, pretrained=False)
    enc.norm = torch.nn.Identity()
    enc.blocks = torch.nn.ModuleList()
    enc.eval()
    return model, enc


def load_recipe(model, enc, sd_np):
    msd = model.state_dict()
    esd = enc.state_dict()
    miss = []
    with torch.no_grad():
        for k, v in sd_np.items():
            t = torch.from_numpy(v)
            if k.startswith('module.'):
                kk = k[len('module.'):]
                if kk in msd:
                    assert tuple(msd[kk].shape) == tuple(t.shape), (k, msd[kk].shape, t.shape)
                    msd[kk].copy_(t)
                else:
                    miss.append(k)
            else:
                kk = k[len('image_encoder.module.'):]
                assert tuple(esd[kk].shape) == tuple(t.shape), (k, esd[kk].shape, t.shape)
                esd[kk].copy_(t)
    unexpected = [k for k in msd if 'module.' + k not in sd_np]
    return miss, unexpected


def digest(t, k=8):
    t = t.detach().float()
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, k).long()
    return {'shape': list(t.shape), 'mean': float(f.double().mean()), 'std': float(f.double().std()),
            'absmax': float(f.abs().max()), 'sample_idx': idx.tolist(), 'sample': f[idx].tolist()}


class _MarginRecorder(object):
    """Records, while the reference's own generate loop runs, how well conditioned each discrete decision is:

    * greedy (modeling_utils.py:846 ``torch.argmax(next_token_logits)``): top-1 minus top-2 logit per sequence and step;
    * beam (modeling_utils.py:996 ``torch.topk(_scores, 2*num_beams)``): the smallest gap between neighbours among the
      2*num_beams+1 best candidate scores per image and step (any swap among them can change the beam contents).

    The device tests demand token-exact agreement with the reference wherever these margins exceed the bf16 noise floor."""

    def __init__(self):
        self.greedy, self.beam = [], []

    def __enter__(self):
        self._argmax, self._topk = torch.argmax, torch.topk
        rec = self

        def argmax(x, *a, **k):
            if x.dim() == 2 and x.shape[-1] == 30522:
                t2 = rec._topk(x, 2, dim=-1).values
                rec.greedy.append((t2[:, 0] - t2[:, 1]).clone())
            return rec._argmax(x, *a, **k)

        def topk(x, k, *a, **kw):
            if x.dim() == 2 and x.shape[-1] % 30522 == 0 and x.shape[-1] > 30522 and k == 2 * (x.shape[-1] // 30522):
                v = rec._topk(x, k + 1, dim=1).values
                rec.beam.append((v[:, :-1] - v[:, 1:]).min(dim=1).values.clone())
            return rec._topk(x, k, *a, **kw)

        torch.argmax, torch.topk = argmax, topk
        return self

    def __exit__(self, *exc):
        torch.argmax, torch.topk = self._argmax, self._topk

    def margins(self, steps=19):
        rows = self.greedy or self.beam
        m = torch.stack(rows, 1)                                   # (B, steps run)
        if m.shape[1] < steps:                                     # the reference stopped early: every sequence had finished
            m = torch.cat([m, torch.full((m.shape[0], steps - m.shape[1]), float('inf'))], 1)
        return m.numpy().copy()


def ref_generate(model, enc, image, num_beams=1, n_tag_visible=0, **over):
    """Notebook cell 15/16 flow == ImageCaptioning.forward test branch (..._bertemb.py:87-184).
    Returns (ids, logprobs, margins (B,19))."""
    sys.path.insert(0, os.path.join(REPO))
    from oracle import vitcap_oracle as O
    B = image.shape[0]
    input_ids, am = O.test_text_inputs(B, n_tag_visible=n_tag_visible)
    img_feats = enc(image)
    full = O.construct_attn_mask(am, img_feats.shape[1])      # restated 30-line mask (pipeline needs cv2)
    kw = dict(is_decode=True, do_sample=False, bos_token_id=101, pad_token_id=0, eos_token_ids=[102],
              mask_token_id=103, add_od_labels=True, od_labels_start_posid=20, max_length=20,
              num_beams=num_beams, temperature=1, top_k=0, top_p=1, repetition_penalty=1,
              length_penalty=1, num_return_sequences=1, num_keep_best=1)
    kw.update(over)
    with torch.no_grad(), _MarginRecorder() as rec:
        ids, lp = model(img_feats=img_feats, input_ids=input_ids, attention_mask=full,
                        masked_pos=torch.ones(B, 70, dtype=torch.int32),
                        token_type_ids=torch.zeros(B, 70, dtype=torch.long),
                        label=torch.zeros(B, 30522), gen_tag_ratio=1, **kw)
    m = rec.margins()
    if num_beams > 1 and m.shape[0] != B:
        m = m.reshape(B, -1)
    return ids, lp, m


def main():
    install_shims()
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = {'torch': torch.__version__, 'seed': 0}
    out = {}

    sd_np = W.make_state_dict(seed=0, tie_weights=True)
    model, enc = build_reference('cls', True)
    miss, unexp = load_recipe(model, enc, sd_np)
    meta['keys_missing_in_reference'] = miss
    meta['reference_keys_not_in_recipe'] = unexp
    meta['n_keys'] = len(sd_np)
    meta['key_shapes'] = {k: list(v.shape) for k, v in sd_np.items()}
    meta['digests'] = {k: W.tensor_digest(v) for k, v in list(sd_np.items())[:12]}
    print('missing', miss, 'unexpected', unexp)

    img = torch.from_numpy(W.gen_image_batch(2, 1234))
    with torch.no_grad():
        # ---- a1
        img_feats = enc(img)
        out['a1_img_feats_b0'] = img_feats[0, :4].numpy().copy()
        meta['a1'] = digest(img_feats)
        # ---- a4 one block, a5 split encoder
        blk0 = model.bert.encoder.blocks[0](img_feats, torch.zeros(2, 1, 577, 577))
        meta['a4_block0'] = digest(blk0)
        out['a4_block0_rows'] = blk0[0, :3].numpy().copy()
        hid, tag_hid = model.bert.encoder(img_feats, torch.zeros(2, 1, 577, 577), head_mask=[None] * 4)
        meta['a5_hidden'] = digest(hid)
        meta['a5_tag_hidden'] = digest(tag_hid)
        out['a5_hidden_rows'] = hid[:, :2].numpy().copy()
        out['a5_tag_hidden_cls'] = tag_hid[:, 0].numpy().copy()
        # ---- a6 tag head
        logit = model.bert.tag_logit(model.bert.pooler(tag_hid))
        prob, pred = torch.sigmoid(logit).topk(50, dim=1)
        meta['a6_logit'] = digest(logit)
        out['a6_logit_head'] = logit[:, :64].numpy().copy()
        out['a6_pred_topk'] = pred.numpy().copy()
        out['a6_prob_topk'] = prob.numpy().copy()
        out['a6_topk_len'] = (prob >= 0.2).sum(1).numpy().copy()
        # ---- a9 one BertLayer at S=630 with a step-1 style mask, a10 head
        g = torch.Generator().manual_seed(7)
        xs = torch.randn(1, 630, 768, generator=g) * 0.5
        m = torch.ones(1, 630, 630)
        m[:, :52, :52] = 0
        m[:, :2, :2] = torch.tril(torch.ones(2, 2))
        m[:, 52:, :52] = 0
        ext = (1.0 - m.unsqueeze(1)) * -10000.0
        y = model.bert.decoder.layer[0](xs, ext, None, None)[0]
        out['a9_in_seed'] = np.array([7])
        out['a9_rows'] = y[0, [0, 1, 2, 52, 629]].numpy().copy()
        meta['a9'] = digest(y)
        z = model.cls(xs[:, :3])
        out['a10_logits_head'] = z[0, :, :128].numpy().copy()
        meta['a10'] = digest(z)

    # ---- end-to-end greedy (pipeline flow: tagemb cls, tied); per-step top-2 margins from the reference's own logits
    img4 = torch.from_numpy(W.gen_image_batch(4, 1234))
    for B in (1, 2, 4):
        ids, lp, m = ref_generate(model, enc, img4[:B])
        out['greedy_b%d_ids' % B] = ids.numpy().copy()
        out['greedy_b%d_logprobs' % B] = lp.numpy().copy()
        out['greedy_b%d_margins' % B] = m
        print('greedy B=%d' % B, ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    # ---- the same with another token as eos_token_ids (a generate() kwarg, modeling_bert.py:928-933): sequences stop at
    # data-dependent lengths, rows that finished emit PAD, the score counts the EOS step (modeling_utils.py:855-877) and the
    # loop leaves early once every sequence has finished (:866).  The token is the one whose first occurrence in the four
    # captions above is most spread out.
    cap = out['greedy_b4_ids'][:, 0]

    def first_pos(tok):
        return tuple(int(np.argmax(r == tok)) if (r == tok).any() else 99 for r in cap)
    cands = [t for t in np.unique(cap[:, 2:19]) if t not in (0, 101, 102)]
    alt_eos = int(max(cands, key=lambda t: (len(set(first_pos(t))), -min(first_pos(t)))))
    ids, lp, m = ref_generate(model, enc, img4, eos_token_ids=[alt_eos])
    out['alt_eos_id'] = np.array([alt_eos])
    out['greedy_alteos_b4_ids'] = ids.numpy().copy()
    out['greedy_alteos_b4_logprobs'] = lp.numpy().copy()
    out['greedy_alteos_b4_margins'] = m
    print('greedy alt eos %d' % alt_eos, ids.tolist(), lp.tolist())
    # ---- well-conditioned inputs: random-init logits are nearly flat, so most 19-step captions contain at least one
    # decision whose margin is below what bf16 arithmetic can resolve.  From 16 candidate images (seed 4321) the oracle's
    # fp32 incremental path (token-exact with the reference, tests/test_oracle_golden.py) picks the 4 whose smallest
    # margin is largest; the REFERENCE is then run on those 4.  The device tests demand whole-caption equality on them.
    cand = torch.from_numpy(W.gen_image_batch(16, 4321))
    sd_t = O.to_torch(sd_np)
    with torch.no_grad():
        _, _, tr = O.greedy_incremental(sd_t, cand, emulate_bf16=False, return_trace=True)
    cm = torch.stack([st['margin'] for st in tr['steps']], 1).min(1).values
    sel = torch.argsort(cm, descending=True)[:4].sort().values
    print('candidate min margins', [round(float(x), 4) for x in cm], 'selected', sel.tolist())
    ids, lp, m = ref_generate(model, enc, cand[sel])
    out['sel_image_seed'] = np.array([4321])
    out['sel_index'] = sel.numpy().copy()
    out['greedy_sel_ids'] = ids.numpy().copy()
    out['greedy_sel_logprobs'] = lp.numpy().copy()
    out['greedy_sel_margins'] = m
    print('greedy selected', ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    # the alternative EOS on the well-conditioned images too: whole captions of different lengths are comparable
    cap = out['greedy_sel_ids'][:, 0]
    cands = [t for t in np.unique(cap[:, 2:19]) if t not in (0, 101, 102)]
    alt2 = int(max(cands, key=lambda t: (len(set(first_pos(t))), -min(first_pos(t)))))
    ids, lp, m = ref_generate(model, enc, cand[sel], eos_token_ids=[alt2])
    out['alt_eos_sel_id'] = np.array([alt2])
    out['greedy_alteos_sel_ids'] = ids.numpy().copy()
    out['greedy_alteos_sel_logprobs'] = lp.numpy().copy()
    out['greedy_alteos_sel_margins'] = m
    print('greedy selected, alt eos %d' % alt2, ids.tolist(), lp.tolist())
    # also store the reference's own step-1 logits row for a direct float comparison
    with torch.no_grad():
        input_ids, am = O.test_text_inputs(1)
        full = O.construct_attn_mask(am, 577)
        step_ids = torch.cat([torch.tensor([[101, 103]]), input_ids[:, 20:]], 1)
        mask = O._remove_rows_cols(full, 2, 20, 2, 20)
        pos = torch.cat([torch.arange(2), torch.arange(20, 70)]).unsqueeze(0)
        res = model.encode_forward(step_ids, enc(img[:1]), mask, position_ids=pos,
                                   token_type_ids=torch.zeros(1, 52, dtype=torch.long), is_training=False,
                                   label=torch.zeros(1, 30522), gen_tag_ratio=1)
        out['step1_logits_row'] = res[0][0, 1].numpy().copy()

    # ---- beam search: beam=2 B=1, beam=5 at B in {1, 2}, beam=3 with the alternative EOS (hypotheses of many
    # lengths), and beam=5 on the two best-conditioned of the 16 candidate images.  Decision margins come from the oracle's
    # restatement of the driver (oracle.beam_bookkeeping: scan boundary, BeamHypotheses.add / is_done comparisons, final
    # ordering) run on the fp32 incremental model -- it must reproduce the reference's output exactly here, or we abort.
    def beam_case(name, images, nb, over):
        ids, lp, _ = ref_generate(model, enc, images, num_beams=nb, **over)
        with torch.no_grad():
            ids_o, lp_o, mg = O.beam_incremental(sd_t, images, num_beams=nb, emulate_bf16=False, return_margins=True,
                                                 eos=over.get('eos_token_ids', [102])[0])
        assert torch.equal(ids_o, ids), (name, ids_o.tolist(), ids.tolist())
        assert torch.allclose(lp_o, lp, atol=2e-5), (name, lp_o, lp)
        out[name + '_ids'] = ids.numpy().copy()
        out[name + '_logprobs'] = lp.numpy().copy()
        out[name + '_margins'] = mg.numpy().copy()
        print(name, ids.tolist(), lp.tolist(), 'min decision gap', mg.min(1).values.tolist())

    beam_case('beam2_b1', img4[:1], 2, {})
    beam_case('beam5_b1', img4[:1], 5, {})
    beam_case('beam5_b2', img4[:2], 5, {})
    beam_case('beam3_alteos_b2', img4[:2], 3, {'eos_token_ids': [alt_eos]})
    with torch.no_grad():
        _, _, mg = O.beam_incremental(sd_t, cand, num_beams=5, emulate_bf16=False, return_margins=True)
    bm = mg.min(1).values
    bsel = torch.argsort(bm, descending=True)[:2].sort().values
    print('candidate beam-5 min gaps', [round(float(x), 4) for x in bm], 'selected', bsel.tolist())
    out['beam_sel_index'] = bsel.numpy().copy()
    beam_case('beam5_sel', cand[bsel], 5, {})

    # ---- the predicted tag tokens VISIBLE to the caption (the mask tensorize_ab builds when a text_b of n tag
    # tokens is attached): all 50 slots, and a partial set of 7; pipeline flow (tagemb 'cls': branch B for steps 1..18, branch A at
    # the last step with the recipe's topk_len = 50, modeling_bert.py:1435-1489)
    for n in (50, 7):
        ids, lp, m = ref_generate(model, enc, img4[:2], n_tag_visible=n)
        out['greedy_tags%d_b2_ids' % n] = ids.numpy().copy()
        out['greedy_tags%d_b2_logprobs' % n] = lp.numpy().copy()
        out['greedy_tags%d_b2_margins' % n] = m
        print('tags visible %d' % n, ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    ids, lp, m = ref_generate(model, enc, cand[sel], n_tag_visible=50)
    out['greedy_tags50_sel_ids'] = ids.numpy().copy()
    out['greedy_tags50_sel_logprobs'] = lp.numpy().copy()
    out['greedy_tags50_sel_margins'] = m
    print('tags visible 50, selected images', ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    with torch.no_grad():
        _, _, pred_l, tl = O.tag_head(sd_t, O.split_encoder(sd_t, O.patch_embed(sd_t, img4[:2]))[1])
    out['tags_topk_len_b2'] = tl.numpy().copy()

    # ---- notebook flow (BASELINE configs[0]): tagemb None, untied
    sd2 = W.make_state_dict(seed=0, tie_weights=False)
    model2, enc2 = build_reference(None, False)
    load_recipe(model2, enc2, sd2)
    for B in (1, 2):
        ids, lp, m = ref_generate(model2, enc2, img4[:B])
        out['greedy_untied_nocls_b%d_ids' % B] = ids.numpy().copy()
        out['greedy_untied_nocls_b%d_logprobs' % B] = lp.numpy().copy()
        out['greedy_untied_nocls_b%d_margins' % B] = m
        print('untied B=%d' % B, ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    ids, lp, m = ref_generate(model2, enc2, img4[:2], n_tag_visible=50)          # tagemb None: extra_embeddings / word embeddings
    out['greedy_untied_tags50_b2_ids'] = ids.numpy().copy()
    out['greedy_untied_tags50_b2_logprobs'] = lp.numpy().copy()
    out['greedy_untied_tags50_b2_margins'] = m
    print('untied, tags visible 50', ids.tolist(), lp.tolist(), 'min margin', m.min(1).tolist())
    meta['recipe_version'] = W.RECIPE_VERSION

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    with open(os.path.join(HERE, 'reference_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('wrote goldens')


if __name__ == '__main__':
    main()
