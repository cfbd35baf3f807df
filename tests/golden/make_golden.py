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


def ref_generate(model, enc, image, num_beams=1):
    """Notebook cell 15/16 flow == ImageCaptioning.forward test branch (..._bertemb.py:87-184)."""
    sys.path.insert(0, os.path.join(REPO))
    from oracle import vitcap_oracle as O
    B = image.shape[0]
    input_ids, am = O.test_text_inputs(B)
    img_feats = enc(image)
    full = O.construct_attn_mask(am, img_feats.shape[1])      # restated 30-line mask (pipeline needs cv2)
    kw = dict(is_decode=True, do_sample=False, bos_token_id=101, pad_token_id=0, eos_token_ids=[102],
              mask_token_id=103, add_od_labels=True, od_labels_start_posid=20, max_length=20,
              num_beams=num_beams, temperature=1, top_k=0, top_p=1, repetition_penalty=1,
              length_penalty=1, num_return_sequences=1, num_keep_best=1)
    with torch.no_grad():
        return model(img_feats=img_feats, input_ids=input_ids, attention_mask=full,
                     masked_pos=torch.ones(B, 70, dtype=torch.int32),
                     token_type_ids=torch.zeros(B, 70, dtype=torch.long),
                     label=torch.zeros(B, 30522), gen_tag_ratio=1, **kw)


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

    # ---- end-to-end greedy (pipeline flow: tagemb cls, tied)
    for B in (1, 2):
        ids, lp = ref_generate(model, enc, img[:B])
        out['greedy_b%d_ids' % B] = ids.numpy().copy()
        out['greedy_b%d_logprobs' % B] = lp.numpy().copy()
        print('greedy B=%d' % B, ids.tolist(), lp.tolist())
    # per-step margins / logits rows from the oracle's as-written path are checked against these ids;
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

    # ---- beam=2 (small) for the beam driver
    ids, lp = ref_generate(model, enc, img[:1], num_beams=2)
    out['beam2_b1_ids'] = ids.numpy().copy()
    out['beam2_b1_logprobs'] = lp.numpy().copy()
    print('beam2', ids.tolist(), lp.tolist())

    # ---- notebook flow: tagemb None, untied
    sd2 = W.make_state_dict(seed=0, tie_weights=False)
    model2, enc2 = build_reference(None, False)
    load_recipe(model2, enc2, sd2)
    ids, lp = ref_generate(model2, enc2, img[:1])
    out['greedy_untied_nocls_b1_ids'] = ids.numpy().copy()
    out['greedy_untied_nocls_b1_logprobs'] = lp.numpy().copy()
    print('untied', ids.tolist(), lp.tolist())

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    with open(os.path.join(HERE, 'reference_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('wrote goldens')


if __name__ == '__main__':
    main()
