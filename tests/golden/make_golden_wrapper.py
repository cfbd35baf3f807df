"""a8 / a16 pinned against the reference's OWN wrapper code, and the population statistic (container only).

Run:  python tests/golden/make_golden_wrapper.py [masks|pin|pop|beam]...   (default: all four parts)
      writes tests/golden/reference_wrapper.npz (parts masks + pin) and tests/golden/reference_population.npz (pop + beam)

Every other generator of this directory hands the reference's ``ViTCAP`` the joint mask / input ids built by the ORACLE's restatement
(oracle.test_text_inputs, oracle.construct_attn_mask).  This one imports ``src.pipelines.tagger_caption_uni_pipeline_expanding_bertemb``
itself (make_golden.install_pipeline_shims stubs the packages it imports at the top and never uses here) and drives

  * the reference's ``CaptionTensorizer.tensorize_ab`` (dataset.py:206-417)      -> input_ids, the 70 x 70 mask, masked_pos / masked_ids
  * the reference's ``ImageCaptioning.construct_attn_mask`` (..._bertemb.py:57-85) -> the 647 x 647 joint mask
  * the reference's ``ImageCaptioning.forward`` (..._bertemb.py:87-184), test and train branches, with ``InputAsDict`` (torch_common.py:270-280)

masks  the joint masks as the reference builds them: test mode with 0 / 7 / 50 visible tag slots, train mode with a 13-token caption
       (bit-packed), next to the reference tensorizer's own input ids
pin    greedy_b4, greedy_tags50_b2 and the training golden again, THROUGH the wrapper's forward: must reproduce the stored goldens
       (reference_vectors.npz, reference_train.npz) bit for bit -- asserted here, the wrapper's outputs are stored as well
pop    the reference's greedy captions + decision margins on 32 images nobody selected (16 uniform-noise, seed 9001; 16 structured,
       seed 9002), through the wrapper
beam   beam = 5 on the first 4 images of either family, through the wrapper; decision margins from the oracle's driver restatement
       (must reproduce the reference's ids exactly, as in make_golden.py)
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402

THREADS = int(os.environ.get('GOLDEN_THREADS', '8'))
CAPTION13 = 'a man riding a brown horse down a city street near a bus'      # 13 word pieces in the shipped vocabulary


def packbits(t):
    a = np.asarray(t).astype(bool)
    return np.packbits(a.reshape(-1)), np.array(a.shape)


def test_batch(tz, images, n_tag_visible=0):
    """One collated test batch as the pipeline's transforms emit it (TransCaptionTensorizer with real_text_a_in_test False,
    ..._bertemb.py:226, 442-448): [MASK] captions; ``n_tag_visible`` > 0 attaches a text_b of n - 1 tag tokens + [SEP]."""
    text_b = ' '.join(['dog'] * (n_tag_visible - 1)) if n_tag_visible else ''
    r = tz.tensorize_ab('', text_b, real_text_a_in_test=False)
    B = images.shape[0]
    rep = lambda t: t.unsqueeze(0).expand(B, *t.shape).clone()           # noqa: E731
    return {'image': images, 'input_ids': rep(r['input_ids']), 'attention_mask': rep(r['attention_mask']),
            'token_type_ids': rep(r['segment_ids']), 'masked_pos': rep(r['masked_pos']), 'label': torch.zeros(B, 30522),
            'key': list(range(B))}, r


def wrapper_generate(wrap, tz, images, n_tag_visible=0):
    data, _ = test_batch(tz, images, n_tag_visible)
    with torch.no_grad(), G._MarginRecorder() as rec:
        ids, lp = wrap(data)
    m = rec.margins()
    return ids, lp, m


def part_masks(out, model, enc):
    from oracle import vitcap_oracle as O
    wrap, tok = G.build_wrapper(model, enc)
    tz = G.reference_tensorizer(tok, False)
    for n in (0, 7, 50):
        data, r = test_batch(tz, torch.zeros(2, 3, 384, 384), n)
        d = {'img_feats': torch.zeros(2, 577, 768), 'input_ids': data['input_ids'], 'attention_mask': data['attention_mask']}
        wrap.construct_attn_mask(d)
        full = d['attention_mask']
        assert full.shape == (2, 647, 647) and bool((full[0] == full[1]).all())
        out['test_n%d_input_ids' % n] = r['input_ids'].numpy().copy()
        out['test_n%d_mask70_bits' % n], out['test_n%d_mask70_shape' % n] = packbits(r['attention_mask'])
        out['test_n%d_full_bits' % n], out['test_n%d_full_shape' % n] = packbits(full[0])
        # the restatement every other generator used, checked here too (the committed test re-checks it against the stored bits)
        ids_o, am_o = O.test_text_inputs(1, n_tag_visible=n)
        assert torch.equal(am_o[0], r['attention_mask'].float()), 'oracle 70x70 test mask != reference tensorizer, n=%d' % n
        assert torch.equal(O.construct_attn_mask(am_o, 577)[0], full[0]), 'oracle joint mask != reference, n=%d' % n
        if n == 0:
            assert torch.equal(ids_o[0], r['input_ids']), 'oracle test input ids != reference tensorizer'
    ttz = G.reference_tensorizer(tok, True, mask_prob=0.15, max_masked_tokens=3)
    random.seed(1313)
    r = ttz.tensorize_ab(CAPTION13, '')
    n_tok = int((r['input_ids'] != 0).sum())
    assert n_tok == 15, n_tok           # [CLS] + 13 + [SEP]
    d = {'img_feats': torch.zeros(1, 577, 768), 'input_ids': r['input_ids'][None], 'attention_mask': r['attention_mask'][None]}
    wrap.construct_attn_mask(d)
    out['train13_seed'] = np.array([1313])
    out['train13_input_ids'] = r['input_ids'].numpy().copy()
    out['train13_origin_input_ids'] = np.asarray(r['origin_input_ids']).copy()
    out['train13_masked_pos'] = r['masked_pos'].numpy().copy()
    out['train13_masked_ids'] = r['masked_ids'].numpy().copy()
    out['train13_mask70_bits'], out['train13_mask70_shape'] = packbits(r['attention_mask'])
    out['train13_full_bits'], out['train13_full_shape'] = packbits(d['attention_mask'][0])
    assert torch.equal(O.construct_attn_mask(r['attention_mask'][None].float(), 577), d['attention_mask'])
    print('masks: stored (test n = 0 / 7 / 50, train 13 tokens, masked_pos sum %d)' % int(r['masked_pos'].sum()))


def part_pin(out, model, enc, sd_np):
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    gold = np.load(os.path.join(HERE, 'reference_vectors.npz'))
    wrap, tok = G.build_wrapper(model, enc)
    tz = G.reference_tensorizer(tok, False)
    img4 = torch.from_numpy(W.gen_image_batch(4, 1234))
    for name, images, n in (('greedy_b4', img4, 0), ('greedy_tags50_b2', img4[:2], 50)):
        ids, lp, m = wrapper_generate(wrap, tz, images, n)
        assert np.array_equal(ids.numpy(), gold[name + '_ids']), (name, ids.tolist(), gold[name + '_ids'].tolist())
        assert np.array_equal(lp.numpy(), gold[name + '_logprobs']), (name, lp.tolist(), gold[name + '_logprobs'].tolist())
        assert np.array_equal(m, gold[name + '_margins']), name
        out['wrapper_' + name + '_ids'] = ids.numpy().copy()
        out['wrapper_' + name + '_logprobs'] = lp.numpy().copy()
        print('pin:', name, 'through ImageCaptioning.forward == stored golden (ids, log-probs, margins bit for bit)')
    # training branch of the wrapper's forward: same batch as make_golden_train.py
    gt = np.load(os.path.join(HERE, 'reference_train.npz'))
    B = 2
    batch = O.synthetic_train_inputs(B)
    wrap.train()
    wrap.module.eval()          # dropout off, as make_golden_train.py (the training branch is chosen by the WRAPPER's mode)
    wrap.image_encoder.eval()
    wrap.iter = 1               # not a multiple of 100: the verbose accuracy logging stays off
    data = {'image': torch.from_numpy(W.gen_image_batch(B, 1234)), 'input_ids': batch['input_ids'].clone(),
            'attention_mask': batch['attention_mask'].clone(), 'masked_pos': batch['masked_pos'].clone(),
            'masked_ids': batch['masked_ids'].clone(), 'token_type_ids': batch['token_type_ids'], 'label': batch['label'],
            'key': [0, 1]}
    loss = wrap(data)['masked_loss']
    assert float(loss) == float(gt['masked_loss']), (float(loss), float(gt['masked_loss']))
    out['wrapper_train_masked_loss'] = np.array(float(loss))
    wrap.eval()
    print('pin: training branch of ImageCaptioning.forward: masked_loss', float(loss), '== stored golden')


def part_pop(pop, model, enc, save):
    from vitcap_amd import weights as W
    wrap, tok = G.build_wrapper(model, enc)
    tz = G.reference_tensorizer(tok, False)
    fams = (('noise', torch.from_numpy(W.gen_image_batch(16, 9001)), 9001), ('struct', torch.from_numpy(W.gen_structured_images(16, 9002)), 9002))
    for fam, images, seed in fams:
        pop['pop_%s_seed' % fam] = np.array([seed])
        ids_all, lp_all, m_all = [], [], []
        for c in range(0, 16, 4):
            ids, lp, m = wrapper_generate(wrap, tz, images[c:c + 4])
            ids_all.append(ids.numpy().copy()); lp_all.append(lp.numpy().copy()); m_all.append(m)
            pop['pop_%s_ids' % fam] = np.concatenate(ids_all)
            pop['pop_%s_logprobs' % fam] = np.concatenate(lp_all)
            pop['pop_%s_margins' % fam] = np.concatenate(m_all)
            save()
            print('pop:', fam, 'images', c, '..', c + 3, 'min margins', m.min(1).round(4).tolist(), flush=True)


def part_beam(pop, model, enc, sd_np, save):
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    sd_t = O.to_torch(sd_np)
    wrap, tok = G.build_wrapper(model, enc, num_beams=5)
    tz = G.reference_tensorizer(tok, False)
    fams = (('noise', torch.from_numpy(W.gen_image_batch(16, 9001))[:4]), ('struct', torch.from_numpy(W.gen_structured_images(16, 9002))[:4]))
    for fam, images in fams:
        ids_all, lp_all, m_all = [], [], []
        for c in range(0, 4, 2):
            data, _ = test_batch(tz, images[c:c + 2])
            with torch.no_grad():
                ids, lp = wrap(data)
                ids_o, lp_o, mg = O.beam_incremental(sd_t, images[c:c + 2], num_beams=5, emulate_bf16=False, return_margins=True)
            assert torch.equal(ids_o, ids), ('beam driver restatement != reference', fam, c)
            ids_all.append(ids.numpy().copy()); lp_all.append(lp.numpy().copy()); m_all.append(mg.numpy().copy())
            pop['beam5_%s_ids' % fam] = np.concatenate(ids_all)
            pop['beam5_%s_logprobs' % fam] = np.concatenate(lp_all)
            pop['beam5_%s_margins' % fam] = np.concatenate(m_all)
            save()
            print('beam:', fam, 'images', c, c + 1, 'min decision gaps', mg.min(1).values.tolist(), flush=True)


def main():
    parts = sys.argv[1:] or ['masks', 'pin', 'pop', 'beam']
    G.install_pipeline_shims()
    from vitcap_amd import weights as W
    torch.manual_seed(0)
    torch.set_num_threads(THREADS)
    sd_np = W.make_state_dict(seed=0, tie_weights=True)
    model, enc = G.build_reference('cls', True)
    G.load_recipe(model, enc, sd_np)
    wpath, ppath = os.path.join(HERE, 'reference_wrapper.npz'), os.path.join(HERE, 'reference_population.npz')
    out = dict(np.load(wpath)) if os.path.exists(wpath) else {}
    pop = dict(np.load(ppath)) if os.path.exists(ppath) else {}

    def save_pop():
        pop['torch_version'] = np.array([torch.__version__])
        np.savez_compressed(ppath, **pop)
    if 'masks' in parts:
        part_masks(out, model, enc)
    if 'pin' in parts:
        part_pin(out, model, enc, sd_np)
    if 'masks' in parts or 'pin' in parts:
        out['torch_version'] = np.array([torch.__version__])
        np.savez_compressed(wpath, **out)
        print('wrote', wpath)
    if 'pop' in parts:
        part_pop(pop, model, enc, save_pop)
    if 'beam' in parts:
        part_beam(pop, model, enc, sd_np, save_pop)


if __name__ == '__main__':
    main()
