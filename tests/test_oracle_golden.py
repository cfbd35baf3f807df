"""Pins the CPU oracle (oracle/vitcap_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vitcap_oracle as O
from vitcap_amd import weights as W

TOL = dict(rtol=2e-4, atol=2e-5)   # same ATen build, same op order -> differences are summation-order only


@pytest.fixture(scope='module')
def img():
    return torch.from_numpy(W.gen_image_batch(2, 1234))


@pytest.fixture(scope='module')
def enc_out(sd_t, img):
    torch.set_num_threads(8)
    with torch.no_grad():
        feats = O.patch_embed(sd_t, img)
        hid, tag = O.split_encoder(sd_t, feats)
    return feats, hid, tag


def test_recipe_matches_reference_layout(golden, sd_np):
    vec, meta = golden
    assert meta['keys_missing_in_reference'] == []
    assert meta['reference_keys_not_in_recipe'] == []
    assert meta['n_keys'] == len(sd_np) == 288
    for k, shp in meta['key_shapes'].items():
        assert list(sd_np[k].shape) == shp
    for k, dg in meta['digests'].items():
        assert W.tensor_digest(sd_np[k]) == dg, k


def test_a1_patch_embed(golden, enc_out):
    vec, meta = golden
    feats = enc_out[0]
    np.testing.assert_allclose(feats[0, :4].numpy(), vec['a1_img_feats_b0'], **TOL)
    assert abs(float(feats.double().mean()) - meta['a1']['mean']) < 1e-6


def test_a4_block(golden, sd_t, enc_out):
    vec, _ = golden
    with torch.no_grad():
        y = O.vit_block(sd_t, 'module.bert.encoder.blocks.0', enc_out[0])
    np.testing.assert_allclose(y[0, :3].numpy(), vec['a4_block0_rows'], **TOL)


def test_a5_split_encoder(golden, enc_out):
    vec, meta = golden
    _, hid, tag = enc_out
    np.testing.assert_allclose(hid[:, :2].numpy(), vec['a5_hidden_rows'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(tag[:, 0].numpy(), vec['a5_tag_hidden_cls'], rtol=1e-3, atol=1e-4)
    assert abs(float(hid.double().std()) - meta['a5_hidden']['std']) < 1e-4


def test_a6_tag_head(golden, sd_t, enc_out):
    vec, _ = golden
    with torch.no_grad():
        logit, prob, pred, tl = O.tag_head(sd_t, enc_out[2])
    np.testing.assert_allclose(logit[:, :64].numpy(), vec['a6_logit_head'], rtol=1e-3, atol=1e-4)
    np.testing.assert_array_equal(pred.numpy(), vec['a6_pred_topk'])
    np.testing.assert_array_equal(tl.numpy(), vec['a6_topk_len'])


def test_a9_bert_layer_and_a10_head(golden, sd_t):
    vec, _ = golden
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(1, 630, 768, generator=g) * 0.5
    m = torch.ones(1, 630, 630)
    m[:, :52, :52] = 0
    m[:, :2, :2] = torch.tril(torch.ones(2, 2))
    m[:, 52:, :52] = 0
    ext = (1.0 - m.unsqueeze(1)) * -10000.0
    with torch.no_grad():
        y = O.bert_layer(sd_t, 'module.bert.decoder.layer.0', xs, ext)
        z = O.lm_head(sd_t, 'module.cls', xs[:, :3])
    np.testing.assert_allclose(y[0, [0, 1, 2, 52, 629]].numpy(), vec['a9_rows'], **TOL)
    np.testing.assert_allclose(z[0, :, :128].numpy(), vec['a10_logits_head'], **TOL)


def test_step1_logits_row(golden, sd_t, img, enc_out):
    """One full `encode_forward(is_training=False)` call of the reference (decode step 1)."""
    vec, _ = golden
    input_ids, am = O.test_text_inputs(1)
    full = O.construct_attn_mask(am, 577)
    step_ids = torch.cat([torch.tensor([[101, 103]]), input_ids[:, 20:]], 1)
    mask = O._remove_rows_cols(full, 2, 20, 2, 20)
    pos = torch.cat([torch.arange(2), torch.arange(20, 70)]).unsqueeze(0)
    with torch.no_grad():
        logits = O.encode_forward_infer(sd_t, step_ids, enc_out[0][:1], mask, pos,
                                        torch.zeros(1, 52, dtype=torch.long),
                                        enc=(enc_out[1][:1], enc_out[2][:1]))
    np.testing.assert_allclose(logits[0, 1].numpy(), vec['step1_logits_row'], rtol=1e-3, atol=2e-4)


def test_greedy_as_written_token_exact(golden, sd_t, img):
    vec, _ = golden
    with torch.no_grad():
        ids, lp = O.greedy_as_written(sd_t, img, reuse_encoder=True)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_b2_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_b2_logprobs'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(ids[:1].numpy(), vec['greedy_b1_ids'])


def test_greedy_incremental_equals_as_written(golden, sd_t, img):
    """The incremental formulation (what the HIP path computes) reproduces the reference's tokens, log-probs AND the
    reference's own per-step top-2 margins (recorded by hooking torch.argmax inside the reference's generate loop)."""
    vec, _ = golden
    img4 = torch.from_numpy(W.gen_image_batch(4, 1234))
    with torch.no_grad():
        ids, lp, trace = O.greedy_incremental(sd_t, img4, emulate_bf16=False, return_trace=True)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_b4_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_b4_logprobs'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(ids[:2].numpy(), vec['greedy_b2_ids'])
    m = torch.stack([s['margin'] for s in trace['steps']], 1).numpy()
    np.testing.assert_allclose(m, vec['greedy_b4_margins'], rtol=0, atol=5e-5)


def test_greedy_alternative_eos_matches_reference(golden, sd_t):
    """eos_token_ids is a generate() kwarg (modeling_bert.py:928-933): with a frequently generated token as EOS the captions
    stop at data-dependent lengths -- finished rows emit PAD, the score counts the EOS step, unfinished rows get the forced
    EOS at position 19 (modeling_utils.py:855-877)."""
    vec, _ = golden
    eos = int(vec['alt_eos_id'][0])
    img4 = torch.from_numpy(W.gen_image_batch(4, 1234))
    with torch.no_grad():
        ids, lp = O.greedy_incremental(sd_t, img4, emulate_bf16=False, eos=eos)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_alteos_b4_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_alteos_b4_logprobs'], rtol=1e-5, atol=1e-5)
    lens = [(r != 0).sum() for r in vec['greedy_alteos_b4_ids'][:, 0]]
    assert len(set(lens)) >= 2, 'the alternative-EOS golden is meant to exercise different caption lengths'
    eos = int(vec['alt_eos_sel_id'][0])
    cand = torch.from_numpy(W.gen_image_batch(16, int(vec['sel_image_seed'][0])))[torch.from_numpy(vec['sel_index'])]
    with torch.no_grad():
        ids, lp = O.greedy_incremental(sd_t, cand, emulate_bf16=False, eos=eos)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_alteos_sel_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_alteos_sel_logprobs'], rtol=1e-5, atol=1e-5)


def test_greedy_selected_images_match_reference(golden, sd_t):
    """The 4 best-conditioned of 16 candidate images (tests/golden/make_golden.py): whole captions comparable in bf16."""
    vec, _ = golden
    cand = torch.from_numpy(W.gen_image_batch(16, int(vec['sel_image_seed'][0])))[torch.from_numpy(vec['sel_index'])]
    with torch.no_grad():
        ids, lp = O.greedy_incremental(sd_t, cand, emulate_bf16=False)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_sel_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_sel_logprobs'], rtol=1e-5, atol=1e-5)
    assert float(vec['greedy_sel_margins'].min()) > 0.012


@pytest.mark.slow
def test_greedy_untied_notebook_flow(golden, img):
    """Notebook flow: tagemb=None, tie_weights=False (Loading Script.ipynb cell 10)."""
    vec, _ = golden
    sd = O.to_torch(W.make_state_dict(seed=0, tie_weights=False))
    with torch.no_grad():
        ids, lp = O.greedy_as_written(sd, img[:1], tagemb=None, reuse_encoder=True)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_untied_nocls_b1_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_untied_nocls_b1_logprobs'], rtol=1e-5, atol=1e-5)


def test_greedy_untied_incremental_matches_reference(golden):
    """BASELINE configs[0] flow (tie_weights=False, tagemb=None) on the incremental formulation, B=2."""
    vec, _ = golden
    sd = O.to_torch(W.make_state_dict(seed=0, tie_weights=False))
    with torch.no_grad():
        ids, lp = O.greedy_incremental(sd, torch.from_numpy(W.gen_image_batch(2, 1234)), emulate_bf16=False)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_untied_nocls_b2_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_untied_nocls_b2_logprobs'], rtol=1e-5, atol=1e-5)


@pytest.mark.slow
def test_tags_visible_as_written_matches_reference(golden, sd_t, img):
    """the mask with the first n tag slots visible to the caption (dataset.py:240-252, 387-390) through the
    oracle's restatement of ViTSplitCLSEmbModel.forward (both embedding branches) against the reference's own captions."""
    vec, _ = golden
    with torch.no_grad():
        ids, lp = O.greedy_as_written(sd_t, img, reuse_encoder=True, n_tag_visible=50)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_tags50_b2_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_tags50_b2_logprobs'], rtol=1e-5, atol=1e-5)
    assert not np.array_equal(vec['greedy_tags50_b2_ids'], vec['greedy_b2_ids'])
    assert int(vec['tags_topk_len_b2'][0]) == 50            # recipe: branch B at steps 1..18, branch A at step 19


@pytest.mark.slow
def test_tag_slot_start_position_as_written_matches_reference():
    """od_labels_start_posid = 40 (> max_length) with the tags visible, notebook flow: the oracle's as-written path against the
    reference's own output (tests/golden/make_golden_tagpos.py); the position ids reach bert.extra_embeddings (:1484-1485)."""
    vec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_tagpos.npz')))
    sd = O.to_torch(W.make_state_dict(seed=0, tie_weights=False, vbias_std=float(vec['untied_vbias_std'][0])))
    img = torch.from_numpy(W.gen_structured_images(48, int(vec['image_seed'][0])))[torch.from_numpy(vec['sel_index'])]
    with torch.no_grad():
        ids, lp = O.greedy_as_written(sd, img, tagemb=None, reuse_encoder=True, n_tag_visible=50, od_labels_start_posid=40)
    np.testing.assert_array_equal(ids.numpy(), vec['untied_pos40_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['untied_pos40_logprobs'], rtol=1e-5, atol=1e-5)
    assert not np.array_equal(vec['untied_pos40_ids'], vec['untied_pos20_ids'])


def test_beam2_as_written_matches_reference(golden, sd_t, img):
    """a13: the beam driver + BeamHypotheses restatement against the reference's own beam=2 output."""
    vec, _ = golden
    with torch.no_grad():
        ids, lp = O.beam_as_written(sd_t, img[:1], num_beams=2)
    np.testing.assert_array_equal(ids.numpy(), vec['beam2_b1_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['beam2_b1_logprobs'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name,nb,B', [('beam5_b2', 5, 2), ('beam3_alteos_b2', 3, 2), ('beam5_sel', 5, 2)])
def test_beam_incremental_matches_reference(golden, sd_t, name, nb, B):
    """beam=5 at B=2, beam=3 with the alternative EOS (hypotheses finishing at many lengths), beam=5 on the two
    best-conditioned candidate images: ids, scores and the decision margins the generator stored."""
    vec, _ = golden
    if name == 'beam5_sel':
        im = torch.from_numpy(W.gen_image_batch(16, int(vec['sel_image_seed'][0])))[torch.from_numpy(vec['beam_sel_index'])]
    else:
        im = torch.from_numpy(W.gen_image_batch(B, 1234))
    eos = int(vec['alt_eos_id'][0]) if 'alteos' in name else 102
    with torch.no_grad():
        ids, lp, mg = O.beam_incremental(sd_t, im, num_beams=nb, emulate_bf16=False, eos=eos, return_margins=True)
    np.testing.assert_array_equal(ids.numpy(), vec[name + '_ids'])
    np.testing.assert_allclose(lp.numpy(), vec[name + '_logprobs'], rtol=2e-5, atol=2e-5)
    got, want = mg.numpy(), vec[name + '_margins']
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all()
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=5e-5)


def test_beam_nbest_incremental_matches_reference(sd_t):
    """a11/a13 with num_keep_best > 1: BeamHypotheses' n-best list (add / worst_score / is_done, modeling_utils.py:1138-1180)
    and the final best-first selection, against the reference's own output (tests/golden/make_golden_nbest.py).  The
    incremental fp32 formulation is used (equal to the as-written one, next test) so the case runs in seconds."""
    import os
    from vitcap_amd import weights as W
    vec = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_nbest.npz'))
    n = 0
    while 'case%d_cfg' % n in vec:
        beams, keep, lp, B = vec['case%d_cfg' % n]
        im = torch.from_numpy(W.gen_image_batch(int(B), int(vec['image_seed'])))
        with torch.no_grad():
            ids, logp = O.beam_incremental(sd_t, im, num_beams=int(beams), emulate_bf16=False, length_penalty=float(lp),
                                           num_keep_best=int(keep))
        assert tuple(ids.shape) == (int(B), int(keep), 20)
        np.testing.assert_array_equal(ids.numpy(), vec['case%d_ids' % n])
        np.testing.assert_allclose(logp.numpy(), vec['case%d_logprobs' % n], rtol=2e-5, atol=2e-5)
        n += 1
    assert n == 2


def test_repetition_penalty_matches_reference(sd_t):
    """generate(repetition_penalty != 1), greedy and beam: the oracle's CTRL penalty (distinct prefix tokens, multiply negative
    logits / divide positive ones, applied before argmax / log_softmax) against the reference's own output
    (tests/golden/make_golden_reppen.py)."""
    import os
    from vitcap_amd import weights as W
    vec = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_reppen.npz'))
    n = 0
    while 'case%d_cfg' % n in vec:
        beams, rp, B = vec['case%d_cfg' % n]
        im = torch.from_numpy(W.gen_image_batch(int(B), int(vec['image_seed'])))
        with torch.no_grad():
            if int(beams) == 1:
                ids, logp = O.greedy_incremental(sd_t, im, emulate_bf16=False, repetition_penalty=float(rp))
            else:
                ids, logp = O.beam_incremental(sd_t, im, num_beams=int(beams), emulate_bf16=False, repetition_penalty=float(rp))
        np.testing.assert_array_equal(ids.numpy(), vec['case%d_ids' % n])
        np.testing.assert_allclose(logp.numpy(), vec['case%d_logprobs' % n], rtol=2e-5, atol=2e-5)
        n += 1
    assert n == 3


def test_beam_incremental_equals_as_written(sd_t, img):
    with torch.no_grad():
        a = O.beam_as_written(sd_t, img[:1], num_beams=3)
        b = O.beam_incremental(sd_t, img[:1], num_beams=3, emulate_bf16=False)
    np.testing.assert_array_equal(a[0].numpy(), b[0].numpy())
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), rtol=1e-5, atol=1e-5)


def test_image_dependent_family_pins_the_oracle(sd_t):
    """tests/golden/reference_imgdep.npz (the reference on structured images; captions differ between images in most positions):
    the oracle's fp32 incremental path reproduces the reference's tokens, log-probs and per-step [MASK]-row logits."""
    import os
    vec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_imgdep.npz')))
    cand = torch.from_numpy(W.gen_structured_images(48, int(vec['image_seed'][0])))
    img = cand[torch.from_numpy(vec['sel_index'])]
    caps = vec['greedy_ids'][:, 0]
    assert min(float((caps[i, 1:19] != caps[j, 1:19]).mean()) for i in range(4) for j in range(i)) >= 0.5
    torch.set_num_threads(8)
    with torch.no_grad():
        ids, lp, tr = O.greedy_incremental(sd_t, img, emulate_bf16=False, return_trace=True)
    np.testing.assert_array_equal(ids.numpy(), vec['greedy_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['greedy_logprobs'], rtol=0, atol=2e-4)
    for i, step in enumerate(vec['step_list']):
        row = tr['steps'][int(step) - 1]['logits_row'].numpy()
        got = np.take_along_axis(row, vec['step_cols'][i].astype(np.int64), axis=1)
        np.testing.assert_allclose(got, vec['step_logits'][i], rtol=1e-3, atol=3e-4)
    # eos_token_ids with three entries (modeling_utils.py:862-871): the restated greedy loop stops at any of them
    eos = [int(x) for x in vec['multi_eos_ids_list']]
    with torch.no_grad():
        ids, lp = O.greedy_incremental(sd_t, img, emulate_bf16=False, eos=eos)
    np.testing.assert_array_equal(ids.numpy(), vec['multi_eos_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['multi_eos_logprobs'], rtol=0, atol=2e-4)


def _unpack(vec, name):
    shape = tuple(int(x) for x in vec[name + '_shape'])
    n = int(np.prod(shape))
    return np.unpackbits(vec[name + '_bits'])[:n].reshape(shape).astype(np.float32)


@pytest.mark.parametrize('n_tag', [0, 7, 50])
def test_construct_attn_mask_equals_reference(n_tag):
    """a8 pinned to the reference's OWN code: tests/golden/reference_wrapper.npz holds the 70 x 70 mask of the reference's
    CaptionTensorizer.tensorize_ab (dataset.py:206-417, test mode, text_b of n_tag slots) and the 647 x 647 joint mask of the
    reference's ImageCaptioning.construct_attn_mask (..._bertemb.py:57-85), both produced by running those functions
    (tests/golden/make_golden_wrapper.py).  The oracle's restatements -- which every other generator and every device test feed
    to both sides -- must reproduce them exactly, and so must the reference tensorizer's input ids."""
    from oracle import vitcap_oracle as O
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_wrapper.npz'))
    ids, am = O.test_text_inputs(2, n_tag_visible=n_tag)
    want70 = _unpack(vec, 'test_n%d_mask70' % n_tag)
    assert np.array_equal(am[0].numpy(), want70) and np.array_equal(am[1].numpy(), want70)
    full = O.construct_attn_mask(am, 577)
    assert tuple(full.shape) == (2, 647, 647)
    assert np.array_equal(full[1].numpy(), _unpack(vec, 'test_n%d_full' % n_tag))
    if n_tag == 0:
        assert np.array_equal(ids[0].numpy(), vec['test_n0_input_ids'])
    else:       # text_b tokens sit in the tag slots of the reference's ids; the model overwrites those rows' embeddings (modeling_bert.py:1449-1489)
        assert np.array_equal(ids[0].numpy()[:20], vec['test_n%d_input_ids' % n_tag][:20])


def test_construct_attn_mask_train_equals_reference():
    """Training mask with a 13-token caption: the reference's tensorize_ab (train mode, seed 1313) + construct_attn_mask."""
    from oracle import vitcap_oracle as O
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_wrapper.npz'))
    m70 = _unpack(vec, 'train13_mask70')
    ids = vec['train13_input_ids']
    n_tok = int((ids != 0).sum())
    assert n_tok == 15 and int(vec['train13_masked_pos'].sum()) >= 1
    # structure the trainer assumes (vitcap_amd/train.py check_text_inputs): causal over the caption's tokens, nothing else visible
    want = np.zeros((70, 70), dtype=np.float32)
    want[:n_tok, :n_tok] = np.tril(np.ones((n_tok, n_tok), dtype=np.float32))
    assert np.array_equal(m70, want)
    full = O.construct_attn_mask(torch.from_numpy(m70)[None], 577)
    assert np.array_equal(full[0].numpy(), _unpack(vec, 'train13_full'))


def test_wrapper_forward_reproduced_the_goldens():
    """a16: greedy_b4, greedy_tags50_b2 and the training loss were produced a second time THROUGH the reference's
    ImageCaptioning.forward (mask by its own construct_attn_mask, inputs by its own tensorizer): same ids / log-probs / loss bit for bit
    as the goldens the oracle-built inputs gave (asserted in the generator; the wrapper's outputs are stored and compared here)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    w, g, t = (np.load(os.path.join(here, f)) for f in ('reference_wrapper.npz', 'reference_vectors.npz', 'reference_train.npz'))
    for name in ('greedy_b4', 'greedy_tags50_b2'):
        assert np.array_equal(w['wrapper_%s_ids' % name], g[name + '_ids'])
        assert np.array_equal(w['wrapper_%s_logprobs' % name], g[name + '_logprobs'])
    assert float(w['wrapper_train_masked_loss']) == float(t['masked_loss'])


# --------------------------------------------------------------------------------------------
# f4  constrained beam search (utils_cbs.py:26-443, 646-871): tests/golden/make_golden_cbs.py ran the reference's own
# ViTCAP.generate(use_cbs=True) and its FiniteStateMachineBuilder
# --------------------------------------------------------------------------------------------
def _cbs_vec():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_cbs.npz'))


def _cbs_case(vec, n):
    B, K, max_given, S = [int(x) for x in vec['case%d_cfg' % n]]
    tab = vec['case%d_constraint_ids' % n]
    cons = []
    for b in range(B):
        per = []
        for c in range(tab.shape[1]):
            words = [[int(f) for f in tab[b, c, w] if f >= 0] for w in range(tab.shape[2]) if (tab[b, c, w] >= 0).any()]
            if words:
                per.append(words)
        cons.append(per)
    fsm = torch.stack([O.fsm_build(per, int(vec['vocab_size']), max_given, 4)[0][:S, :S] for per in cons])
    return B, K, max_given, S, cons, fsm, torch.from_numpy(vec['case%d_num_constraints' % n])


def test_cbs_fsm_builders_equal_the_reference_machines():
    """The oracle's fsm_build (token ids in) and the shipped host builder vitcap_amd.cbs.FiniteStateMachineBuilder (tokenizer + word
    form tables in, the reference's constructor) against digests of the machines the REFERENCE builder produced: words per
    transition and a position-weighted checksum per transition, for single- and two-word constraints and 2 / 3 given slots."""
    from vitcap_amd import cbs
    vec = _cbs_vec()
    V = int(vec['vocab_size'])
    wts = (torch.arange(V, dtype=torch.int64) % 8191) + 1
    table = lambda key: {kv.split('=')[0]: kv.split('=')[1].split(',') for kv in vec[key].tolist()}
    c2t, forms = table('c2t'), table('forms')
    word_id = {}          # the words' ids, read back from the stored constraint tables (no vocabulary file needed)
    n = 0
    while 'case%d_cfg' % n in vec:
        tab = vec['case%d_constraint_ids' % n]
        for b, names in enumerate(vec['case%d_constraint_names' % n].tolist()):
            for c, name in enumerate(names.split('|')):
                toks = [t for w in name.split() for t in c2t[w]]
                for w, t in enumerate(toks):
                    for f, form in enumerate(forms.get(t, [t])):
                        word_id[form] = int(tab[b, c, w, f])
        n += 1

    class Tok(object):
        vocab_size = V

        @staticmethod
        def convert_tokens_to_ids(tokens):
            return [word_id[t] for t in tokens]

    n = 0
    while 'case%d_cfg' % n in vec:
        B, K, max_given, S, cons, fsm, ncons = _cbs_case(vec, n)
        builder = cbs.FiniteStateMachineBuilder(Tok, c2t, forms, max_given)
        fsm2, ncons2 = cbs.batch_fsm(builder, [s.split('|') for s in vec['case%d_constraint_names' % n].tolist()])
        assert fsm2.shape == fsm.shape and bool((fsm2 == fsm).all()) and ncons2.tolist() == ncons.tolist()
        np.testing.assert_array_equal(fsm.long().sum(-1).numpy(), vec['case%d_fsm_count' % n])
        np.testing.assert_array_equal((fsm.long() * wts).sum(-1).numpy(), vec['case%d_fsm_check' % n])
        n += 1
    assert n == 4


def test_cbs_incremental_matches_reference(sd_t):
    """ViTCAP.generate(use_cbs=True) of the reference (search + select_best_beam_with_constraints) against the oracle's restatement
    on the incremental fp32 formulation: returned ids and log-probabilities of four cases (1 and 2 images, 1..3 beams per state,
    single- and two-word constraints; with 2 images the first step reads image 0's distribution for both, as written; one case with
    decoding_constraint_flag and bad_ending_ids), and every valid state's best beam as `search` returned it."""
    vec = _cbs_vec()
    n = 0
    while 'case%d_cfg' % n in vec:
        B, K, max_given, S, cons, fsm, ncons = _cbs_case(vec, n)
        im = torch.from_numpy(W.gen_image_batch(B, int(vec['image_seed'])))
        rules = {}
        if 'case%d_no_repeat' % n in vec:     # generate's decoding_constraint_flag / bad_ending_ids (utils_cbs.py:187-198)
            rules = {'no_repeat': bool(vec['case%d_no_repeat' % n]), 'bad_ending_ids': vec['case%d_bad_ending_ids' % n].tolist()}
        with torch.no_grad():
            ids, lp, m_search, m_sel, beams, scores = O.cbs_incremental(sd_t, im, fsm, ncons, K, 2, return_margins=True, **rules)
        np.testing.assert_array_equal(ids.numpy(), vec['case%d_ids' % n])
        np.testing.assert_allclose(lp.numpy(), vec['case%d_logprobs' % n], rtol=0, atol=2e-4)
        want_b, want_s = vec['case%d_beams' % n], vec['case%d_scores' % n]
        for b in range(B):
            given = int(ncons[b])
            for s in range(2 ** given):
                if bin(s).count('1') >= min(given, 2):
                    np.testing.assert_array_equal(beams[b, s, 0].numpy(), want_b[b, s, 0])
                    np.testing.assert_allclose(float(scores[b, s, 0]), want_s[b, s, 0], rtol=0, atol=2e-3)
        np.testing.assert_allclose(m_search.numpy(), vec['case%d_margin_search' % n], rtol=0, atol=1e-4)
        n += 1
    assert n == 4


@pytest.mark.slow
def test_cbs_as_written_matches_reference(sd_t):
    """The same with the model re-run on every prefix, as the reference does (its `state` stays None): smallest case."""
    vec = _cbs_vec()
    B, K, max_given, S, cons, fsm, ncons = _cbs_case(vec, 0)
    im = torch.from_numpy(W.gen_image_batch(B, int(vec['image_seed'])))
    with torch.no_grad():
        ids, lp = O.cbs_as_written(sd_t, im, fsm, ncons, K, 2)
    np.testing.assert_array_equal(ids.numpy(), vec['case0_ids'])
    np.testing.assert_allclose(lp.numpy(), vec['case0_logprobs'], rtol=0, atol=2e-4)


def test_cbs_search_on_a_table_model():
    """Bookkeeping invariants on a synthetic table model (no network): beams of a state only ever pass through words its machine
    allows, a finished beam pads with EOS at no cost, the search stops once every slot has ended."""
    g = torch.Generator().manual_seed(5)
    Vn, B, K = 40, 2, 2
    fsm1, used = O.fsm_build([[[7, 8]], [[9], [10]]], Vn, max_given_constraints=2, max_words_per_constraint=2)
    fsm = torch.stack([fsm1[:used, :used]] * B)
    tab = torch.randn(64, Vn, generator=g)
    tab[:, 2] += 4.0                                    # word 2 = EOS is likely: the sequences end early

    def step(ids, parents):
        return tab[(ids[:, -1] * 5 + ids.shape[1] * 11) % 64]
    beams, scores = O.cbs_search(step, fsm, K, max_length=12, eos=2)
    S = fsm.shape[1]
    assert beams.shape[:3] == (B, S, K) and beams.shape[3] <= 11
    for b in range(B):
        for s in range(4):
            for k in range(K):
                if scores[b, s, k] < -1e19:
                    continue
                seq, state = beams[b, s, k].tolist(), 0
                for i, w in enumerate(seq):
                    nxt = [s2 for s2 in range(S) if fsm[b, state, s2, w]]
                    assert nxt, (b, s, k, i)
                    state = nxt[-1] if len(nxt) > 1 and nxt[-1] != state else nxt[0]
                    if w == 2:
                        assert all(x == 2 for x in seq[i:])
                        break
    ids, lp = O.cbs_select_best(beams, scores, torch.tensor([2, 2]), 2, eos=2)
    for b in range(B):                                   # both constraints are in the chosen caption: 7|8, and 9 followed by 10
        seq = ids[b].tolist()
        assert (7 in seq or 8 in seq) and any(a == 9 and c == 10 for a, c in zip(seq, seq[1:]))
