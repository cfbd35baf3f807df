"""End-to-end parity of the HIP captioning engine against the REFERENCE's own outputs (tests/golden) and the CPU oracle.

What is asserted (tolerances stated inline; the noise floors are defined in tests/conftest.py):
  1. bf16 HIP path vs the reference's fp32 tokens (goldens produced by running /root/reference itself): token ids
     IDENTICAL on every decision whose reference margin exceeds the bf16 noise floor -- whole captions on the four
     well-conditioned images, comparable prefixes elsewhere -- for the pipeline flow (tied, tagemb='cls'), the notebook flow
     (untied, tagemb=None: BASELINE configs[0]) and a run with another EOS token (captions of different lengths); caption
     log-probs within 1e-2 (bf16 vs fp32).
  2. bf16 HIP path vs the oracle's bf16-rounding emulation of the SAME incremental algorithm (for options without a
     reference golden: n-best lists, repetition penalty, max_length): encoder activations within 2e-2 relative L2, last-step
     logits within 2e-2, token ids identical wherever the emulation's margin exceeds the same noise floor, log-probs within
     1e-2; beam results conditioned on the oracle's decision gaps, else scores within 1e-2.
  3. properties at the benchmark sizes (B=64 greedy, 256 x beam 5): determinism, batch-composition invariance, well-formed
     ids; hipGraph replay, early exit, max_length != 20, option and text-input validation, two host threads on one engine.
"""
import os
import threading

import numpy as np
import pytest
import torch

from conftest import (BEAM_MARGIN_FLOOR, GREEDY_MARGIN_FLOOR, assert_tokens_match_reference, relevant_margins)

pytestmark = pytest.mark.gpu

# The oracle's bf16 emulation rounds where the kernels store bf16, but the kernels' fp32 arithmetic differs from ATen's
# (polynomial erfc GELU, exp2-domain softmax, summation order): a difference below one bf16 ulp before a rounding point
# becomes a full ulp after it, so device-vs-emulation noise is of the same size as device-vs-fp32 (measured: encoder output
# 3.9e-3 relative L2 against the emulation, 5.6e-3 against fp32).  The same floors apply as against the reference.
MARGIN_TOL = GREEDY_MARGIN_FLOOR
BEAM_GAP_TOL = BEAM_MARGIN_FLOOR
# beam goldens (name, image) accepted on score evidence instead of identical ids -- reviewed exceptions only; empty = all exact
BEAM_SCORE_ONLY_OK = {
    # round 3 (dense attention now sums the bf16-ROUNDED probabilities through the matrix pipe): image 1 of beam5_b2 resolves a
    # 9.2e-6 decision gap (floor 3e-2) the other way; best score -6.44116 vs the reference's -6.44042.
    # round 4, justified by a RATE (tools/exact_rate_ab.sh, 32 greedy + 18 beam reference fixtures, MI355X): shipped normaliser
    # (rounded P, matrix pipe) greedy 31/32, beam 15/18; rounded P on the vector ALU the same 31/32, 15/18 (same misses); UNROUNDED P
    # (round 2's form) greedy 30/32, beam 15/18 -- it wins this image back and loses three population images and one greedy caption.
    ('beam5_b2', 1),
    ('beam3_alteos_b2', 1),      # the same image and the same 9.2e-6 gap (the alternative EOS is not reached before it)
}


@pytest.fixture(scope='module')
def model():
    assert torch.cuda.is_available()
    from vitcap_amd.model import ImageCaptioning
    m = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    m.pack('cuda')
    return m


@pytest.fixture(scope='module')
def oracle_run(sd_t):
    from oracle import vitcap_oracle as O
    from vitcap_amd import weights as W
    torch.set_num_threads(max(1, torch.get_num_threads()))
    img = torch.from_numpy(W.gen_image_batch(4, 1234))
    with torch.no_grad():
        ids, lp, tr = O.greedy_incremental(sd_t, img, emulate_bf16=True, return_trace=True)
    return img, ids, lp, tr


def _images(n, seed=1234):
    from vitcap_amd import weights as W
    return torch.from_numpy(W.gen_image_batch(n, seed))


def _selected(vec, key='sel_index'):
    return _images(16, int(vec['sel_image_seed'][0]))[torch.from_numpy(vec[key])]


def test_state_dict_roundtrip(model, sd_np):
    sd = model.state_dict()
    assert list(sd.keys()) == list(sd_np.keys())
    for k, v in sd_np.items():
        assert tuple(sd[k].shape) == v.shape
    assert sd['module.cls.predictions.decoder.weight'].data_ptr() == \
        sd['module.bert.embeddings.word_embeddings.weight'].data_ptr()


# ------------------------------------------------------------------------------------------------ vs the reference
def test_greedy_tokens_equal_reference_goldens(model, golden):
    """Token ids of the reference itself (fp32, full re-encode per step) on the same seeded weights / images."""
    vec, _ = golden
    # (a) the four well-conditioned images: every decision is above the floor -> whole captions must be identical
    ids, lp = model.generate(_selected(vec).cuda())
    rep = assert_tokens_match_reference(ids.cpu().numpy(), vec['greedy_sel_ids'], vec['greedy_sel_margins'],
                                        GREEDY_MARGIN_FLOOR, min_full=4, what='greedy/selected')
    assert all(r[4] for r in rep)
    np.testing.assert_allclose(lp.cpu().numpy(), vec['greedy_sel_logprobs'], rtol=0, atol=1e-2)
    # (b) the default images 0..3: identical up to each caption's first ill-conditioned decision
    ids, lp = model.generate(_images(4).cuda())
    rep = assert_tokens_match_reference(ids.cpu().numpy(), vec['greedy_b4_ids'], vec['greedy_b4_margins'],
                                        GREEDY_MARGIN_FLOOR, min_full=0, what='greedy/B=4')
    print('greedy B=4 (sequence, comparable decisions, whole, prefix ok, whole caption equal):', rep)
    np.testing.assert_allclose(lp.cpu().numpy(), vec['greedy_b4_logprobs'], rtol=0, atol=1e-2)
    # B=1 and B=2 goldens are the same images in smaller batches (batch invariance of the reference and of the engine)
    assert np.array_equal(vec['greedy_b4_ids'][:2], vec['greedy_b2_ids']) and np.array_equal(vec['greedy_b4_ids'][:1], vec['greedy_b1_ids'])
    ids1, _ = model.generate(_images(1).cuda())
    assert torch.equal(ids1, ids[:1])


def test_greedy_alternative_eos_equals_reference(model, golden):
    """generate(eos_token_ids=[x]) with a frequently generated token: captions end at different lengths; PAD after EOS, the
    score counts the EOS step, forced EOS at the last position for unfinished rows (modeling_utils.py:855-877).  On the
    well-conditioned images whole captions are comparable; on images 0..3 the comparable prefixes."""
    vec, _ = golden
    for key, eos_key, img, min_full in (('greedy_alteos_sel', 'alt_eos_sel_id', _selected(vec), 4), ('greedy_alteos_b4', 'alt_eos_id', _images(4), 0)):
        eos = int(vec[eos_key][0])
        ids, lp = model.generate(img.cuda(), eos_token_ids=[eos])
        want = vec[key + '_ids']
        rep = assert_tokens_match_reference(ids.cpu().numpy(), want, vec[key + '_margins'], GREEDY_MARGIN_FLOOR, min_full=min_full,
                                            what=key, eos=eos)
        print(key, rep)
        got = ids.cpu().numpy()
        same = (got == want).all(-1).all(-1)
        np.testing.assert_allclose(lp.cpu().numpy()[same], vec[key + '_logprobs'][same], rtol=0, atol=1e-2)
        assert len({int((r != 0).sum()) for r in want[:, 0]}) >= 2, 'captions of different lengths expected'


def test_untied_notebook_flow_equals_reference(golden):
    """BASELINE configs[0]: the notebook's model (tie_weights=False, tagemb=None, Loading Script.ipynb cell 10) on the device."""
    from vitcap_amd.model import ImageCaptioning
    vec, _ = golden
    m = ImageCaptioning(tie_weights=False, tagemb=None).load_recipe(0).eval()
    m.pack('cuda')
    sd = m.state_dict()
    assert sd['module.cls.predictions.decoder.weight'].data_ptr() != sd['module.bert.embeddings.word_embeddings.weight'].data_ptr()
    img = _images(2).cuda()
    ids, lp = m({'image': img, 'key': [0, 1]})
    rep = assert_tokens_match_reference(ids.cpu().numpy(), vec['greedy_untied_nocls_b2_ids'], vec['greedy_untied_nocls_b2_margins'],
                                        GREEDY_MARGIN_FLOOR, min_full=2, what='untied notebook flow')
    assert all(r[4] for r in rep)
    np.testing.assert_allclose(lp.cpu().numpy(), vec['greedy_untied_nocls_b2_logprobs'], rtol=0, atol=1e-2)
    ids1, lp1 = m.generate(img[:1].contiguous())
    assert np.array_equal(ids1.cpu().numpy(), vec['greedy_untied_nocls_b1_ids'])


# ------------------------------------------------------------------------------------------------ image-dependent family
@pytest.fixture(scope='module')
def imgdep():
    """tests/golden/reference_imgdep.npz (make_golden_imgdep.py): the reference on STRUCTURED images, whose captions differ between
    images in most positions -- parity on them is sensitive to everything image-dependent (encoder, visual K/V, cross-attention)."""
    import os
    from vitcap_amd import weights as W
    vec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_imgdep.npz')))
    cand = torch.from_numpy(W.gen_structured_images(48, int(vec['image_seed'][0])))
    return vec, cand


def test_image_dependent_captions_equal_reference(model, imgdep):
    """Whole captions, token for token, on four images whose reference captions share at most 2 of 18 positions pairwise; every
    decision clears the bf16 noise floor (min margin 0.0185), so nothing is excused."""
    vec, cand = imgdep
    want = vec['greedy_ids']
    caps = want[:, 0]
    assert min(float((caps[i, 1:19] != caps[j, 1:19]).mean()) for i in range(4) for j in range(i)) >= 0.5
    assert float(vec['greedy_margins'].min()) > GREEDY_MARGIN_FLOOR
    img = cand[torch.from_numpy(vec['sel_index'])].cuda()
    ids, lp = model({'image': img, 'key': [0, 1, 2, 3]})
    rep = assert_tokens_match_reference(ids.cpu().numpy(), want, vec['greedy_margins'], GREEDY_MARGIN_FLOOR, min_full=4,
                                        what='image-dependent goldens')
    assert all(r[4] for r in rep)
    np.testing.assert_allclose(lp.cpu().numpy(), vec['greedy_logprobs'], rtol=0, atol=1e-2)
    # the same images one at a time (other batch composition, other tile plans): identical ids
    ids1, _ = model.generate(img[2:3].contiguous())
    assert torch.equal(ids1.cpu(), ids[2:3].cpu())


# ---- population statistic (round 4): 32 images nobody selected -- 16 uniform-noise (seed 9001), 16 structured (seed 9002) -- captioned
# by the reference THROUGH its own ImageCaptioning.forward (tests/golden/make_golden_wrapper.py, reference_population.npz).
POP_MIN_EXACT = 0.90          # measured on MI355X (round 4): 31 / 32 (the one other caption first differs at a margin of 0.0006); a regression must not hide
POP_BEAM_MIN_EXACT = 0.75     # measured: 7 / 8


@pytest.fixture(scope='module')
def population():
    from vitcap_amd import weights as W
    vec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_population.npz')))
    imgs = {'noise': torch.from_numpy(W.gen_image_batch(16, int(vec['pop_noise_seed'][0]))),
            'struct': torch.from_numpy(W.gen_structured_images(16, int(vec['pop_struct_seed'][0])))}
    return vec, imgs


def test_population_greedy_captions_vs_reference(model, population):
    """Not a hand-picked set: on 32 images the device's greedy caption equals the reference's token for token, or the FIRST
    differing position is a decision whose reference margin is below the bf16 noise floor (tests/conftest.py) -- asserted for every
    image; the share of exactly equal captions is printed and must not fall below POP_MIN_EXACT."""
    vec, imgs = population
    exact, whole, n, first_bad_margins = 0, 0, 0, []
    for fam in ('noise', 'struct'):
        want, margins = vec['pop_%s_ids' % fam], vec['pop_%s_margins' % fam]
        ids, lp = model({'image': imgs[fam].cuda(), 'key': list(range(16))})
        got = ids.cpu().numpy()
        rep = assert_tokens_match_reference(got, want, margins, GREEDY_MARGIN_FLOOR, min_full=0, what='population ' + fam)
        for b, p, is_whole, ok, equal in rep:
            n += 1
            exact += int(equal)
            whole += int(is_whole)
            if not equal:
                d = int(np.nonzero(got[b, 0] != want[b, 0])[0][0])          # first differing position; decision d-1 chose it
                m = relevant_margins(want[b, 0], margins[b], 102)
                first_bad_margins.append(float(m[d - 1]))
                assert m[d - 1] < GREEDY_MARGIN_FLOOR, (fam, b, d, float(m[d - 1]))
        np.testing.assert_allclose(lp.cpu().numpy()[[r[0] for r in rep if r[4]]], vec['pop_%s_logprobs' % fam][[r[0] for r in rep if r[4]]], rtol=0, atol=1e-2)
    print('population greedy: %d / %d captions exactly equal to the reference (%d whole-caption comparable); margins at the first '
          'differing decision of the others: %s (floor %.3g)' % (exact, n, whole, ['%.4f' % x for x in sorted(first_bad_margins)], GREEDY_MARGIN_FLOOR))
    assert exact >= whole, 'a whole-caption-comparable image differs'
    assert exact / n >= POP_MIN_EXACT


def test_population_beam5_vs_reference(model, population):
    """Beam = 5 on 8 of those images (reference through its own wrapper; decision gaps from the oracle's driver restatement, which
    reproduced the reference's ids exactly when the fixture was made).  Identical ids wherever every gap clears the floor; otherwise
    identical ids or a length-normalised score within 1e-2.  The exact-match rate is printed (the A/B figure for kernel changes that
    move near ties, e.g. the attention normaliser) and must not fall below POP_BEAM_MIN_EXACT."""
    vec, imgs = population
    exact, n = 0, 0
    for fam in ('noise', 'struct'):
        want, want_lp, gaps = vec['beam5_%s_ids' % fam], vec['beam5_%s_logprobs' % fam], vec['beam5_%s_margins' % fam]
        ids, lp = model.generate_beam(imgs[fam][:4].cuda(), 5)
        got, got_lp = ids.cpu().numpy(), lp.cpu().numpy()
        for b in range(4):
            same = bool(np.array_equal(got[b], want[b]))
            n += 1
            exact += int(same)
            if float(np.min(gaps[b])) > BEAM_MARGIN_FLOOR:
                assert same, (fam, b)
            assert same or abs(float(got_lp[b, 0]) - float(want_lp[b, 0])) < 1e-2, (fam, b, float(got_lp[b, 0]), float(want_lp[b, 0]))
    print('population beam 5: %d / %d results exactly equal to the reference' % (exact, n))
    assert exact / n >= POP_BEAM_MIN_EXACT


@pytest.mark.parametrize('step_i', [0, 1, 2, 3])
def test_per_step_logits_vs_reference(model, imgdep, step_i):
    """The [MASK]-row logits the reference's greedy loop hands to argmax at decode steps 1, 5, 10 and 19 (recorded while the
    reference ran) against the device's logits of the same step: a run with max_length = step + 1 ends at that step, and the
    engine keeps the last step's row.  Tolerance = the measured bf16 noise on a logit (1.8e-3 rms / 9e-3 max, DESIGN.md
    section 5) with a factor of two of head-room; the reference's 8 largest logits of every row are among the compared columns."""
    vec, cand = imgdep
    step = int(vec['step_list'][step_i])
    img = cand[torch.from_numpy(vec['sel_index'])].cuda()
    opts = model.gen_options(max_length=step + 1)
    ids, _ = model.run(img, opts)
    torch.cuda.synchronize()
    # the prefix decided so far must be the reference's (else the rows are not comparable)
    assert np.array_equal(ids.cpu().numpy()[:, 0, :step], vec['greedy_ids'][:, 0, :step])
    logits = model.tap('logits_last', 4, (4, 30592), opts=opts).cpu().numpy()
    cols, want = vec['step_cols'][step_i], vec['step_logits'][step_i]
    got = np.take_along_axis(logits, cols.astype(np.int64), axis=1)
    err = got - want
    rms, mx = float(np.sqrt((err ** 2).mean())), float(np.abs(err).max())
    print('step %d logits vs reference: rms %.3e max %.3e (row std %.3f)' % (step, rms, mx, float(vec['step_logit_std'][step_i])))
    assert rms < 4e-3 and mx < 2e-2


def test_several_eos_ids_equal_reference(model, imgdep):
    """eos_token_ids = [102, a, b] (modeling_utils.py:862-871): a sequence finishes at ANY of the ids and the forced token at the
    last position is the first id.  Reference golden on the four structured images: two stop early at different ids, two run to
    the end; the early-exit counter and the PAD fill behind a finished row ride along.  Also: the sampling loop takes the list
    (same bookkeeping kernel family), beam search refuses it like the reference's own assert does."""
    vec, cand = imgdep
    eos = [int(x) for x in vec['multi_eos_ids_list']]
    img = cand[torch.from_numpy(vec['sel_index'])].cuda()
    ids, lp = model.generate(img, eos_token_ids=eos)
    want = vec['multi_eos_ids']
    lens = sorted(int((r != 0).sum()) for r in want[:, 0])
    assert lens[0] < lens[-1]
    rep = assert_tokens_match_reference(ids.cpu().numpy(), want, vec['multi_eos_margins'], GREEDY_MARGIN_FLOOR, min_full=3,
                                        what='several EOS ids', eos=eos)
    print('several EOS ids', eos, rep)
    full = [r[0] for r in rep if r[4]]
    np.testing.assert_allclose(lp.cpu().numpy()[full], vec['multi_eos_logprobs'][full], rtol=0, atol=1e-2)
    ids_off, _ = model.generate(img, eos_token_ids=eos, early_exit=False)
    assert torch.equal(ids_off, ids)
    ids_s, _ = model.generate_multi(img, 2, eos_token_ids=eos, seed=3, temperature=0.7)
    got = ids_s.cpu().numpy()[:, 0]
    for r in got:       # after the first EOS id of the list only PAD may follow
        hit = [k for k in range(1, 20) if int(r[k]) in eos]
        assert hit and all(int(t) == 0 for t in r[hit[0] + 1:])
    with pytest.raises(NotImplementedError, match='num_beams == 1'):
        model.generate_beam(img, 3, eos_token_ids=eos)


def test_image_dependent_beam5_equals_reference(model, imgdep):
    vec, cand = imgdep
    img = cand[torch.from_numpy(vec['beam_index'])].cuda()
    ids, lp = model.generate_beam(img, 5)
    got, want = ids.cpu().numpy(), vec['beam5_ids']
    for b in range(2):
        same = bool((got[b] == want[b]).all())
        print('imgdep beam5 image %d: ids equal %s, score %.5f vs %.5f (min gap %.2e)' % (b, same, float(lp[b, 0]), float(vec['beam5_logprobs'][b, 0]),
                                                                                       float(vec['beam5_margins'][b].min())))
        if ('imgdep_beam5', b) in BEAM_SCORE_ONLY_OK:
            assert same or abs(float(lp[b, 0]) - float(vec['beam5_logprobs'][b, 0])) < 1e-2
        else:
            assert same, 'imgdep beam5 image %d no longer matches the reference exactly: review, then list it in BEAM_SCORE_ONLY_OK' % b
        assert abs(float(lp[b, 0]) - float(vec['beam5_logprobs'][b, 0])) < 2e-2


def test_untied_flow_multi_token_caption(imgdep):
    """The notebook flow (untied LM head, tagemb=None) on a recipe whose vocabulary-bias sigma (0.25) lets the caption run for
    15-19 tokens (with sigma 1 the [SEP] bias wins at the second token): comparable prefixes of 4 and 13 decisions here."""
    from vitcap_amd.model import ImageCaptioning
    vec, cand = imgdep
    m = ImageCaptioning(tie_weights=False, tagemb=None).load_recipe(0, vbias_std=float(vec['untied_vbias_std'][0])).eval()
    m.pack('cuda')
    img = cand[torch.from_numpy(vec['sel_index'])][:2].cuda()
    ids, lp = m({'image': img, 'key': [0, 1]})
    rep = assert_tokens_match_reference(ids.cpu().numpy(), vec['untied_ids'], vec['untied_margins'], GREEDY_MARGIN_FLOOR, min_full=0,
                                        what='untied flow, sigma 0.25')
    print('untied sigma 0.25 (sequence, comparable decisions, whole, prefix ok, whole caption equal):', rep)
    assert sum(r[1] for r in rep) >= 8, 'too few comparable decisions to mean anything'
    if all(r[4] for r in rep):
        np.testing.assert_allclose(lp.cpu().numpy(), vec['untied_logprobs'], rtol=0, atol=1e-2)


def test_tag_slot_start_position(imgdep):
    """`od_labels_start_posid` above the generation length (the pipeline's own default is max_seq_a_length = 40, ..._bertemb.py:197,
    597): the position ids of the tag slots reach the model through bert.extra_embeddings only (tagemb != 'cls', branch B,
    modeling_bert.py:1484-1485) -- tests/golden/make_golden_tagpos.py shows the reference's tied flow identical for 20 and 40 and
    its untied flow different.  Device, tags visible, against the reference's own output at 40 and at 20; the option must reach
    the kernel (the two device runs differ) and must not move the tied flow."""
    from vitcap_amd.model import ImageCaptioning
    _, cand = imgdep
    vec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_tagpos.npz')))
    img = cand[torch.from_numpy(vec['sel_index'])].cuda()
    assert np.array_equal(vec['tied_pos20_ids'], vec['tied_pos40_ids']) and np.array_equal(vec['tied_pos20_logprobs'], vec['tied_pos40_logprobs'])
    assert not np.array_equal(vec['untied_pos20_logprobs'], vec['untied_pos40_logprobs'])
    m = ImageCaptioning(tie_weights=False, tagemb=None).load_recipe(0, vbias_std=float(vec['untied_vbias_std'][0])).eval()
    m.pack('cuda')
    got = {}
    for pos in (20, 40):
        assert m.gen_options(tag_visible=50, od_labels_start_posid=pos).tag_pos0 == pos
        ids, lp = m.generate(img, tag_visible=50, od_labels_start_posid=pos)
        rep = assert_tokens_match_reference(ids.cpu().numpy(), vec['untied_pos%d_ids' % pos], vec['untied_pos%d_margins' % pos],
                                            GREEDY_MARGIN_FLOOR, min_full=0, what='untied, tags visible, start position %d' % pos)
        print('start position', pos, rep, lp.flatten().tolist(), vec['untied_pos%d_logprobs' % pos].flatten().tolist())
        assert sum(r[1] for r in rep) >= 8 or all(r[4] for r in rep), 'too few comparable decisions to mean anything'
        same = np.array([r[4] for r in rep])
        if same.any():
            np.testing.assert_allclose(lp.cpu().numpy()[same], vec['untied_pos%d_logprobs' % pos][same], rtol=0, atol=1e-2)
        got[pos] = (ids.clone(), lp.clone())
    assert not torch.equal(got[20][1], got[40][1]), 'the start position never reached bert.extra_embeddings'
    mt = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    mt.pack('cuda')
    a = [t.clone() for t in mt.generate(img, tag_visible=50, od_labels_start_posid=20)]
    b = mt.generate(img, tag_visible=50, od_labels_start_posid=40)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "tagemb 'cls' uses encode_tag_to_embedding's literal 20"
    assert_tokens_match_reference(b[0].cpu().numpy(), vec['tied_pos40_ids'], vec['tied_pos40_margins'], GREEDY_MARGIN_FLOOR, min_full=0,
                                  what='tied, tags visible, start position 40')


@pytest.mark.parametrize('name,beams', [('beam2_b1', 2), ('beam5_b1', 5), ('beam5_b2', 5), ('beam5_sel', 5), ('beam3_alteos_b2', 3)])
def test_beam_search_vs_reference_goldens(model, golden, name, beams):
    """a13 against the reference's own beam output.  Beam decisions on random-init logits are ill-conditioned (the stored
    decision gaps are 1e-5..1e-3 against a bf16 floor of 3e-2), so: identical ids where every gap clears the floor, else
    identical ids OR a length-normalised score within 1e-2 of the reference's (a near-tie resolved the other way)."""
    vec, _ = golden
    want, want_lp, gaps = vec[name + '_ids'], vec[name + '_logprobs'], vec[name + '_margins']
    B = want.shape[0]
    img = _selected(vec, 'beam_sel_index') if name == 'beam5_sel' else _images(B)
    kw = {'eos_token_ids': [int(vec['alt_eos_id'][0])]} if 'alteos' in name else {}
    ids, lp = model.generate_beam(img.cuda(), beams, **kw)
    got, got_lp = ids.cpu().numpy(), lp.cpu().numpy()
    for b in range(B):
        same = bool((got[b] == want[b]).all())
        print('%s image %d: min decision gap %.2e, ids equal: %s, score %.5f vs %.5f' % (name, b, float(gaps[b].min()), same,
                                                                                         float(got_lp[b, 0]), float(want_lp[b, 0])))
        if float(gaps[b].min()) >= BEAM_MARGIN_FLOOR:
            assert same, 'beam result differs from the reference although every decision gap clears the floor'
        # Expected state, recorded: EVERY beam golden is reproduced token for token.  None of them is gap-comparable (random-init
        # logits leave decision gaps of 1e-5..1e-3), so a kernel change may legitimately flip one -- then the case goes on the
        # reviewed allow-list below with its score evidence (|score - reference| < 1e-2), instead of passing silently.
        if (name, b) in BEAM_SCORE_ONLY_OK:
            assert same or abs(float(got_lp[b, 0]) - float(want_lp[b, 0])) < 1e-2
        else:
            assert same, ('beam golden %s image %d no longer matches the reference exactly (score %.5f vs %.5f, min decision gap %.2e): '
                          'review, then list it in BEAM_SCORE_ONLY_OK' % (name, b, float(got_lp[b, 0]), float(want_lp[b, 0]), float(gaps[b].min())))
        assert abs(float(got_lp[b, 0]) - float(want_lp[b, 0])) < 2e-2


def test_device_floats_vs_reference_goldens(model, golden):
    """Device FLOATS against the reference's own fp32 floats (tests/golden, written by running /root/reference), not through the
    oracle's bf16 emulation: the encoder's hidden rows (a5), the tag head's logits (a6) and the whole 30522-wide logits row of
    decode step 1 (`encode_forward`, modeling_bert.py:751-807).  bf16 operands / fp32 accumulation against fp32 everywhere:
    the stated tolerances are the measured activation-rounding noise (1.8e-3 rms / 9e-3 max on a logit,
    5.6e-3 relative L2 on the encoder output) with a factor of two of head-room."""
    vec, _ = golden
    img = _images(2)
    # one decode step: max_length = 2 makes step 1 the last (and only) step, whose logits row the engine keeps
    opts = model.gen_options(max_length=2)
    model.run(img.cuda(), opts)
    torch.cuda.synchronize()
    logits = model.tap('logits_last', 2, (2, 30592), opts=opts).cpu()[0, :30522].numpy()
    want = vec['step1_logits_row']
    err = logits - want
    print('step-1 logits vs reference: rms %.3e max %.3e (logit std %.3f)' % (float(np.sqrt((err ** 2).mean())), float(np.abs(err).max()), float(want.std())))
    assert float(np.sqrt((err ** 2).mean())) < 4e-3 and float(np.abs(err).max()) < 2e-2
    assert int(np.argmax(logits)) == int(np.argmax(want)) or np.sort(want)[-1] - np.sort(want)[-2] < GREEDY_MARGIN_FLOOR
    hid = model.tap('hidden', 2, (2, 577, 768), opts=opts).cpu()[:, :2].numpy()
    rel = float(np.linalg.norm(hid - vec['a5_hidden_rows']) / np.linalg.norm(vec['a5_hidden_rows']))
    print('hidden[:, :2] vs reference: rel L2 %.3e' % rel)
    assert rel < 1.2e-2
    tagc = model.tap('tag_hidden', 2, (2, 577, 768), opts=opts).cpu()[:, 0].numpy()
    rel_t = float(np.linalg.norm(tagc - vec['a5_tag_hidden_cls']) / np.linalg.norm(vec['a5_tag_hidden_cls']))
    print('tag_hidden[:, 0] vs reference: rel L2 %.3e' % rel_t)
    assert rel_t < 1.2e-2
    model.generate(img.cuda(), want_tags=True)
    tl = model.last_tags[0].cpu().numpy()[:, :64]
    terr = float(np.abs(tl - vec['a6_logit_head']).max())
    print('tag logits[:, :64] vs reference: max abs %.3e' % terr)
    assert terr < 2e-2


def test_fp32_checkpoint_weights_rounded_by_pack(sd_np):
    """Every recipe matrix is bf16-exact, so the parity tests above never exercise the rounding `pack()` applies to a REAL fp32
    checkpoint.  Here the matrices keep their fp32 values (`bf16_exact=False`): the device rounds them to bf16, the fp32 oracle
    (token-exact with the reference, tests/test_oracle_golden.py) uses them as they are.  Weight rounding adds to the activation
    rounding: measured 3.0e-3 rms on a logit against 1.8e-3 with identical weights; asserted with head-room,
    and tokens must agree on every decision whose fp32 margin clears a floor scaled by the same ratio (0.02)."""
    from oracle import vitcap_oracle as O
    from vitcap_amd import weights as W
    from vitcap_amd.model import ImageCaptioning
    sd_f = W.make_state_dict(seed=0, tie_weights=True, bf16_exact=False)
    k = 'module.bert.encoder.blocks.0.attn.qkv.weight'
    assert not np.array_equal(sd_f[k], sd_np[k]) and np.abs(sd_f[k] - sd_np[k]).max() < 2 ** -8 * np.abs(sd_np[k]).max() * 1.01
    m = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0, bf16_exact=False).eval()
    m.pack('cuda')
    img = _images(2)
    ids, lp = m({'image': img.cuda(), 'key': [0, 1]})
    torch.cuda.synchronize()
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        ids_o, lp_o, tr = O.greedy_incremental(O.to_torch(sd_f), img, emulate_bf16=False, return_trace=True)
    hid = m.tap('hidden', 2, (2, 577, 768)).cpu()
    rel = float((hid - tr['hidden']).norm() / tr['hidden'].norm())
    print('fp32-weights run: hidden rel L2 vs fp32 oracle %.3e' % rel)
    assert rel < 1.5e-2
    margins = torch.stack([s['margin'] for s in tr['steps']], 1)
    rep = assert_tokens_match_reference(ids.cpu().numpy(), ids_o.numpy(), margins.numpy(), 0.02, min_full=0, what='fp32 checkpoint weights')
    print('fp32-weights run (sequence, comparable decisions, whole, prefix ok, whole caption equal):', rep)
    assert sum(r[1] for r in rep) >= 6
    if all(r[4] for r in rep):
        logits = m.tap('logits_last', 2, (2, 30592)).cpu()[:, :30522]
        err = logits - tr['steps'][-1]['logits_row']
        rms = float(err.pow(2).mean().sqrt())
        print('fp32-weights run: last-step logits rms %.3e max %.3e' % (rms, float(err.abs().max())))
        assert rms < 6e-3 and float(err.abs().max()) < 3e-2
        np.testing.assert_allclose(lp.cpu().numpy(), lp_o.numpy(), rtol=0, atol=1.5e-2)


# ------------------------------------------------------------------------------------------------ vs the bf16 emulation
def test_engine_vs_oracle_emulation(model, oracle_run):
    img, ids_o, lp_o, tr = oracle_run
    B = img.shape[0]
    ids, lp = model({'image': img.cuda(), 'key': list(range(B))})
    torch.cuda.synchronize()
    assert ids.shape == (B, 1, 20) and ids.dtype == torch.int64 and lp.shape == (B, 1)
    # encoder taps
    hid = model.tap('hidden', B, (B, 577, 768)).cpu()
    # of the tag branch's output only the CLS row is ever read (pooler input and first visual token, modeling_bert.py:1424,
    # 1493): the engine computes the last tag block for that row alone, so that row is what is compared
    tag = model.tap('tag_hidden', B, (B, 577, 768)).cpu()[:, :1]
    for name, got, want in (('hidden', hid, tr['hidden']), ('tag_hidden[:, 0]', tag, tr['tag_hidden'][:, :1])):
        rel = float((got - want).norm() / want.norm())
        print('%s rel L2 err vs emulation: %.3e' % (name, rel))
        assert rel < 2e-2, name
    margins = torch.stack([s['margin'] for s in tr['steps']], 1)        # (B,19)
    print('oracle margins min %.4f median %.4f' % (float(margins.min()), float(margins.median())))
    rep = assert_tokens_match_reference(ids.cpu().numpy(), ids_o.numpy(), margins.numpy(), MARGIN_TOL, min_full=0,
                                        what='device vs bf16 emulation')
    print('device vs emulation (sequence, comparable decisions, whole, prefix ok, whole caption equal):', rep)
    same = torch.tensor([r[4] for r in rep])
    np.testing.assert_allclose(lp.cpu().numpy(), lp_o.numpy(), rtol=0, atol=1e-2)
    if bool(same.all()):
        logits = model.tap('logits_last', B, (B, 30592)).cpu()[:, :30522]
        want = tr['steps'][-1]['logits_row']
        err = float((logits - want).abs().max())
        print('last-step logits max abs err %.3e (logit std %.3f)' % (err, float(want.std())))
        assert err < 2e-2


def test_tag_head(model, oracle_run):
    img, _, _, tr = oracle_run
    B = img.shape[0]
    model.generate(img.cuda(), want_tags=True)
    logits, topk = model.last_tags
    o_logit, o_prob, o_pred, o_len = tr['tags']
    err = float((logits.cpu() - o_logit).abs().max())
    print('tag logits max abs err %.3e' % err)
    assert err < 2e-2
    # top-50 as a set, allowing swaps among candidates whose probabilities differ by < 1e-3
    for b in range(B):
        got, want = set(topk[b].cpu().tolist()), set(o_pred[b].tolist())
        diff = got ^ want
        p = torch.sigmoid(o_logit[b])
        assert all(abs(float(p[i]) - float(o_prob[b, -1])) < 1e-3 for i in diff), (b, diff)
    assert torch.equal(model.tap('tag_len', B, (B,), torch.int64).cpu(), o_len)


@pytest.mark.parametrize('beams,keep', [(2, 1), (5, 1), (3, 3)])
def test_beam_search_vs_oracle(model, sd_t, beams, keep):
    """a13 (and a11 num_keep_best): device beam search == the oracle's driver on the bf16-emulated incremental model, for every
    image whose smallest decision gap in the oracle run exceeds the fp32-summation-order floor; all kept hypotheses, in order."""
    from oracle import vitcap_oracle as O
    B = 3
    img = _images(B)
    with torch.no_grad():
        ids_o, lp_o, gaps = O.beam_incremental(sd_t, img, num_beams=beams, emulate_bf16=True, num_keep_best=keep, return_margins=True)
    model.test_extra_input.update(num_beams=beams, num_keep_best=keep)
    try:
        ids, lp = model({'image': img.cuda(), 'key': list(range(B))})
    finally:
        model.test_extra_input.update(num_beams=1, num_keep_best=1)
    torch.cuda.synchronize()
    ids, lp = ids.cpu(), lp.cpu()
    assert ids.shape == (B, keep, 20) and lp.shape == (B, keep)
    assert bool((lp[:, :-1] >= lp[:, 1:]).all())
    ok = gaps.min(1).values > BEAM_GAP_TOL
    same = (ids == ids_o).all(-1).all(-1)
    print('min decision gaps', gaps.min(1).values.tolist(), 'hip', lp.tolist(), 'oracle', lp_o.tolist(), 'ids equal', same.tolist())
    # beam decisions on random-init logits are ill-conditioned (gaps 1e-5..1e-3 against a 3e-2 floor): identical ids are demanded
    # where the gaps clear the floor; everywhere the kept scores must agree (a near-tie resolved the other way moves them by
    # less than the floor).  The bookkeeping itself is pinned exactly on synthetic logits by test_hip_beam_bookkeeping.py.
    assert bool(same[ok].all()), 'beam result differs from the emulation on an image whose decision gaps clear the floor'
    np.testing.assert_allclose(lp.numpy(), lp_o.numpy(), atol=1e-2)
    if keep > 1:
        ids1, lp1 = model.generate_beam(img.cuda(), beams)
        assert torch.equal(ids[:, :1], ids1.cpu()) and torch.equal(lp[:, :1], lp1.cpu())


@pytest.mark.parametrize('beams,rp', [(1, 1.3), (3, 1.3), (1, 0.8)])
def test_repetition_penalty_vs_oracle(model, sd_t, beams, rp):
    """generate(repetition_penalty=rp) through ImageCaptioning.forward, greedy and beam, against the bf16-emulating oracle
    (itself pinned to the reference by tests/test_oracle_golden.py); and the penalty really changes the caption."""
    from oracle import vitcap_oracle as O
    B = 3
    img = _images(B)
    with torch.no_grad():
        if beams == 1:
            ids_o, lp_o, tr = O.greedy_incremental(sd_t, img, emulate_bf16=True, repetition_penalty=rp, return_trace=True)
            ok = torch.stack([s['margin'] for s in tr['steps']], 1).min(1).values > MARGIN_TOL
        else:
            ids_o, lp_o, gaps = O.beam_incremental(sd_t, img, num_beams=beams, emulate_bf16=True, repetition_penalty=rp,
                                                   return_margins=True)
            ok = gaps.min(1).values > BEAM_GAP_TOL
    plain, _ = model({'image': img.cuda(), 'key': [0, 1, 2]})
    plain = plain.clone()
    model.test_extra_input.update(num_beams=beams, repetition_penalty=rp)
    try:
        ids, lp = model({'image': img.cuda(), 'key': [0, 1, 2]})
        ids, lp = ids.cpu(), lp.cpu()
    finally:
        model.test_extra_input.update(num_beams=1, repetition_penalty=1)
    again, _ = model({'image': img.cuda(), 'key': [0, 1, 2]})
    assert torch.equal(again, plain), 'options are per call: repetition_penalty=1 restores the plain caption'
    same = (ids == ids_o).all(-1).all(-1)
    print('hip', ids[:, 0].tolist(), lp.flatten().tolist(), 'oracle', lp_o.flatten().tolist(), 'comparable', ok.tolist(), 'equal', same.tolist())
    if beams == 1:
        assert_tokens_match_reference(ids.numpy(), ids_o.numpy(), torch.stack([s['margin'] for s in tr['steps']], 1).numpy(),
                                      MARGIN_TOL, min_full=0, what='repetition penalty %.1f' % rp)
    assert bool(same[ok].all())
    np.testing.assert_allclose(lp.numpy(), lp_o.numpy(), atol=1e-2)
    if beams == 1:
        assert not torch.equal(ids, plain.cpu())
        if rp > 1:       # a penalised greedy caption repeats fewer tokens than the plain one
            rep = lambda t: sum(len(r) - len(set(r)) for r in t[:, 0].tolist())
            assert rep(ids) < rep(plain.cpu())


@pytest.mark.parametrize('max_length', [8, 33])
def test_max_length_other_than_20(model, sd_t, max_length):
    """max_length is a generate() kwarg (modeling_bert.py:928-933; max_gen_length in the YAML): 8 and 33 tokens against the
    oracle's bf16 emulation, greedy and beam; output rows are max_length wide."""
    from oracle import vitcap_oracle as O
    B = 2
    img = _images(B)
    with torch.no_grad():
        ids_o, lp_o, tr = O.greedy_incremental(sd_t, img, emulate_bf16=True, max_length=max_length, return_trace=True)
    ids, lp = model.generate(img.cuda(), max_length=max_length)
    assert ids.shape == (B, 1, max_length)
    margins = torch.stack([s['margin'] for s in tr['steps']], 1)
    assert_tokens_match_reference(ids.cpu().numpy(), ids_o.numpy(), margins.numpy(), MARGIN_TOL, min_full=0, what='max_length=%d' % max_length)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_o.numpy(), atol=1e-2)
    with torch.no_grad():
        bi_o, bl_o, gaps = O.beam_incremental(sd_t, img, num_beams=2, emulate_bf16=True, max_length=max_length, return_margins=True)
    bi, bl = model.generate_beam(img.cuda(), 2, max_length=max_length)
    assert bi.shape == (B, 1, max_length)
    ok = gaps.min(1).values > BEAM_GAP_TOL
    assert bool(((bi.cpu() == bi_o).all(-1).all(-1))[ok].all())
    np.testing.assert_allclose(bl.cpu().numpy(), bl_o.numpy(), atol=1e-2)
    # the default length is untouched by the per-call override
    assert model.generate(img.cuda())[0].shape == (B, 1, 20)


COMBOS = [
    # generate() kwargs in combination (each is also tested alone above): the decode loop's branches interact -- penalty before
    # filter before draw, EOS bookkeeping at other lengths, length penalty with n-best lists, the cap of 40 tokens
    dict(max_length=12, eos=18218, repetition_penalty=1.2),
    dict(max_length=40, repetition_penalty=0.9),
    dict(max_length=25, eos=18218, repetition_penalty=1.1, sample=dict(temperature=0.8, top_k=30, top_p=0.9, seed=21)),
    dict(num_beams=3, num_keep_best=2, length_penalty=0.7, repetition_penalty=1.3, max_length=15),
    dict(num_beams=4, num_keep_best=4, length_penalty=1.4, eos=18218, max_length=24),
    dict(num_beams=2, num_keep_best=2, max_length=16, repetition_penalty=1.2, sample=dict(temperature=1.1, top_k=0, top_p=0.8, seed=22)),
]


@pytest.mark.parametrize('combo', range(len(COMBOS)))
def test_option_combinations_vs_oracle(model, sd_t, combo):
    """Differential test of COMBINED generate() options against the oracle's incremental model (bf16 emulation; the oracle's
    option handling is pinned to the reference one option at a time by tests/test_oracle_golden.py / test_oracle_sample.py).
    An image is compared when the oracle's own fp32 and bf16-emulated runs give the same tokens (its decisions then clear the
    rounding noise without reference to the device); its device tokens must be identical, log-probs within 1e-2."""
    from oracle import vitcap_oracle as O
    c = dict(COMBOS[combo])
    B = 3
    img = _images(B, seed=4000 + combo)
    K, keep = c.get('num_beams', 1), c.get('num_keep_best', 1)
    L, eos, rp, lpen = c.get('max_length', 20), c.get('eos', 102), c.get('repetition_penalty', 1.0), c.get('length_penalty', 1.0)
    smp = c.get('sample')

    steps = {}

    def run_oracle(emulate):
        """-> ids (B, keep, L), log-probs (B, keep), smallest decision margin per image (inf where the oracle does not track it)"""
        with torch.no_grad():
            if K == 1:
                sampler = O.make_sampler(smp['temperature'], smp['top_k'], smp['top_p'], smp['seed']) if smp else None
                ids, lp, tr = O.greedy_incremental(sd_t, img, emulate_bf16=emulate, max_length=L, sampler=sampler,
                                                   repetition_penalty=rp, eos=eos, return_trace=True)
                steps[emulate] = torch.stack([st['margin'] for st in tr['steps']], 1)
                mg = torch.tensor([min(relevant_margins(ids[b].reshape(-1).tolist(), steps[emulate][b].tolist(), eos)) for b in range(B)])
                return ids.view(B, 1, L), lp.view(B, 1), mg
            if smp:
                ids, lp = O.beam_incremental(sd_t, img, num_beams=K, emulate_bf16=emulate, max_length=L, length_penalty=lpen,
                                             num_keep_best=keep, repetition_penalty=rp, eos=eos, sample=dict(smp))
                return ids, lp, torch.full((B,), float('inf'))
            ids, lp, gaps = O.beam_incremental(sd_t, img, num_beams=K, emulate_bf16=emulate, max_length=L, length_penalty=lpen,
                                               num_keep_best=keep, repetition_penalty=rp, eos=eos, return_margins=True)
            return ids, lp, gaps.min(1).values
    ids_e, lp_e, mg_e = run_oracle(True)
    ids_f, _, mg_f = run_oracle(False)
    # comparable: the oracle's fp32 and bf16-emulated runs agree AND (where the oracle tracks decision margins: greedy argmax /
    # Gumbel-perturbed draw, beam bookkeeping) every decision clears the noise floor in both
    floor = MARGIN_TOL if K == 1 else BEAM_GAP_TOL
    ok = (ids_e == ids_f).flatten(1).all(1) & (mg_e > floor) & (mg_f > floor)
    kw = dict(num_beams=K, num_keep_best=keep, max_length=L, eos_token_ids=[eos], repetition_penalty=rp, length_penalty=lpen,
              do_sample=bool(smp), num_return_sequences=1)
    if smp:
        kw.update(smp)
    ids, lp = model.run(img.cuda(), model.gen_options(**kw))
    ids, lp = ids.cpu(), lp.cpu()
    print('combo', c, 'comparable', ok.tolist(), 'equal', (ids == ids_e).flatten(1).all(1).tolist())
    assert ids.shape == (B, keep, L) and lp.shape == (B, keep)
    print('margins', mg_e.tolist(), 'hip lp', lp.tolist(), 'oracle lp', lp_e.tolist())
    if K == 1:        # token by token: identical up to each caption's first decision below the floor (in the emulation's own margins)
        assert_tokens_match_reference(ids.numpy(), ids_e.numpy(), steps[True].numpy(), MARGIN_TOL, min_full=0, what=str(c), eos=eos)
    if smp:
        assert int(ok.sum()) >= 1, 'no image of this combination is well conditioned: pick another seed'
    assert torch.equal(ids[ok], ids_e[ok]), (ids[ok].tolist(), ids_e[ok].tolist())
    # every image, comparable or not: the kept scores agree (a near-tie resolved the other way moves them by less than the floor)
    fin = lp_e > -1e4
    assert bool(((lp > -1e4) == fin).all())
    np.testing.assert_allclose(lp[fin].numpy(), lp_e[fin].numpy(), atol=3e-2 if K > 1 else 1e-2)
    # well-formedness everywhere: starts with [CLS]; after the first EOS only padding
    for row in ids.flatten(0, 1).tolist():
        assert row[0] == 101
        if eos in row[1:]:
            assert all(t == 0 for t in row[row.index(eos, 1) + 1:])


# ------------------------------------------------------------------------------------------------ properties / behaviour
def test_batch64_properties(model):
    """Size-independent properties at the benchmark batch size."""
    B = 64
    img = _images(B).cuda().to(torch.bfloat16)
    ids1, lp1 = model.generate(img)
    ids1, lp1 = ids1.clone(), lp1.clone()
    ids2, lp2 = model.generate(img)
    assert torch.equal(ids1, ids2) and torch.equal(lp1, lp2), 'non-deterministic'
    ids_small, lp_small = model.generate(img[:4].contiguous())
    assert torch.equal(ids_small, ids1[:4]), 'batch composition changed a caption'
    np.testing.assert_allclose(lp_small.cpu().numpy(), lp1[:4].cpu().numpy(), atol=1e-6)
    i = ids1.cpu()[:, 0]
    assert (i[:, 0] == 101).all() and ((i >= 0) & (i < 30522)).all()
    # after the first EOS only PAD may follow
    for row in i.tolist():
        if 102 in row:
            k = row.index(102)
            assert all(v == 0 for v in row[k + 1:])
    assert torch.isfinite(lp1).all()


def test_batch64_vs_oracle(model, sd_t):
    """BASELINE configs[1]'s batch (64 images) compared with the ORACLE, not only with itself: the oracle's
    incremental fp32 model (token-exact with the reference on every golden, tests/test_oracle_golden.py) captions the same 64 images on
    the host; every device caption equals the oracle's up to its first decision below the bf16 noise floor, whole captions wherever
    every decision clears it, log-probs of the equal captions within 1e-2."""
    from oracle import vitcap_oracle as O
    B = 64
    img = _images(B, seed=6400)
    with torch.no_grad():
        ids_o, lp_o, tr = O.greedy_incremental(sd_t, img, emulate_bf16=False, return_trace=True)
    margins = torch.stack([st['margin'] for st in tr['steps']], 1).numpy()
    ids, lp = model.generate(img.cuda().to(torch.bfloat16))
    got, want = ids.cpu().numpy(), ids_o.view(B, 1, -1).numpy()
    rep = assert_tokens_match_reference(got, want, margins, GREEDY_MARGIN_FLOOR, min_full=8, what='B = 64 vs oracle')
    equal = [r[0] for r in rep if r[4]]
    print('B = 64: %d / %d captions equal the oracle\'s, %d whole-caption comparable' % (len(equal), B, sum(1 for r in rep if r[2])))
    np.testing.assert_allclose(lp.cpu().numpy()[equal, 0], lp_o.view(B, -1).numpy()[equal, 0], rtol=0, atol=1e-2)
    assert len(equal) >= int(0.8 * B)


def test_early_exit_and_finished_rows(model, golden):
    """`if cur_unfinished.max() == 0: break` (modeling_utils.py:866) on the device: with a frequent token as EOS and a batch in
    which every caption ends early, the live counter reaches 0, the remaining steps' kernels return at entry, and ids / scores
    equal the run with early_exit off bit for bit; mixed batches (some rows finished, some not) are covered by the
    alternative-EOS golden above."""
    vec, _ = golden
    img = _images(6).cuda()
    full, _ = model.generate(img)
    full = full.cpu()[:, 0]
    # a token every caption contains (position 1..6) ends all of them early
    common = [t for t in full[0, 1:8].tolist() if all(t in r[1:8].tolist() for r in full)]
    assert common, full.tolist()
    eos = common[0]
    a_ids, a_lp = [t.clone() for t in model.generate(img, eos_token_ids=[eos], early_exit=True)]
    live = int(model.tap('live', 6, (1,), torch.int32)[0])
    b_ids, b_lp = [t.clone() for t in model.generate(img, eos_token_ids=[eos], early_exit=False)]
    assert live == 0, 'every sequence finished, the live counter must have reached 0'
    assert torch.equal(a_ids, b_ids) and torch.equal(a_lp, b_lp)
    rows = a_ids.cpu()[:, 0].tolist()
    for r in rows:
        k = r.index(eos)
        assert k < 8 and all(v == 0 for v in r[k + 1:])
    # not finished -> counter stays positive with the default EOS
    model.generate(img)
    assert int(model.tap('live', 6, (1,), torch.int32)[0]) == 6
    # beam search: `if all(done): break` (modeling_utils.py:1072)
    c_ids, c_lp = [t.clone() for t in model.generate_beam(img, 3, eos_token_ids=[eos], early_exit=True)]
    d_ids, d_lp = model.generate_beam(img, 3, eos_token_ids=[eos], early_exit=False)
    assert torch.equal(c_ids, d_ids) and torch.equal(c_lp, d_lp)


def test_decode_loop_hipgraph_replay(model):
    """vitcap_gen_opts.use_graph: the decode loop is captured once per (batch, workspace, options) and replayed; ids and scores
    are identical to eager launches, for greedy and beam search (BASELINE configs[2] asks for a graph-captured beam step)."""
    img = _images(5, 77).cuda().to(torch.bfloat16)
    img2 = _images(5, 78).cuda().to(torch.bfloat16)
    n0 = model.graph_count()
    for kind in ('greedy', 'beam'):
        run = (lambda im, **kw: model.generate(im, **kw)) if kind == 'greedy' else (lambda im, **kw: model.generate_beam(im, 4, **kw))
        e1 = [t.clone() for t in run(img)]
        e2 = [t.clone() for t in run(img2)]
        g1 = [t.clone() for t in run(img, use_graph=True)]        # captures
        g2 = [t.clone() for t in run(img2, use_graph=True)]       # replays on another input
        g3 = [t.clone() for t in run(img, use_graph=True)]        # replays again
        for a, b in ((e1, g1), (e2, g2), (e1, g3)):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), kind
    assert model.graph_count() == n0 + 2, 'one graph per (batch, workspace, options), reused by later calls'


def test_split_decode_loop_equals_single_chain(model):
    """vitcap_gen_opts.decode_streams: the greedy / sampling loop cut into two slices on two streams gives bit-identical ids,
    scores and last tokens -- eager and graph-replayed, even and odd batch sizes, several sequences per image."""
    for B in (16, 33):
        img = _images(B, 900 + B).cuda().to(torch.bfloat16)
        one = [t.clone() for t in model.generate(img, decode_streams=1)]
        two = [t.clone() for t in model.generate(img, decode_streams=2)]
        auto = [t.clone() for t in model.generate(img)]
        gr = [t.clone() for t in model.generate(img, decode_streams=2, use_graph=True)]
        gr2 = [t.clone() for t in model.generate(img, decode_streams=2, use_graph=True)]
        for other in (two, auto, gr, gr2):
            assert torch.equal(one[0], other[0]) and torch.equal(one[1], other[1]), B
    img = _images(6, 77).cuda().to(torch.bfloat16)
    samp = dict(temperature=0.9, top_k=50, top_p=0.9, seed=21)
    a = [t.clone() for t in model.generate_multi(img, 3, want_last=True, decode_streams=1, **samp)]
    b = [t.clone() for t in model.generate_multi(img, 3, want_last=True, decode_streams=2, **samp)]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # an alternative EOS ends sequences early in both slices; the shared live counter still reaches 0
    full = model.generate(img)[0].cpu()[:, 0]
    common = [t for t in full[0, 1:8].tolist() if all(t in r[1:8].tolist() for r in full)]
    c = [t.clone() for t in model.generate(img, eos_token_ids=[common[0]], decode_streams=1)]
    d = [t.clone() for t in model.generate(img, eos_token_ids=[common[0]], decode_streams=2)]
    assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])
    assert int(model.tap('live', 6, (1,), torch.int32)[0]) == 0


@pytest.mark.parametrize('flow', ['cls', 'untied'])
def test_tag_tokens_visible_to_caption(golden, flow):
    """with the mask tensorize_ab builds for a text_b of n tag tokens (dataset.py:240-252, 387-390) every
    caption row attends the first n predicted tag tokens, whose embeddings follow the two branches of modeling_bert.py:1435-1489
    (re-selected at every step by `topk_len[0] + 20 <= L`).  Device captions against the reference's own output for n = 50 and
    n = 7 (pipeline flow, tagemb 'cls') and n = 50 in the notebook flow (tagemb None: bert.extra_embeddings / word embeddings),
    driven through ImageCaptioning.forward with the caller's attention_mask; the tags really change the captions."""
    from oracle import vitcap_oracle as O
    from vitcap_amd.model import ImageCaptioning
    vec, _ = golden
    if flow == 'cls':
        m = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
        cases = [('greedy_tags50_b2', 50, _images(2), 0), ('greedy_tags7_b2', 7, _images(2), 0), ('greedy_tags50_sel', 50, _selected(vec), 0)]
        plain = vec['greedy_b2_ids']
    else:
        m = ImageCaptioning(tie_weights=False, tagemb=None).load_recipe(0).eval()
        cases = [('greedy_untied_tags50_b2', 50, _images(2), 0)]
        plain = vec['greedy_untied_nocls_b2_ids']
    m.pack('cuda')
    key0 = cases[0][0]
    plain_lp = vec['greedy_b2_logprobs' if flow == 'cls' else 'greedy_untied_nocls_b2_logprobs']
    assert not np.array_equal(vec[key0 + '_ids'], plain) or float(np.abs(vec[key0 + '_logprobs'] - plain_lp).max()) > 5e-3, \
        'golden: visible tags were meant to change the caption or at least its score'
    for name, n, img, min_full in cases:
        B = img.shape[0]
        input_ids, am = O.test_text_inputs(B, n_tag_visible=n)
        data = {'image': img.cuda(), 'key': list(range(B)), 'input_ids': input_ids.cuda(), 'attention_mask': am.cuda(),
                'token_type_ids': torch.zeros(B, 70, dtype=torch.long).cuda(), 'masked_pos': torch.ones(B, 70, dtype=torch.int32).cuda()}
        ids, lp = m(data)
        rep = assert_tokens_match_reference(ids.cpu().numpy(), vec[name + '_ids'], vec[name + '_margins'], GREEDY_MARGIN_FLOOR,
                                            min_full=min_full, what=name)
        print(name, rep, lp.flatten().tolist(), vec[name + '_logprobs'].flatten().tolist())
        same = np.array([r[4] for r in rep])          # a sequence that left the reference's path at a sub-floor decision has another score
        assert same.any()
        np.testing.assert_allclose(lp.cpu().numpy()[same], vec[name + '_logprobs'][same], rtol=0, atol=1e-2)
        # the same through the option instead of the mask, and a beam search over the same keys runs and is well formed
        ids2, _ = m.generate(img.cuda(), tag_visible=n)
        assert torch.equal(ids, ids2)
    bi, bl = m.generate_beam(cases[0][2].cuda(), 3, tag_visible=50)
    assert bi.shape == (cases[0][2].shape[0], 1, 20) and torch.isfinite(bl).all() and (bi[:, 0, 0] == 101).all()
    # the shipped mask is untouched by the option's existence
    ids0, _ = m.generate(_images(2).cuda())
    assert_tokens_match_reference(ids0.cpu().numpy(), plain, vec['greedy_b2_margins' if flow == 'cls' else 'greedy_untied_nocls_b2_margins'],
                                  GREEDY_MARGIN_FLOOR, min_full=0, what='plain after tags')


def test_encoder_parts_equal_single_chain(model):
    """vitcap_gen_opts.encode_parts: encoder + prefill of the batch as 2..4 independent chains of batch parts on separate streams
    give bit-identical captions, scores, tag outputs and hidden states (even and odd batch sizes, greedy and beam)."""
    for B in (9, 64):
        img = _images(B, 1300 + B).cuda().to(torch.bfloat16)
        one = [t.clone() for t in model.generate(img, encode_parts=1, want_tags=True)]
        tags1 = [t.clone() for t in model.last_tags]
        hid1 = model.tap('hidden', B, (B, 577, 768))
        for parts in (2, 3, 4):
            got = [t.clone() for t in model.generate(img, encode_parts=parts, want_tags=True)]
            assert torch.equal(one[0], got[0]) and torch.equal(one[1], got[1]), (B, parts)
            assert torch.equal(tags1[0], model.last_tags[0]) and torch.equal(tags1[1], model.last_tags[1])
            assert torch.equal(hid1, model.tap('hidden', B, (B, 577, 768)))
        b1 = [t.clone() for t in model.generate_beam(img[:9].contiguous(), 3, encode_parts=1)]
        b2 = model.generate_beam(img[:9].contiguous(), 3, encode_parts=2)
        assert torch.equal(b1[0], b2[0]) and torch.equal(b1[1], b2[1])


def test_text_inputs_are_validated(model):
    """a8 / a16: forward() checks the caller's text tensors against the mask structure the kernels implement."""
    from oracle import vitcap_oracle as O
    B = 2
    img = _images(B).cuda()
    input_ids, am = O.test_text_inputs(B)                  # what CaptionTensorizer emits at test time (dataset.py:218-219, 377-390)
    data = {'image': img, 'key': [0, 1], 'input_ids': input_ids.cuda(), 'attention_mask': am.cuda(),
            'token_type_ids': torch.zeros(B, 70, dtype=torch.long).cuda(), 'masked_pos': torch.ones(B, 70, dtype=torch.int32).cuda()}
    ids, _ = model(data)
    ref, _ = model({'image': img, 'key': [0, 1]})
    assert torch.equal(ids, ref)
    bad = dict(data)
    m2 = am.clone()
    m2[:, :20, 20:] = 1                                    # caption rows attending the tag slots: not the shipped pattern
    bad['attention_mask'] = m2.cuda()
    with pytest.raises(NotImplementedError, match='attention_mask'):
        model(bad)
    bad = dict(data)
    m3 = am.clone()
    m3[1, 3, 5] = 1                                        # one extra visible position in one sample
    bad['attention_mask'] = m3.cuda()
    with pytest.raises(NotImplementedError, match='attention_mask'):
        model(bad)
    bad = dict(data)
    bad['token_type_ids'] = torch.ones(B, 70, dtype=torch.long).cuda()
    with pytest.raises(NotImplementedError, match='token_type_ids'):
        model(bad)
    bad = dict(data)
    bad['attention_mask'] = am[:1].cuda()
    with pytest.raises(ValueError, match='batch'):
        model(bad)


def test_two_host_threads_share_one_engine(model):
    """Options are per call and the engine serialises its enqueues: one thread running generate() (persistent GEMMs, its own
    stream and workspace slot) while another drives the two-slot pipeline (one tile per workgroup) gets the results each gets
    alone."""
    img_a = _images(6, 300).cuda().to(torch.bfloat16)
    img_b = _images(7, 301).cuda().to(torch.bfloat16)
    want_a = [t.clone() for t in model.generate(img_a)]
    want_b = [t.clone() for t in model.generate(img_b)]
    torch.cuda.synchronize()
    errs = []

    def plain():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    ids, lp = model.generate(img_a, slot='thread_a')
                    st.synchronize()
                    assert torch.equal(ids, want_a[0]) and torch.equal(lp, want_a[1])
        except BaseException as e:       # noqa: BLE001
            errs.append(e)

    def piped():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                pend = [model.generate_async(img_b, lane=7) for _ in range(6)]
                for p in pend:
                    ids, lp = p.result()
                    assert torch.equal(ids, want_b[0]) and torch.equal(lp, want_b[1])
        except BaseException as e:       # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=plain), threading.Thread(target=piped)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_beam1_equals_greedy_tokens(model):
    """Beam search with one beam must pick the greedy tokens (scores are length-normalised differently)."""
    img = _images(2).cuda()
    g_ids, _ = model.generate(img)
    b_ids, _ = model.generate_beam(img, 1)
    assert torch.equal(g_ids, b_ids)


def test_run_py_pipeline_eval_surface(tmp_path, monkeypatch, golden):
    """run.py -c yaml flow: plugin-loaded pipeline, seeded weights saved as a reference-style checkpoint
    (DDP 'module.' prefixes) named by `model_file`, suffix-matching load, captions written as the reference's predict TSV rows."""
    import json
    import yaml
    import run
    from vitcap_amd.model import ImageCaptioning
    vec, _ = golden
    monkeypatch.chdir(tmp_path)
    enc = tmp_path / 'enc'
    enc.mkdir()
    toks = ['[PAD]'] + ['w%d' % i for i in range(1, 30522)]
    toks[100], toks[101], toks[102], toks[103] = '[UNK]', '[CLS]', '[SEP]', '[MASK]'
    (enc / 'vocab.txt').write_text('\n'.join(toks) + '\n')
    sd = ImageCaptioning().load_recipe(0).state_dict()
    ck = tmp_path / 'base.pt'
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}, 'iteration': 0}, ck)
    cfg = {'type': 'pipeline_eval_multi',
           'all_test_data': [{'test_data': 'synthetic', 'test_split': 'test'}],
           'param': {'full_expid': 'E', 'max_iter': 10, 'model_file': str(ck), 'text_encoder_type': str(enc),
                     'tagemb': 'cls', 'test_batch_size': 2, 'synthetic_num_images': 3, 'force_predict': True,
                     'pipeline_type': {'from': 'vitcap_amd.pipeline', 'import': 'CaptionUniPipeline'}}}
    yf = tmp_path / 'exp.yaml'
    yf.write_text(yaml.safe_dump(cfg))
    kw = run.parse_general_args(['-c', str(yf)])
    fn = kw.pop('type')
    getattr(run, fn)(**kw)
    out = str(ck) + '.synthetic.test.predict.tsv'
    rows = [l.rstrip('\n').split('\t') for l in open(out)]
    assert [r[0] for r in rows] == ['0_0', '0_1', '0_2']
    cap = json.loads(rows[0][1])[0]
    first = ' '.join('w%d' % t for t in vec['greedy_b1_ids'][0, 0, 1:4])          # tokens of the golden caption of image 0
    assert cap['caption'].startswith(first) and 0 < cap['conf'] < 1


def test_pipeline_predict_honours_the_batches_masks(tmp_path, monkeypatch, golden):
    """pipeline.predict (the overlapped 2-slot path) decodes every batch with the options ITS attention_mask asks for: a batch whose
    mask shows 50 visible tag slots gets tag_visible = 50 (round 2 validated the mask and then decoded with tag_visible = 0), host and
    device text tensors alike; the device tensors of the steady state are checked without a host synchronisation and a bad one is
    reported when its captions are collected."""
    import json
    from oracle import vitcap_oracle as O
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.pipeline import CaptionUniPipeline
    monkeypatch.chdir(tmp_path)
    enc = tmp_path / 'enc'
    enc.mkdir()
    toks = ['[PAD]'] + ['w%d' % i for i in range(1, 30522)]
    toks[100], toks[101], toks[102], toks[103] = '[UNK]', '[CLS]', '[SEP]', '[MASK]'
    (enc / 'vocab.txt').write_text('\n'.join(toks) + '\n')
    img = _images(2)

    def batch(n_tag, dev, keys):
        input_ids, am = O.test_text_inputs(2, n_tag_visible=n_tag)
        return {'image': img.clone(), 'key': keys, 'input_ids': input_ids.to(dev), 'attention_mask': am.to(dev),
                'token_type_ids': torch.zeros(2, 70, dtype=torch.long, device=dev)}

    def run(batches, name):
        p = CaptionUniPipeline(full_expid='E', init_recipe_seed=0, text_encoder_type=str(enc), tagemb='cls', force_predict=True,
                               test_batches=batches, model_file=str(tmp_path / (name + '.pt')))
        out = p.ensure_predict()
        rows = {}
        for l in open(out):
            key, js = l.rstrip('\n').split('\t')
            rec = json.loads(js)[0]
            rows[key] = (rec['caption'].split(), rec['conf'])
        return rows

    # what forward() -- whose tag handling is checked against the reference goldens in test_tag_tokens_visible_to_caption --
    # returns for the same batches: the pipeline's rows must be exactly these (same kernels, bit-identical pipeline)
    ref = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    ref.pack('cuda')

    def expect(n_tag):
        d = batch(n_tag, 'cuda', [0, 1])
        d['image'] = d['image'].cuda()
        ids, lp = ref(d)
        return [(['w%d' % int(t) for t in ids[i, 0, 1:] if int(t) not in (0, 101, 102)], float(torch.exp(lp[i, 0]))) for i in range(2)]
    want0, want50 = expect(0), expect(50)
    assert want0 != want50, 'visible tags must change the caption or its confidence'
    rows = run([batch(0, 'cpu', ['a0', 'a1']), batch(50, 'cpu', ['b0', 'b1']), batch(50, 'cuda', ['c0', 'c1']), batch(50, 'cuda', ['d0', 'd1'])],
               'mixed')
    for k, want in (('a', want0), ('b', want50), ('c', want50), ('d', want50)):
        for i in range(2):
            cap, conf = rows['%s%d' % (k, i)]
            assert cap == want[i][0] and abs(conf - want[i][1]) <= 1e-7 * want[i][1], (k, i, conf, want[i][1])
    # consecutive device-resident batches that CHANGE the count: one batch is always in flight, so the second of two
    # batches with a new count was submitted against the old one as well -- it must be decoded again with its own options, not
    # reported as an unsupported mask; and a third change back is handled the same way
    rows = run([batch(50, 'cuda', ['f0', 'f1']), batch(0, 'cuda', ['g0', 'g1']), batch(0, 'cuda', ['h0', 'h1']), batch(0, 'cuda', ['i0', 'i1']),
                batch(50, 'cuda', ['j0', 'j1']), batch(50, 'cuda', ['k0', 'k1'])], 'changing')
    for k, want in (('f', want50), ('g', want0), ('h', want0), ('i', want0), ('j', want50), ('k', want50)):
        for i in range(2):
            cap, conf = rows['%s%d' % (k, i)]
            assert cap == want[i][0] and abs(conf - want[i][1]) <= 1e-7 * want[i][1], (k, i, conf, want[i][1])
    bad = batch(50, 'cuda', ['e0', 'e1'])
    bad['attention_mask'][1, 3, 9] = 1
    with pytest.raises(NotImplementedError, match='mask structure'):
        run([batch(50, 'cuda', ['c0', 'c1']), bad], 'bad')


def test_generate_async_pipeline_equals_generate(model):
    """Two-slot batch pipeline (encode of batch i+1 overlapping the decode of batch i on a second stream): every batch's
    ids / log-probs are bit-identical to the one-stream generate(), in order, across slot reuse and changing batch size."""
    from vitcap_amd import weights as W
    imgs = [torch.from_numpy(W.gen_image_batch(b, 100 + i)).cuda().to(torch.bfloat16) for i, b in enumerate((8, 8, 5, 8, 3, 8))]
    want = [tuple(t.clone() for t in model.generate(im)) for im in imgs]
    torch.cuda.synchronize()
    pend = [model.generate_async(im) for im in imgs]
    for (ids_w, lp_w), p in zip(want, pend):
        ids, lp = p.result()
        assert torch.equal(ids, ids_w) and torch.equal(lp, lp_w)
    # non-blocking hand-over to the caller's stream
    p = model.generate_async(imgs[0])
    ids, lp = p.wait()
    assert torch.equal(ids.clone(), want[0][0])


def test_pipeline_streams_belong_to_the_process(model):
    """A second model object of the process runs its 2-slot pipeline on the SAME encoder / decode streams as the first (model.role_stream)
    and gives the same results: with a fresh stream pair per model every new model drew another stream -> hardware-queue arrangement, and
    some of them halved the pipeline's rate (profiles/r05_hw_queue_aliasing.txt)."""
    from vitcap_amd import weights as W
    from vitcap_amd.model import ImageCaptioning, role_stream
    imgs = [torch.from_numpy(W.gen_image_batch(8, 300 + i)).cuda().to(torch.bfloat16) for i in range(3)]
    want = [model.generate_async(im).result() for im in imgs]
    other = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    other.pack('cuda')
    got = [other.generate_async(im).result() for im in imgs]
    for (a, b), (c, d) in zip(want, got):
        assert torch.equal(a, c) and torch.equal(b, d)
    pa, pb = model._pipes[0], other._pipes[0]
    assert pa['enc'] is pb['enc'] and pa['dec'] is pb['dec']
    dev = torch.device('cuda', torch.cuda.current_device())
    assert role_stream(dev, 'enc0', lambda: None) is pa['enc']


def test_generate_async_beam_equals_generate_beam(model):
    from vitcap_amd import weights as W
    imgs = [torch.from_numpy(W.gen_image_batch(3, 200 + i)).cuda().to(torch.bfloat16) for i in range(3)]
    want = [tuple(t.clone() for t in model.generate_beam(im, 3)) for im in imgs]
    pend = [model.generate_async(im, num_beams=3) for im in imgs]
    for (ids_w, lp_w), p in zip(want, pend):
        ids, lp = p.result()
        assert torch.equal(ids, ids_w) and torch.equal(lp, lp_w)


def test_beam5_batch256_properties(model, sd_t):
    """BASELINE configs[2] size (beam=5, 256 images = 1280 sequences, decode loop replayed from a hipGraph): determinism, graph
    replay == eager, batch invariance (an image captioned alone gets the same beam result), well-formed ids, finite scores -- and
     an ORACLE comparison at this size: the first 4 of the 256 images against the oracle's beam driver on the
    bf16-emulated incremental model (itself pinned to the reference's beam goldens), under the rule of test_beam_search_vs_oracle."""
    from oracle import vitcap_oracle as O
    B = 256
    img = _images(B, 4321).cuda().to(torch.bfloat16)
    ids1, lp1 = [t.clone() for t in model.generate_beam(img, 5, use_graph=True)]
    with torch.no_grad():
        ids_o, lp_o, gaps = O.beam_incremental(sd_t, img[:4].float().cpu(), num_beams=5, emulate_bf16=True, return_margins=True)
    ok = gaps.min(1).values > BEAM_GAP_TOL
    same = (ids1[:4].cpu() == ids_o).all(-1).all(-1)
    print('beam 5 x 256 vs oracle on images 0-3: ids equal %s, min decision gaps %s, |score diff| %s' % (
        same.tolist(), ['%.1e' % g for g in gaps.min(1).values.tolist()], ['%.1e' % d for d in (lp1[:4].cpu() - lp_o).abs().view(-1).tolist()]))
    assert bool(same[ok].all()), 'B = 256 beam result differs from the oracle on an image whose decision gaps clear the floor'
    np.testing.assert_allclose(lp1[:4].cpu().numpy(), lp_o.numpy(), atol=1e-2)
    ids2, lp2 = model.generate_beam(img, 5, use_graph=True)
    assert torch.equal(ids1, ids2) and torch.equal(lp1, lp2), 'non-deterministic'
    ids3, lp3 = model.generate_beam(img, 5, use_graph=False)
    assert torch.equal(ids1, ids3) and torch.equal(lp1, lp3), 'hipGraph replay differs from eager launches'
    ids_s, lp_s = model.generate_beam(img[:210].contiguous(), 5)          # 2100 rows: same split-K class as 2560 -> bit-identical
    assert torch.equal(ids_s, ids1[:210]) and torch.allclose(lp_s, lp1[:210], atol=1e-6)
    ids_s, lp_s = model.generate_beam(img[:100].contiguous(), 5)          # 1000 rows: 6 / 12 split-K slabs, other summation order
    assert torch.allclose(lp_s, lp1[:100], atol=5e-3)
    ids_s, lp_s = model.generate_beam(img[:3].contiguous(), 5)            # 30 rows: whole-K decode GEMMs, other summation order
    assert torch.allclose(lp_s, lp1[:3], atol=5e-3)
    i = ids1.cpu()[:, 0]
    assert (i[:, 0] == 101).all() and ((i >= 0) & (i < 30522)).all()
    for row in i.tolist():
        assert 102 in row, 'a beam hypothesis always ends with [SEP]'
        k = row.index(102)
        assert all(v == 0 for v in row[k + 1:])
    assert torch.isfinite(lp1).all() and float(lp1.max()) <= 0.0
    # beam search never scores far below its own greedy member under the same normalisation
    _, lp_g = model.generate(img)
    assert float((lp1 - lp_g).min()) > -0.35


@pytest.mark.parametrize('B', [1, 5, 63, 65, 129])
def test_ragged_batch_sizes(model, B):
    """Edge batch sizes (row counts B*577 that end inside a GEMM tile, a single image, one more / one fewer than the
    benchmark batch): every caption equals the one the image gets in another batch composition, for the one-stream path,
    the two-slot pipeline and (small B) beam search; fp32 and bf16 image inputs agree."""
    img = _images(max(B, 6)).cuda()
    ref_ids, ref_lp = model.generate(img[:6].to(torch.bfloat16).contiguous())
    ref_ids, ref_lp = ref_ids.clone(), ref_lp.clone()
    x = img[:B].to(torch.bfloat16).contiguous()
    ids, lp = model.generate(x)
    ids, lp = ids.clone(), lp.clone()
    n = min(B, 6)
    assert ids.shape == (B, 1, 20) and torch.equal(ids[:n], ref_ids[:n])
    # up to 128 sequences the decode-step GEMMs run in their whole-K form, above in the split-K form: same kernels ->
    # bit-identical scores; across the boundary the summation order differs (bf16-level noise in the scores)
    np.testing.assert_allclose(lp[:n].cpu().numpy(), ref_lp[:n].cpu().numpy(), atol=1e-6 if B <= 128 else 2e-3)
    ids32, lp32 = model.generate(img[:B].to(torch.bfloat16).float().contiguous())      # fp32 input holding bf16 values
    assert torch.equal(ids32, ids) and torch.equal(lp32, lp)
    a1, a2 = model.generate_async(x), model.generate_async(x)
    for h in (a1, a2):
        i2, l2 = h.result()
        assert torch.equal(i2, ids) and torch.equal(l2, lp)
    if B <= 5:
        bi, bl = model.generate_beam(x, 3)
        bi = bi.clone()
        bi6, _ = model.generate_beam(img[:6].to(torch.bfloat16).contiguous(), 3)
        assert torch.equal(bi, bi6[:B])


def test_cls_only_tag_block_equals_full_block(monkeypatch):
    """The last tag block computed for the CLS row alone (default) against the full 577-row block (VITCAP_FULL_TAG_BLOCK=1):
    same captions, log-probabilities and tag logits -- the rows that are skipped feed nothing."""
    from vitcap_amd.model import ImageCaptioning
    img = _images(5, 77).cuda().to(torch.bfloat16)
    outs = []
    for full in ('0', '1'):
        monkeypatch.setenv('VITCAP_FULL_TAG_BLOCK', full)
        m = ImageCaptioning().load_recipe(0).eval()
        m.pack('cuda')
        ids, lp = m.generate(img, want_tags=True)
        torch.cuda.synchronize()
        tag_logits = m.tap('tag_logits', 5, (5, 30592)).cpu()[:, :30522]
        tag_cls = m.tap('tag_hidden', 5, (5, 577, 768)).cpu()[:, 0]
        outs.append((ids.cpu().clone(), lp.cpu().clone(), tag_logits.clone(), tag_cls.clone()))
    (i0, l0, t0, c0), (i1, l1, t1, c1) = outs
    assert torch.equal(i0, i1)
    np.testing.assert_allclose(l0.numpy(), l1.numpy(), atol=2e-5)
    rel = float((c0 - c1).norm() / c1.norm())
    print('tag CLS row rel diff %.2e, tag logits max diff %.2e' % (rel, float((t0 - t1).abs().max())))
    assert rel < 2e-3
    np.testing.assert_allclose(t0.numpy(), t1.numpy(), atol=2e-2)
