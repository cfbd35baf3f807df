"""Sampling branch of the decode step: the oracle's filter against the reference's own
top_k_top_p_filtering (tests/golden/reference_sample.npz), and the Gumbel-max draw against the softmax it must follow."""
import os

import numpy as np
import torch

from oracle import vitcap_oracle as O
from tests.golden.make_golden_sample import CASES, make_logits

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'reference_sample.npz')


def test_filter_equals_reference_golden():
    z = np.load(GOLD)
    x = make_logits(int(z['logits_seed']))
    assert abs(float(x.double().abs().sum()) - float(z['logits_digest'])) < 1e-6, 'seeded logits changed'
    for n, (k, p) in enumerate(CASES):
        assert z['case%d_kp' % n].tolist() == [k, p]
        y = O.top_k_top_p_filter(x, k, p)
        keep = torch.isfinite(y).numpy()
        want = np.unpackbits(z['case%d_keep' % n], axis=1)[:, :x.shape[1]].astype(bool)
        assert np.array_equal(keep, want), (k, p, keep.sum(1), want.sum(1))
        assert torch.equal(y[torch.from_numpy(keep)], x[torch.from_numpy(keep)])     # survivors untouched


def test_counter_rng_uniform():
    r = O.rng_mix(O.rng_mix(np.uint32(7), np.uint32(3)), np.arange(200000, dtype=np.uint32))
    u = O.rng_uniform(r)
    assert u.dtype == np.float32 and u.min() > 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 2e-3
    # known answers pin the bit-level definition shared with csrc/rng.h
    assert [int(v) for v in O.rng_mix(np.uint32(0), np.arange(3, dtype=np.uint32))] == KNOWN_MIX


KNOWN_MIX = [33350994, 2672842292, 127880910]      # vc_mix(0, 0..2) computed by the C definition in csrc/rng.h


def test_gumbel_max_follows_softmax():
    """Chi-square of 40000 draws against softmax(filtered logits): the restated draw has the reference's distribution."""
    V, N = 12, 40000
    logits = torch.tensor([[2.0, 1.5, 1.0, 0.5, 0.0, -0.5, -1.0, 3.0, -2.0, 0.2, 0.1, 1.2]])
    samp = O.make_sampler(temperature=0.8, top_k=8, top_p=0.95, seed=11)
    x = O.top_k_top_p_filter(logits / 0.8, 8, 0.95)
    p = torch.softmax(x, -1)[0].numpy()
    counts = np.zeros(V)
    big = logits.expand(500, V).contiguous()
    for t in range(N // 500):
        tok, lp, _ = samp(big, t)
        counts += np.bincount(tok.numpy(), minlength=V)
        assert torch.allclose(lp, torch.log(torch.from_numpy(p))[tok], atol=1e-6)
    assert counts[p == 0].sum() == 0
    e = p[p > 0] * N
    chi2 = float(((counts[p > 0] - e) ** 2 / e).sum())
    assert chi2 < 30.0, chi2            # dof <= 7; P(chi2 > 30) < 1e-4


BS_GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'reference_beamsample.npz')


def test_filter_min_tokens_to_keep_equals_reference_golden():
    """The beam-sampling call of the filter (min_tokens_to_keep = 2, modeling_utils.py:970-972): k = max(top_k, 2) and ranks
    0..2 always survive top-p (the reference clears the flags of ranks 0..1 before its shift-by-one)."""
    from tests.golden.make_golden_beamsample import FILTER_CASES
    z = np.load(BS_GOLD)
    x = make_logits(2024)
    for n, (k, p) in enumerate(FILTER_CASES):
        assert z['filter%d_kp' % n].tolist() == [k, p]
        keep = torch.isfinite(O.top_k_top_p_filter(x, k, p, min_tokens_to_keep=2)).numpy()
        want = np.unpackbits(z['filter%d_keep' % n], axis=1)[:, :x.shape[1]].astype(bool)
        assert np.array_equal(keep, want), (k, p, keep.sum(1), want.sum(1))


def test_gumbel_top2_is_sampling_without_replacement():
    """Two largest of x + G == two draws without replacement from softmax(x): first-draw frequencies follow p, and the
    second draw given the first follows p renormalised without it (chi-square on the pair table)."""
    x = torch.tensor([[1.5, 0.5, 0.0, -0.5, 1.0]])
    p = torch.softmax(x, -1)[0].double().numpy()
    V, N = 5, 60000
    pair = np.zeros((V, V))
    big = x.expand(1000, V).contiguous()
    for t in range(N // 1000):
        idx = O.gumbel_top2(big, seed=3, t=t).numpy()
        assert (idx[:, 0] != idx[:, 1]).all()
        np.add.at(pair, (idx[:, 0], idx[:, 1]), 1)
    e = np.array([[0 if i == j else p[i] * p[j] / (1 - p[i]) for j in range(V)] for i in range(V)]) * N
    m = e > 0
    chi2 = float(((pair[m] - e[m]) ** 2 / e[m]).sum())
    assert chi2 < 55.0, chi2            # dof 19; P(chi2 > 55) < 1e-4


def test_beam_sampling_bookkeeping_matches_reference(sd_t):
    """num_beams > 1 with do_sample (modeling_utils.py:966-985) against the reference's own output with torch.multinomial
    replaced by the counter-based Gumbel top-2 draw (tests/golden/make_golden_beamsample.py): pins temperature -> filter with
    min_tokens_to_keep 2 -> log-softmax scores, the positional candidate order, the beam attribution `p % num_beams` of the
    reference's "match shape of greedy beam search" step, repetition penalty and the n-best list."""
    from vitcap_amd import weights as W
    z = np.load(BS_GOLD)
    assert int(z['recipe_version']) == W.RECIPE_VERSION
    n = 0
    while 'case%d_cfg' % n in z:
        beams, keep, B, temp, top_k, top_p, rep, seed = z['case%d_cfg' % n]
        im = torch.from_numpy(W.gen_image_batch(int(B), int(z['image_seed'])))
        sd_ = int(seed)
        draw = lambda x, t: O.gumbel_top2(torch.log(torch.softmax(x, dim=-1)), sd_, t)
        with torch.no_grad():
            ids, lp = O.beam_incremental(sd_t, im, num_beams=int(beams), num_keep_best=int(keep), repetition_penalty=float(rep),
                                         sample=dict(temperature=float(temp), top_k=int(top_k), top_p=float(top_p), seed=sd_,
                                                     draw=draw))
        np.testing.assert_array_equal(ids.numpy(), z['case%d_ids' % n])
        np.testing.assert_allclose(lp.numpy(), z['case%d_logprobs' % n], rtol=2e-5, atol=2e-5)
        n += 1
    assert n == 3
