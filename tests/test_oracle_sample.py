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
