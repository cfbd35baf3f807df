"""Host side of the self-critical step: the CIDEr-D restatement (properties; the pyciderevalcap package is absent, so its
values are not pinned) and the reward rule of ScstRewardCriterion (utils_caption_evaluate.py:172-202)."""
import math

import torch

from vitcap_amd.scst import CiderD, scst_rewards


def test_ciderd_properties():
    refs = [['a man riding a horse on the street', 'a person rides a brown horse'],
            ['two dogs play with a red ball in the grass', 'dogs running after a ball'],
            ['a plate of food with pizza on a table', 'a pizza on a wooden table']]
    sc = CiderD()
    _, same = sc.compute_score(refs, [r[0] for r in refs])
    _, other = sc.compute_score(refs, [refs[1][0], refs[2][0], refs[0][0]])
    _, empty = sc.compute_score(refs, ['', '', ''])
    assert all(a > 3.0 for a in same) and all(a > 10 * b for a, b in zip(same, other)) and empty == [0.0, 0.0, 0.0]
    # length penalty: repeating a perfect caption twice lowers the score (gaussian on the length difference, clipped counts)
    _, doubled = sc.compute_score(refs[:1] * 1 + refs[1:], [refs[0][0] + ' ' + refs[0][0], refs[1][0], refs[2][0]])
    assert doubled[0] < same[0]
    # order of the references does not matter; scores are finite and non-negative
    _, swapped = sc.compute_score([r[::-1] for r in refs], [r[0] for r in refs])
    assert all(abs(a - b) < 1e-9 for a, b in zip(same, swapped)) and all(math.isfinite(v) and v >= 0 for v in other)


def test_reward_rule():
    gts = [['a man riding a horse'], ['a dog with a ball']]
    greedy = ['a man riding a horse', 'a cat']
    samples = ['a man riding a horse', 'a man', 'a dog with a ball', 'a cat']      # K = 2 per image
    r, score = scst_rewards(gts, greedy, samples)
    assert r.shape == (4,) and r.dtype == torch.float32
    assert abs(float(r[0])) < 1e-6                 # sample == greedy of the same image -> zero advantage
    assert float(r[1]) < 0 and float(r[2]) > 0     # worse / better than the baseline
    assert abs(float(r[3])) < 1e-6
    assert score > 0
