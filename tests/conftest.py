import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: CPU test that takes more than ~30 s')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    import json
    d = os.path.join(REPO, 'tests', 'golden')
    vec = dict(np.load(os.path.join(d, 'reference_vectors.npz')))
    with open(os.path.join(d, 'reference_meta.json')) as f:
        meta = json.load(f)
    return vec, meta


@pytest.fixture(scope='session')
def sd_np():
    from vitcap_amd import weights as W
    return W.make_state_dict(seed=0, tie_weights=True)


@pytest.fixture(scope='session')
def sd_t(sd_np):
    from oracle import vitcap_oracle as O
    return O.to_torch(sd_np)


# ---------------------------------------------------------------------------------------------------------------
# Token-exact comparison with the REFERENCE's own fp32 output (tests/golden, produced by running /root/reference).
# The device computes in bf16 with fp32 accumulation; its logits differ from the reference's fp32 logits by activation
# rounding noise (measured on the recipe: 1.8e-3 rms per logit, 9e-3 max over a 30522-wide row).  The
# goldens carry, for every discrete decision the reference took, the margin of that decision (greedy: top-1 minus top-2
# logit; beam: smallest gap between neighbours among the 2*beams+1 best candidate scores).  Above the floor the device must
# reproduce the reference's decision exactly; the first sub-floor decision ends the comparable prefix of that sequence.
GREEDY_MARGIN_FLOOR = 0.012      # logit units: > 4 sigma of the noise on a logit difference (sqrt(2) * 1.8e-3 = 2.5e-3)
BEAM_MARGIN_FLOOR = 0.03         # cumulative log-prob scores: noise grows with sqrt(steps), plus the logsumexp term


def comparable_prefix(margins_row, floor):
    """Number of leading decisions (steps) whose reference margin is >= floor."""
    import numpy as np
    below = np.nonzero(np.asarray(margins_row) < floor)[0]
    return int(below[0]) if len(below) else len(margins_row)


def relevant_margins(want_row, margins_row, eos):
    """Margins of the decisions that can change the returned ids: the reference records one per step for every row, also after
    the row has finished (those tokens are replaced by PAD, modeling_utils.py:855-858) and at the last position, where an
    unfinished row gets the forced EOS whatever was chosen (modeling_utils.py:870-871)."""
    import numpy as np
    m = np.array(margins_row, dtype=np.float64)
    L = len(want_row)
    eos_set = set(int(e) for e in eos) if isinstance(eos, (list, tuple, set)) else {int(eos)}     # eos_token_ids may hold several ids
    hits = [k for k in range(1, L) if int(want_row[k]) in eos_set]
    end = hits[0] if hits else L - 1           # position of the terminating EOS (chosen or forced)
    m[end:] = np.inf                           # decisions for positions > end
    if not hits or hits[0] == L - 1:
        m[L - 2] = np.inf                      # the last position's choice does not reach the ids
    return m


def assert_tokens_match_reference(got_ids, want_ids, margins, floor, min_full=1, what='', eos=102):
    """got_ids / want_ids: (B, n_best, 20) with n_best == 1 for the comparison; margins (B, 19), decision t-1 picks the
    token at position t.  Asserts got == want on every sequence's comparable prefix (positions 0..p where p = number of
    leading above-floor decisions) and on the WHOLE sequence when all its decisions are above the floor; at least `min_full`
    sequences must be whole-sequence comparable so that the test cannot pass vacuously.  Returns the per-sequence status."""
    import numpy as np
    got, want = np.asarray(got_ids)[:, 0], np.asarray(want_ids)[:, 0]
    margins = np.stack([relevant_margins(want[b], margins[b], eos) for b in range(want.shape[0])])
    full, report = 0, []
    for b in range(want.shape[0]):
        p = comparable_prefix(margins[b], floor)
        whole = p == margins.shape[1]
        full += int(whole)
        n = want.shape[1] if whole else p + 1              # positions 0..p inclusive (position 0 is [CLS])
        ok = bool((got[b, :n] == want[b, :n]).all())
        report.append((b, p, whole, ok, bool((got[b] == want[b]).all())))
        assert ok, ('%s sequence %d differs from the reference inside its comparable prefix (%d decisions above the %.3g '
                    'floor, min margin %.4g):\n got  %s\n want %s' % (what, b, p, floor, float(np.min(margins[b, :max(p, 1)])),
                                                                      got[b].tolist(), want[b].tolist()))
    assert full >= min_full, '%s: only %d sequences are whole-sequence comparable (need %d): %s' % (what, full, min_full, report)
    return report
