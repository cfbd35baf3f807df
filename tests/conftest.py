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
