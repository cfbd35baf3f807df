"""Golden vectors for the sampling branch of the decode step (container only).

Run:  python tests/golden/make_golden_sample.py     (writes tests/golden/reference_sample.npz)

Imports the reference's own ``top_k_top_p_filtering`` (src/layers/bert/modeling_utils.py:1103-1135) with the shims
of make_golden.py and records, for seeded logits rows, which entries it keeps for a grid of (temperature-scaled
logits, top_k, top_p).  Only inputs' seeds and expected keep masks are stored -- fixtures are data.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims  # noqa: E402

CASES = [(0, 0.9), (0, 0.5), (0, 0.05), (7, 1.0), (50, 1.0), (50, 0.8), (1, 1.0), (3, 0.3), (2000, 0.95)]


def make_logits(seed, B=4, V=3001, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, V, generator=g) * scale
    x[0, 10] = x[0, 11]                       # an exact tie inside the row
    return x


def main():
    install_shims()
    from src.layers.bert.modeling_utils import top_k_top_p_filtering
    out = {}
    x0 = make_logits(2024)
    out['logits_seed'] = np.array(2024)
    out['logits_digest'] = np.array(float(x0.double().abs().sum()))
    for n, (k, p) in enumerate(CASES):
        y = top_k_top_p_filtering(x0.clone(), top_k=k, top_p=p)
        keep = torch.isfinite(y)
        out['case%d_kp' % n] = np.array([k, p], dtype=np.float64)
        out['case%d_keep' % n] = np.packbits(keep.numpy(), axis=1)
        out['case%d_count' % n] = keep.sum(1).numpy()
    out['torch_version'] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, 'reference_sample.npz'), **out)
    print({k: (v.tolist() if v.size < 8 else v.shape) for k, v in out.items() if 'count' in k})


if __name__ == '__main__':
    main()
