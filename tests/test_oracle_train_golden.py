"""Pins the oracle's training step (loss, autograd gradients, clip, AdamW with the reference's parameter groups) against
vectors produced by the reference itself (tests/golden/make_golden_train.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vitcap_oracle as O
from vitcap_amd import weights as W

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(os.path.join(HERE, 'golden', 'reference_train.npz')))


@pytest.fixture(scope='module')
def step(sd_t):
    torch.set_num_threads(8)
    img = torch.from_numpy(W.gen_image_batch(2, 1234))
    return O.train_step_as_written(sd_t, img, O.synthetic_train_inputs(2), step=1), sd_t


@pytest.mark.slow
def test_losses_and_grad_norm(gold, step):
    res, _ = step
    assert abs(res['loss'] - float(gold['masked_loss'])) < 1e-4
    assert abs(res['tag_loss'] - float(gold['tag_loss'])) < 0.5
    assert abs(res['grad_norm'] - float(gold['grad_norm'])) < 1e-3 * float(gold['grad_norm'])


@pytest.mark.slow
def test_gradients_and_no_grad_set(gold, step):
    res, sd = step
    grads = res['grads']     # already clipped, like the reference's after clip_grad_norm_
    no_grad = set(gold['no_grad_keys'].tolist())
    have = set(grads.keys())
    for k in sd:
        if k == W.TIED_DST:
            continue
        assert (k in have) == (k not in no_grad), k
    for key in gold:
        if not key.startswith('grad_head__'):
            continue
        k = key[len('grad_head__'):]
        g = grads[W.TIED_SRC if k == W.TIED_DST else k]
        np.testing.assert_allclose(g.reshape(-1)[:32].numpy(), gold[key], rtol=2e-3, atol=1e-7, err_msg=k)
        assert abs(float(g.norm()) - float(gold['grad_norm__' + k])) <= 2e-3 * float(gold['grad_norm__' + k]) + 1e-9


@pytest.mark.slow
def test_adamw_update_matches_reference_optimizer(gold, step):
    res, sd = step
    for key in gold:
        if not key.startswith('delta_head__'):
            continue
        k = key[len('delta_head__'):]
        delta = (res['params'][k] - sd[k]).reshape(-1)[:32].numpy()
        np.testing.assert_allclose(delta, gold[key], rtol=2e-3, atol=2e-9, err_msg=k)


def test_param_groups_table():
    names = list(W.state_dict_spec().keys())
    pg = O.param_groups(names)
    assert pg['module.cls.predictions.bias'] is None
    assert pg['module.bert.encoder.blocks.7.attn.qkv.weight'] == (1e-5, 0.05)
    assert pg['module.bert.encoder.blocks.8.attn.qkv.weight'] == (1e-4, 0.05)
    assert pg['module.bert.encoder.tag_blocks.0.norm1.weight'] == (1e-5, 0.05)       # timm norms ARE decayed
    assert pg['module.bert.decoder.layer.0.output.LayerNorm.weight'] == (1e-4, 0.0)
    assert pg['module.bert.encoder.blocks.9.mlp.fc1.bias'] == (1e-4, 0.0)
    assert pg['module.bert.embeddings.word_embeddings.weight'] == (1e-4, 0.05)
