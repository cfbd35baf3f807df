"""Golden vectors for generate() with repetition_penalty != 1 (container only).

Run:  python tests/golden/make_golden_reppen.py      (writes tests/golden/reference_reppen.npz)

Imports the reference model with the shims of make_golden.py, loads the seeded recipe weights and runs ViTCAP.generate
(modeling_bert.py:928-1059; penalty at modeling_utils.py:828-836 greedy, 955-963 beam) on seeded images.
Stored: the returned ids and logprobs per case (num_beams, repetition_penalty, images).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, build_reference, load_recipe, REPO  # noqa: E402

CASES = [(1, 1.3, 2), (2, 1.3, 1), (1, 0.8, 1)]          # beams, repetition_penalty, images


def main():
    install_shims()
    sys.path.insert(0, REPO)
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, enc = build_reference('cls', True)
    load_recipe(model, enc, W.make_state_dict(seed=0, tie_weights=True))
    out = {'torch_version': np.array(torch.__version__), 'image_seed': np.array(1234)}
    for n, (beams, rp, B) in enumerate(CASES):
        img = torch.from_numpy(W.gen_image_batch(B, 1234))
        input_ids, am = O.test_text_inputs(B)
        with torch.no_grad():
            img_feats = enc(img)
            full = O.construct_attn_mask(am, img_feats.shape[1])
            ids, logp = model(img_feats=img_feats, input_ids=input_ids, attention_mask=full,
                              masked_pos=torch.ones(B, 70, dtype=torch.int32),
                              token_type_ids=torch.zeros(B, 70, dtype=torch.long), label=torch.zeros(B, 30522),
                              gen_tag_ratio=1, is_decode=True, do_sample=False, bos_token_id=101, pad_token_id=0,
                              eos_token_ids=[102], mask_token_id=103, add_od_labels=True, od_labels_start_posid=20,
                              max_length=20, num_beams=beams, temperature=1, top_k=0, top_p=1, repetition_penalty=rp,
                              length_penalty=1, num_return_sequences=1, num_keep_best=1)
        out['case%d_cfg' % n] = np.array([beams, rp, B], dtype=np.float64)
        out['case%d_ids' % n] = ids.numpy().copy()
        out['case%d_logprobs' % n] = logp.numpy().copy()
        print(beams, rp, ids.tolist(), logp.tolist())
    np.savez_compressed(os.path.join(HERE, 'reference_reppen.npz'), **out)


if __name__ == '__main__':
    main()
