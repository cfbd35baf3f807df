"""Golden vectors for beam search WITH sampling (num_beams > 1 and do_sample; container only).

Run:  python tests/golden/make_golden_beamsample.py      (writes tests/golden/reference_beamsample.npz)

Imports the reference model with the shims of make_golden.py, loads the seeded recipe weights and runs ViTCAP.generate
(modeling_bert.py:928-1059 -> _generate_beam_search, do_sample branch modeling_utils.py:966-985) on seeded images.

The reference draws with ``torch.multinomial(softmax(filtered), num_samples=2)``; its generator stream is not something
another implementation can replay, so FOR THE DURATION OF THE REFERENCE CALL ``torch.multinomial`` is replaced by a
deterministic draw from the same distribution: the two largest of log(p) + Gumbel noise, the noise being the counter-based
stream of oracle.gumbel_noise (seed, row, step).  Everything else -- repetition penalty, temperature, the top-k / top-p
filter with min_tokens_to_keep = 2, the log-softmax scores, the "match shape of greedy beam search" re-indexing, the
BeamHypotheses bookkeeping -- is the reference's own code.  Stored: the configuration, the returned ids and logprobs, and the
filter's keep masks for min_tokens_to_keep = 2 on seeded rows.  The oracle's restatement must reproduce the ids exactly
(asserted here and in tests/test_oracle_golden.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, build_reference, load_recipe, REPO  # noqa: E402
from make_golden_sample import make_logits  # noqa: E402

# beams, keep, images, temperature, top_k, top_p, repetition_penalty, seed
CASES = [(3, 1, 2, 1.0, 0, 1.0, 1.0, 7), (2, 2, 2, 0.8, 50, 0.9, 1.0, 8), (4, 1, 1, 1.2, 0, 0.7, 1.2, 9)]
FILTER_CASES = [(0, 0.9), (0, 0.05), (1, 1.0), (1, 0.3), (50, 0.01), (3, 0.5)]


class _FixedDraw:
    """torch.multinomial(p, num_samples=2) -> Gumbel-top-2 of log p with the oracle's counter-based noise; counts the calls
    (= decode steps, starting at cur_len 1)."""

    def __init__(self, O, seed):
        self.O, self.seed, self.t = O, seed, 1

    def __enter__(self):
        self._orig = torch.multinomial

        def multinomial(p, num_samples, *a, **k):
            assert num_samples == 2 and p.dim() == 2
            idx = self.O.gumbel_top2(torch.log(p), self.seed, self.t)
            self.t += 1
            return idx
        torch.multinomial = multinomial
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._orig


def main():
    install_shims()
    sys.path.insert(0, REPO)
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    from src.layers.bert.modeling_utils import top_k_top_p_filtering
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = W.make_state_dict(seed=0, tie_weights=True)
    sd_t = O.to_torch(sd)
    model, enc = build_reference('cls', True)
    load_recipe(model, enc, sd)
    out = {'torch_version': np.array(torch.__version__), 'image_seed': np.array(1234), 'recipe_version': np.array(W.RECIPE_VERSION)}
    x0 = make_logits(2024)
    for n, (k, p) in enumerate(FILTER_CASES):
        keep = torch.isfinite(top_k_top_p_filtering(x0.clone(), top_k=k, top_p=p, min_tokens_to_keep=2))
        out['filter%d_kp' % n] = np.array([k, p], dtype=np.float64)
        out['filter%d_keep' % n] = np.packbits(keep.numpy(), axis=1)
        out['filter%d_count' % n] = keep.sum(1).numpy()
        print('filter', k, p, keep.sum(1).tolist())
    for n, (beams, keep, B, temp, top_k, top_p, rep, seed) in enumerate(CASES):
        img = torch.from_numpy(W.gen_image_batch(B, 1234))
        input_ids, am = O.test_text_inputs(B)
        with torch.no_grad(), _FixedDraw(O, seed):
            img_feats = enc(img)
            full = O.construct_attn_mask(am, img_feats.shape[1])
            ids, logp = model(img_feats=img_feats, input_ids=input_ids, attention_mask=full,
                              masked_pos=torch.ones(B, 70, dtype=torch.int32),
                              token_type_ids=torch.zeros(B, 70, dtype=torch.long), label=torch.zeros(B, 30522),
                              gen_tag_ratio=1, is_decode=True, do_sample=True, bos_token_id=101, pad_token_id=0,
                              eos_token_ids=[102], mask_token_id=103, add_od_labels=True, od_labels_start_posid=20,
                              max_length=20, num_beams=beams, temperature=temp, top_k=top_k, top_p=top_p,
                              repetition_penalty=rep, length_penalty=1, num_return_sequences=1, num_keep_best=keep)
        # the oracle's restatement, drawing from log_softmax of the filtered logits exactly as the patched call above did
        draw = lambda x, t: O.gumbel_top2(torch.log(torch.softmax(x, dim=-1)), seed, t)
        with torch.no_grad():
            ids_o, lp_o = O.beam_incremental(sd_t, img, num_beams=beams, num_keep_best=keep, repetition_penalty=rep,
                                             sample=dict(temperature=temp, top_k=top_k, top_p=top_p, seed=seed, draw=draw))
        assert torch.equal(ids_o, ids), (n, ids_o.tolist(), ids.tolist())
        assert torch.allclose(lp_o, logp, atol=2e-5), (lp_o, logp)
        out['case%d_cfg' % n] = np.array([beams, keep, B, temp, top_k, top_p, rep, seed], dtype=np.float64)
        out['case%d_ids' % n] = ids.numpy().copy()
        out['case%d_logprobs' % n] = logp.numpy().copy()
        print(beams, keep, B, ids.tolist(), logp.tolist())
    np.savez_compressed(os.path.join(HERE, 'reference_beamsample.npz'), **out)


if __name__ == '__main__':
    main()
