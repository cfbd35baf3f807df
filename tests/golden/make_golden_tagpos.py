"""Tag tokens visible to the caption with `od_labels_start_posid` above the generation length (container only; imports
/root/reference through make_golden.py's shims).

Run:  python tests/golden/make_golden_tagpos.py        (writes tests/golden/reference_tagpos.npz)

ViTCAP.generate gives the tag slots the position ids max(od_labels_start_posid, max_length) + j (modeling_bert.py:958-959,
983-992); the pipeline passes od_labels_start_posid = max_seq_a_length (..._bertemb.py:597), 20 in the shipped YAML and 40 by the
pipeline's own default (..._bertemb.py:197).  But the tag rows are OVERWRITTEN by ViTSplitCLSEmbModel.forward (:1435-1489), and
only one of its four embedding forms reads those position ids: bert.extra_embeddings (tagemb != 'cls', branch B, :1484-1485);
encode_tag_to_embedding uses a literal caption_len = 20 (:1381, :1396).  The goldens of reference_vectors.npz all use 20; this
family pins 40 on two structured images of reference_imgdep.npz, next to the same run at 20:
  tied_pos{20,40}_ids/logprobs/margins      pipeline flow (tagemb 'cls'): IDENTICAL for 20 and 40 -- asserted below
  untied_pos{20,40}_ids/logprobs/margins    notebook flow (tagemb None, vocabulary-bias sigma 0.25): differ -- asserted below
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402


def main():
    G.install_shims()
    from vitcap_amd import weights as W
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sel = np.load(os.path.join(HERE, 'reference_imgdep.npz'))['sel_index']
    cand = torch.from_numpy(W.gen_structured_images(48, 777))
    img = cand[torch.from_numpy(sel)][:2]
    out = {'sel_index': sel[:2].copy(), 'image_seed': np.array([777])}
    sd_np = W.make_state_dict(seed=0, tie_weights=True)
    model, enc = G.build_reference('cls', True)
    G.load_recipe(model, enc, sd_np)
    sd2 = W.make_state_dict(seed=0, tie_weights=False, vbias_std=0.25)
    model2, enc2 = G.build_reference(None, False)
    G.load_recipe(model2, enc2, sd2)
    for flow, (m_, e_) in (('tied', (model, enc)), ('untied', (model2, enc2))):
        for pos in (20, 40):
            ids, lp, mg = G.ref_generate(m_, e_, img, n_tag_visible=50, od_labels_start_posid=pos)
            print(flow, 'pos', pos, ids.tolist(), lp.tolist(), 'min margin', mg.min(1).tolist())
            out['%s_pos%d_ids' % (flow, pos)] = ids.numpy().copy()
            out['%s_pos%d_logprobs' % (flow, pos)] = lp.numpy().copy()
            out['%s_pos%d_margins' % (flow, pos)] = mg
        moved = (not np.array_equal(out[flow + '_pos20_ids'], out[flow + '_pos40_ids'])
                 or float(np.abs(out[flow + '_pos20_logprobs'] - out[flow + '_pos40_logprobs']).max()) > 0)
        print(flow, 'position changes the result:', moved)
        assert moved == (flow == 'untied'), 'od_labels_start_posid reaches the tag rows through bert.extra_embeddings only'
    out['untied_vbias_std'] = np.array([0.25], dtype=np.float32)
    out['torch_version'] = np.array([torch.__version__])
    np.savez_compressed(os.path.join(HERE, 'reference_tagpos.npz'), **out)
    print('wrote reference_tagpos.npz')


if __name__ == '__main__':
    main()
