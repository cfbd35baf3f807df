"""Exact-match rate of the device's captions against every reference-produced fixture (GPU box):
greedy: reference_population.npz (32 images); beam: the 5 beam goldens of reference_vectors.npz (7 images), the image-dependent
beam-5 pair and the 8 population images.  Used to A/B kernel variants that move near ties (tools/exact_rate_ab.sh)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from vitcap_amd import weights as W          # noqa: E402
from vitcap_amd.model import ImageCaptioning  # noqa: E402

G = os.path.join(REPO, 'tests', 'golden')


def main():
    m = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    m.pack('cuda')
    pop, vec, dep = (dict(np.load(os.path.join(G, f))) for f in ('reference_population.npz', 'reference_vectors.npz', 'reference_imgdep.npz'))
    img4 = torch.from_numpy(W.gen_image_batch(4, 1234))
    cand16 = torch.from_numpy(W.gen_image_batch(16, 4321))
    fams = {'noise': torch.from_numpy(W.gen_image_batch(16, int(pop['pop_noise_seed'][0]))),
            'struct': torch.from_numpy(W.gen_structured_images(16, int(pop['pop_struct_seed'][0])))}
    g_eq = g_n = 0
    for fam, imgs in fams.items():
        ids, _ = m({'image': imgs.cuda(), 'key': list(range(16))})
        eq = (ids.cpu().numpy() == pop['pop_%s_ids' % fam]).all(-1).all(-1)
        g_eq += int(eq.sum()); g_n += len(eq)
    b_eq, b_n, detail = 0, 0, []

    def beam(name, images, nb, want, **kw):
        nonlocal b_eq, b_n
        ids, lp = m.generate_beam(images.cuda(), nb, **kw)
        eq = (ids.cpu().numpy() == want).all(-1).all(-1)
        b_eq += int(eq.sum()); b_n += len(eq)
        detail.append('%s %s' % (name, ''.join('=' if e else 'x' for e in eq)))
    beam('beam2_b1', img4[:1], 2, vec['beam2_b1_ids'])
    beam('beam5_b1', img4[:1], 5, vec['beam5_b1_ids'])
    beam('beam5_b2', img4[:2], 5, vec['beam5_b2_ids'])
    beam('beam3_alteos_b2', img4[:2], 3, vec['beam3_alteos_b2_ids'], eos_token_ids=[int(vec['alt_eos_id'][0])])
    beam('beam5_sel', cand16[torch.from_numpy(vec['beam_sel_index'])], 5, vec['beam5_sel_ids'])
    cand48 = torch.from_numpy(W.gen_structured_images(48, int(dep['image_seed'][0])))
    beam('imgdep_beam5', cand48[torch.from_numpy(dep['beam_index'])], 5, dep['beam5_ids'])
    for fam, imgs in fams.items():
        beam('pop_' + fam, imgs[:4], 5, pop['beam5_%s_ids' % fam])
    print('EXACT greedy %d/%d  beam %d/%d  [%s]' % (g_eq, g_n, b_eq, b_n, '; '.join(detail)))


if __name__ == '__main__':
    main()
