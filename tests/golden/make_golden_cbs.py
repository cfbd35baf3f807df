"""Golden vectors for constrained beam search, `use_cbs=True` (container only).

Run:  python tests/golden/make_golden_cbs.py      (writes tests/golden/reference_cbs.npz)

The reference's ViTCAP.generate takes `use_cbs`, `fsm`, `num_constraints`, `min_constraints_to_satisfy`
(modeling_bert.py:928-933, 949-953, 1035-1057) and runs `ConstrainedBeamSearch.search` + `select_best_beam_with_constraints`
(src/tools/captioning/utils_cbs.py:26-443).  As shipped that branch cannot even be imported: utils_cbs.py:1 imports `anytree`
(absent here; only `ConstraintFilter` uses it) and utils_cbs.py:9 imports `BeamHypotheses` from `src.tools.layers.bert.modeling_utils`,
a path that does not exist in the tree -- the class lives in `src.layers.bert.modeling_utils`.  Two throw-away shims make it run:
the inert `anytree` stub of make_golden.install_pipeline_shims and a sys.modules alias of the missing path to the reference's OWN
module.  Nothing else is touched; the search, the selection and the FSM builder execute as written.

Stored per case: the constraints as token ids (the FSM is rebuilt from them by the oracle's `fsm_build`, which this script checks
entry for entry against the reference's `FiniteStateMachineBuilder.build`), the returned ids / logprobs, every state's beams and
scores as `search` returned them (captured at the call of `select_best_beam_with_constraints`), and the decision margins from the
oracle's restatement, which must reproduce the reference's output exactly here or the script aborts."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_pipeline_shims, build_reference, load_recipe, REPO, REF  # noqa: E402

# (images, num_beams, max_given_constraints, per image: list of constraints, a constraint = class name of one or more words)
CASES = [
    (1, 2, 2, [['dog']]),
    (1, 3, 2, [['cat', 'fire hydrant']]),
    (2, 1, 3, [['dog'], ['cat', 'tree']]),
    # generate()'s decoding_constraint_flag (no word twice in a row) and bad_ending_ids (no EOS right behind these words)
    (1, 2, 2, [['dog']], {'decoding_constraint_flag': True, 'bad_ending_ids': [9138, 27024, 3899]}),
]
# constraint word -> tokens (constraint2tokens TSV), token -> word forms (tokenforms TSV); all single WordPiece tokens
C2T = {'dog': ['dog'], 'cat': ['cat'], 'fire': ['fire'], 'hydrant': ['hydrant'], 'tree': ['tree']}
FORMS = {'dog': ['dog', 'dogs'], 'cat': ['cat', 'cats'], 'tree': ['tree', 'trees']}


def alias_missing_module():
    import src.layers.bert.modeling_utils as mu
    for name in ('src.tools.layers', 'src.tools.layers.bert'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules['src.tools.layers.bert.modeling_utils'] = mu


def main():
    install_pipeline_shims()
    sys.path.insert(0, REPO)
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    alias_missing_module()
    import src.tools.captioning.utils_cbs as cbs
    from src.layers.bert.tokenization_bert import BertTokenizer
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tok = BertTokenizer(os.path.join(REF, 'yaml', 'VILT-L12-H784-uncased_16_384', 'vocab.txt'), do_lower_case=True)
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'c2t.tsv'), 'w') as f:
        for k, v in C2T.items():
            f.write('%s\t%s\n' % (k, ','.join(v)))
    with open(os.path.join(tmp, 'forms.tsv'), 'w') as f:
        for k, v in FORMS.items():
            f.write('%s\t%s\n' % (k, ','.join(v)))

    def ids_of(constraint):
        words = []
        for w in constraint.split():
            words.extend(C2T[w])
        return [tok.convert_tokens_to_ids(FORMS.get(w, [w])) for w in words]

    if '--fsm-only' not in sys.argv:
        model, enc = build_reference('cls', True)
        sd_np = W.make_state_dict(seed=0, tie_weights=True)
        load_recipe(model, enc, sd_np)
        sd = O.to_torch(sd_np)
    captured = {}
    real_select = cbs.select_best_beam_with_constraints

    def spy(beams, scores, *a, **k):
        captured['beams'], captured['scores'] = beams.clone(), scores.clone()
        return real_select(beams, scores, *a, **k)

    cbs.select_best_beam_with_constraints = spy
    out = {'torch_version': np.array(torch.__version__), 'image_seed': np.array(1234), 'vocab_size': np.array(tok.vocab_size)}
    fsm_only = '--fsm-only' in sys.argv          # refresh the builder digests only, keep the decoded vectors of the file on disk
    if fsm_only:
        out = dict(np.load(os.path.join(HERE, 'reference_cbs.npz')))
    out['c2t'] = np.array(['%s=%s' % (k, ','.join(v)) for k, v in C2T.items()])
    out['forms'] = np.array(['%s=%s' % (k, ','.join(v)) for k, v in FORMS.items()])
    only = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--only=')]      # recompute these cases, keep the others from disk
    if only:
        out = dict(np.load(os.path.join(HERE, 'reference_cbs.npz')))
    for n, case in enumerate(CASES):
        if only and n not in only:
            continue
        B, K, max_given, per_image = case[:4]
        extra = case[4] if len(case) > 4 else {}
        builder = cbs.FiniteStateMachineBuilder(tok, os.path.join(tmp, 'c2t.tsv'), os.path.join(tmp, 'forms.tsv'), max_given)
        fsms, used, cons_ids = [], [], []
        for cons in per_image:
            f_ref, nxt = builder.build(cons)
            cid = [ids_of(c) for c in cons]
            f_mine, nxt2 = O.fsm_build(cid, tok.vocab_size, max_given, 4)
            assert nxt == nxt2 and bool((f_ref == f_mine).all()), 'fsm_build differs from the reference builder'
            fsms.append(f_ref)
            used.append(nxt)
            cons_ids.append(cid)
        S = max(used)                               # "dynamically trim unused sub-states" (utils_cbs.py:685): the batch's largest
        fsm = torch.stack([f[:S, :S] for f in fsms])
        num_constraints = torch.tensor([len(c) for c in per_image])
        # digest of the REFERENCE builder's machines: words per transition, and a position-weighted checksum per transition
        wts = (torch.arange(tok.vocab_size, dtype=torch.int64) % 8191) + 1
        out['case%d_fsm_count' % n] = fsm.long().sum(-1).numpy().copy()
        out['case%d_fsm_check' % n] = (fsm.long() * wts).sum(-1).numpy().copy()
        out['case%d_constraint_names' % n] = np.array(['|'.join(c) for c in per_image])
        if fsm_only:
            continue
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
                              max_length=20, num_beams=K, temperature=1, top_k=0, top_p=1, repetition_penalty=1,
                              length_penalty=1, num_return_sequences=1, num_keep_best=1,
                              use_cbs=True, fsm=fsm, num_constraints=num_constraints, min_constraints_to_satisfy=2, **extra)
            o_ids, o_lp, m_search, m_sel, o_beams, o_scores = O.cbs_incremental(
                sd, img, fsm, num_constraints, K, 2, return_margins=True, no_repeat=bool(extra.get('decoding_constraint_flag')),
                bad_ending_ids=extra.get('bad_ending_ids'))
        ids, logp = ids[:, 0], logp[:, 0]
        print('case', n, 'S', S, 'ids', ids.tolist(), 'lp', logp.tolist())
        print('  oracle', o_ids.tolist(), o_lp.tolist(), 'margins', m_search.min(1).values.tolist(), m_sel.tolist())
        assert ids.shape == o_ids.shape and bool((ids == o_ids).all()), 'oracle restatement differs from the reference'
        assert float((logp - o_lp).abs().max()) < 2e-4
        # every VALID state's best beam must agree too (the other slots may hold -1e20 fillers whose order torch leaves open)
        for b in range(B):
            given = int(num_constraints[b])
            for s in range(2 ** given):
                if bin(s).count('1') >= min(given, 2):
                    assert bool((captured['beams'][b, s, 0] == o_beams[b, s, 0]).all()), (b, s)
        out['case%d_cfg' % n] = np.array([B, K, max_given, S], dtype=np.int64)
        out['case%d_no_repeat' % n] = np.array(int(bool(extra.get('decoding_constraint_flag'))))
        out['case%d_bad_ending_ids' % n] = np.array(extra.get('bad_ending_ids') or [], dtype=np.int64)
        out['case%d_num_constraints' % n] = num_constraints.numpy().copy()
        # constraints as a padded id table: [image, constraint, word, form] (-1 = unused)
        tab = -np.ones((B, 3, 4, 4), dtype=np.int64)
        for b, cid in enumerate(cons_ids):
            for c, words in enumerate(cid):
                for w, forms in enumerate(words):
                    tab[b, c, w, :len(forms)] = forms
        out['case%d_constraint_ids' % n] = tab
        out['case%d_ids' % n] = ids.numpy().copy()
        out['case%d_logprobs' % n] = logp.numpy().copy()
        out['case%d_beams' % n] = captured['beams'].numpy().copy()
        out['case%d_scores' % n] = captured['scores'].numpy().copy()
        out['case%d_margin_search' % n] = m_search.numpy().copy()
        out['case%d_margin_select' % n] = m_sel.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'reference_cbs.npz'), **out)


if __name__ == '__main__':
    main()
