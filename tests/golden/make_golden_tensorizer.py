"""Golden vectors for the text side (container only): the reference's own BertTokenizer and CaptionTensorizer
(src/layers/bert/tokenization_bert.py, src/data_layer/dataset.py:158-417) on a small synthetic vocabulary.

Run:  python tests/golden/make_golden_tensorizer.py   (writes tests/golden/reference_tensorizer.json)
Stores the vocabulary (generated here, not the reference's vocab.txt), the input strings / seeds / settings and the
outputs (token lists and tensors as lists)."""
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims  # noqa: E402

WORDS = ('a an the man woman dog cat sitting standing on in at of with and table bench street red blue two people riding '
         'horse bike next to large small plate food pizza holding umbrella tennis court player ball kite field grass '
         'un ##aff ##able ##s ##ing ##ed ##ly ##er walk play eat run jump cafe naive 你 好 , . ! ? \' " - ( ) $ ^').split()
TEXTS = ['A man riding a horse on the street.', 'Two people sitting at a table with pizza!',
         "The dog's red ball -- unaffable players walking quickly", 'Café naïve   tennis\tplayer\nrunning (jumped)',
         '你好 the cat [MASK] on a bench', 'zzzz qqqq ' + 'x' * 120, '', 'a ' * 60 + 'dog']
CASES = [dict(mask_type='seq2seq', is_train=True), dict(mask_type='seq2seq', is_train=False),
         dict(mask_type='seq2seq_off', is_train=True, ignore_sep=True), dict(mask_type='bidirectional', is_train=True, mask_b=True),
         dict(mask_type='seq2seq', is_train=True, max_masked_tokens=5, mask_prob=0.5, replace_by_mask_prob=0.3)]


def pack(t):
    import numpy as np
    return {'shape': list(t.shape), 'hex': np.packbits(t.numpy().astype(bool).reshape(-1)).tobytes().hex()}


def main():
    install_shims()
    from src.layers.bert.tokenization_bert import BertTokenizer
    # src/data_layer/dataset.py imports nltk, cv2, ... at module level (absent here, unused by this class): run only the
    # CaptionTensorizer class statement of the reference module, in place, with the two names it needs
    import torch
    src = open('/root/reference/src/data_layer/dataset.py').read()
    a = src.index('class CaptionTensorizer(object):')
    b = src.index('\nclass ', a + 10)
    ns = {'torch': torch, 'random': random}
    exec(compile(src[a:b], '/root/reference/src/data_layer/dataset.py', 'exec'), ns)
    CaptionTensorizer = ns['CaptionTensorizer']
    vocab = ['[PAD]'] + ['[unused%d]' % i for i in range(1, 100)] + ['[UNK]', '[CLS]', '[SEP]', '[MASK]'] + sorted(set(WORDS))
    d = tempfile.mkdtemp()
    vf = os.path.join(d, 'vocab.txt')
    with open(vf, 'w', encoding='utf-8') as fp:
        fp.write('\n'.join(vocab) + '\n')
    tok = BertTokenizer(vf, do_lower_case=True)
    out = {'vocab': vocab, 'texts': TEXTS, 'tokenize': [tok.tokenize(t) for t in TEXTS], 'cases': []}
    for ci, kw in enumerate(CASES):
        tz = CaptionTensorizer(tok, **kw)
        for ti, text in enumerate(TEXTS):
            for text_b in (None, 'dog cat table') if ti % 2 == 0 else (None,):
                seed = 1000 * ci + ti
                random.seed(seed)
                try:
                    r = tz.tensorize_ab(text, text_b)
                except KeyError:           # get_random_token drew len(vocab): the reference crashes there
                    continue
                after = random.random()
                out['cases'].append({'kw': kw, 'text': ti, 'text_b': text_b, 'seed': seed, 'rng_after': after,
                                     'out': {k: (pack(v) if k == 'attention_mask' else v.tolist()) for k, v in r.items()}})
    with open(os.path.join(HERE, 'reference_tensorizer.json'), 'w') as fp:
        json.dump(out, fp)
    print(len(out['cases']), 'cases;', out['tokenize'][2], out['tokenize'][4])


if __name__ == '__main__':
    main()
