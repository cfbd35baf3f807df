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
    # ---- CaptionTaggerTensorizer (dataset.py:774-820), same in-place class statement; nltk is absent, so the 'nltk' branch
    # runs on a stand-in whose word_tokenize is str.split and whose pos_tag replays a fixed table (stored with the case):
    # what is pinned is the reference's use of the tags (JJ / NN / NNP words -> vocabulary ids), not nltk itself
    a = src.index('class CaptionTaggerTensorizer(object):')
    b = src.index('\nclass ', a + 10)
    POS = {'man': 'NN', 'horse': 'NN', 'red': 'JJ', 'street': 'NN', 'riding': 'VBG', 'a': 'DT', 'on': 'IN', 'the': 'DT',
           'pizza': 'NNP', 'two': 'CD', 'people': 'NNS', 'large': 'JJ', 'zzzz': 'NN'}

    class _Nltk(object):
        @staticmethod
        def word_tokenize(c):
            return c.split()

        @staticmethod
        def pos_tag(words):
            return [(w, POS.get(w, 'XX')) for w in words]
    ns2 = {'torch': torch, 'nltk': _Nltk}
    exec(compile(src[a:b], '/root/reference/src/data_layer/dataset.py', 'exec'), ns2)
    Tagger = ns2['CaptionTaggerTensorizer']
    labels = [{'class': 'dog', 'conf': 0.9}, {'class': 'tennis court', 'conf': 0.2}, {'class': 'cat', 'conf': 0.19},
              {'class': 'frisbee', 'conf': 0.8}]
    caps = ['a man riding a red horse on the street', 'two people pizza zzzz large', None]
    out['tagger'] = {'labels': labels, 'pos_table': POS, 'cases': []}
    for encode in ('nltk', 'bert', None):
        for caption_only in (False, True):
            for ci, cap in enumerate(caps):
                if cap is not None and encode is None:
                    continue               # the reference asserts encode is not None when a caption is given
                tz = Tagger(None, tok, threshold=0.2, category='bert', encode=encode, caption_only=caption_only)
                v = tz.tensorize(labels, cap)['label']
                out['tagger']['cases'].append({'encode': encode, 'caption_only': caption_only, 'caption': cap,
                                               'size': int(v.numel()), 'nonzero': v.nonzero().view(-1).tolist()})
    with open(os.path.join(HERE, 'reference_tensorizer.json'), 'w') as fp:
        json.dump(out, fp)
    print(len(out['cases']), 'cases;', out['tokenize'][2], out['tokenize'][4])


if __name__ == '__main__':
    main()
