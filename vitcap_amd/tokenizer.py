"""ids -> caption string, the part of BertTokenizer the captioning output needs
(src/layers/bert/tokenization_utils.py:430-471, 506-510; tokenization_bert.py:188-191):
skip special ids, join WordPiece tokens, merge ``##`` continuations, clean up punctuation spacing."""

SPECIAL = ('[UNK]', '[SEP]', '[PAD]', '[CLS]', '[MASK]')


def clean_up_tokenization(s):
    return (s.replace(' .', '.').replace(' ?', '?').replace(' !', '!').replace(' ,', ',').replace(" ' ", "'")
            .replace(" n't", "n't").replace(" 'm", "'m").replace(" do not", " don't").replace(" 's", "'s")
            .replace(" 've", "'ve").replace(" 're", "'re"))


class CaptionDetokenizer(object):
    def __init__(self, vocab_file=None, tokens=None):
        if tokens is None:
            with open(vocab_file, 'r', encoding='utf-8') as fp:
                tokens = [line.rstrip('\n') for line in fp]
        self.ids_to_tokens = list(tokens)
        self.vocab = {t: i for i, t in enumerate(self.ids_to_tokens)}
        self.all_special_ids = set(self.vocab[t] for t in SPECIAL if t in self.vocab)

    def convert_tokens_to_ids(self, toks):
        unk = self.vocab.get('[UNK]', 100)
        return [self.vocab.get(t, unk) for t in toks]

    def decode(self, token_ids, skip_special_tokens=False, clean_up_tokenization_spaces=True):
        toks = [self.ids_to_tokens[i] for i in token_ids
                if not (skip_special_tokens and i in self.all_special_ids)]
        text = ' '.join(toks).replace(' ##', '').strip()
        return clean_up_tokenization(text) if clean_up_tokenization_spaces else text
