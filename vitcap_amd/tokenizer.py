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


# ------------------------------------------------------------------------------------------------------------------
# text -> WordPiece tokens (training-side tensorizer).  Same published algorithm as the reference's
# BertTokenizer (src/layers/bert/tokenization_bert.py:97-420, tokenization_utils.py:364-428): split on special tokens,
# clean control characters / whitespace, space out CJK ideographs, lower-case + strip accents (NFD, drop Mn),
# split punctuation, then greedy longest-match-first WordPiece with '##' continuations (words > 100 chars -> [UNK]).
import random as _random
import unicodedata as _ud

_CJK = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF),
        (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _is_space(ch):
    return ch in ' \t\n\r' or _ud.category(ch) == 'Zs'


def _is_ctrl(ch):
    return ch not in '\t\n\r' and _ud.category(ch).startswith('C')


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True                     # all non-alphanumeric ASCII counts, e.g. '^', '$', '`'
    return _ud.category(ch).startswith('P')


class BertWordPieceTokenizer(CaptionDetokenizer):
    cls_token, sep_token, pad_token, mask_token, unk_token = '[CLS]', '[SEP]', '[PAD]', '[MASK]', '[UNK]'

    def __init__(self, vocab_file=None, tokens=None, do_lower_case=True, max_chars_per_word=100):
        super(BertWordPieceTokenizer, self).__init__(vocab_file, tokens)
        self.do_lower_case = do_lower_case
        self.max_chars_per_word = max_chars_per_word
        self.vocab_size = len(self.vocab)

    # -- stage 1: words
    def _words(self, text):
        buf = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_ctrl(ch):
                continue
            if _is_space(ch):
                buf.append(' ')
            elif any(lo <= cp <= hi for lo, hi in _CJK):
                buf.append(' ' + ch + ' ')
            else:
                buf.append(ch)
        out = []
        for w in ''.join(buf).split():
            if self.do_lower_case:
                w = ''.join(c for c in _ud.normalize('NFD', w.lower()) if _ud.category(c) != 'Mn')
            piece = ''
            for c in w:
                if _is_punct(c):
                    if piece:
                        out.append(piece)
                        piece = ''
                    out.append(c)
                else:
                    piece += c
            if piece:
                out.append(piece)
        return out

    # -- stage 2: word pieces
    def _pieces(self, word):
        if len(word) > self.max_chars_per_word:
            return [self.unk_token]
        res, start = [], 0
        while start < len(word):
            end = len(word)
            while end > start:
                cand = ('##' if start else '') + word[start:end]
                if cand in self.vocab:
                    break
                end -= 1
            if end == start:
                return [self.unk_token]
            res.append(cand)
            start = end
        return res

    def tokenize(self, text):
        specials = [t for t in SPECIAL if t in text]
        chunks = [text]
        for sp in specials:                       # special tokens written literally in the text survive as they are
            nxt = []
            for c in chunks:
                if c in SPECIAL:
                    nxt.append(c)
                    continue
                parts = c.split(sp)
                for i, p_ in enumerate(parts):
                    if p_.strip():
                        nxt.append(p_.strip())
                    if i + 1 < len(parts):
                        nxt.append(sp)
            chunks = nxt
        out = []
        for c in chunks:
            if c in SPECIAL:
                out.append(c)
            else:
                for w in self._words(c):
                    out.extend(self._pieces(w))
        return out

    def get_random_token(self, rng=None):
        """tokenization_bert.py:208-210 draws randint(0, len(vocab)) -- inclusive, so one draw in 30 523 indexes past the
        vocabulary (a KeyError upstream); that draw maps to [UNK] here, every other draw is identical."""
        i = (rng or _random).randint(0, len(self.vocab))
        return self.ids_to_tokens[i] if i < len(self.ids_to_tokens) else self.unk_token
