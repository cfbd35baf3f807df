"""Training / test text tensors of the captioning path: CaptionTensorizer.tensorize_ab of the reference
(src/data_layer/dataset.py:158-417) for the mask types the shipped pipeline uses (seq2seq, seq2seq_off, bidirectional).

Layout: [CLS] caption [SEP] <pad to max_seq_a_len when a text_b follows> text_b [SEP] <pad to max_seq_len>.
Training picks min(max(round(mask_prob * len_a), 1), max_masked_tokens) positions among the caption tokens and [SEP]
(never [CLS] / padding), replaces 80 % by [MASK], 10 % by a random token, keeps 10 %; the draws come from Python's
`random` in the reference's order (shuffle of the candidate list, then per position random() and, if that exceeds 0.8, a
second random()), so a seeded run reproduces the reference's batches bit for bit (tests/golden/reference_tensorizer.json).
The attention mask is the (L, L) matrix `construct_attn_mask` later extends with the visual tokens."""
import random

import torch


class CaptionTensorizer(object):
    def __init__(self, tokenizer, max_img_seq_length=50, max_seq_length=70, max_seq_a_length=40, mask_prob=0.15,
                 max_masked_tokens=3, mask_type='seq2seq', is_train=True, mask_b=False, replace_by_mask_prob=0.8,
                 replace_by_rand_prob=0.1, ignore_sep=False):
        if mask_type not in ('seq2seq', 'seq2seq_off', 'bidirectional'):
            raise NotImplementedError('mask_type %s' % mask_type)
        self.tok = tokenizer
        self.max_img_seq_len, self.L, self.La = max_img_seq_length, max_seq_length, max_seq_a_length
        self.mask_prob, self.max_masked = mask_prob, max_masked_tokens
        self.mask_type, self.is_train, self.mask_b, self.ignore_sep = mask_type, is_train, mask_b, ignore_sep
        self.p_mask, self.p_rand = replace_by_mask_prob, replace_by_rand_prob

    def tensorize_ab(self, text_a, text_b=None, pad_to_max=True, real_text_a_in_test=True, rng=None):
        """`rng`: a random.Random for the masking draws (default: Python's global generator, as in the reference)."""
        t = self.tok
        R = rng if rng is not None else random
        if not real_text_a_in_test and not self.is_train:
            ta = [t.mask_token] * (self.La - 2)
        else:
            ta = t.tokenize(text_a)[:self.La - 2]
        toks = [t.cls_token] + ta + [t.sep_token]
        seg = [0] * len(toks)
        len_a = padded_a = len(toks)
        if text_b:
            if pad_to_max:
                toks += [t.pad_token] * (self.La - len_a)
                seg += [0] * (self.La - len_a)
                padded_a = self.La
            tb = t.tokenize(text_b)[:max(self.L - len(toks) - 1, 0)]
            toks += tb + [t.sep_token]
            seg += [1] * (len(tb) + 1)
        n_real = len(toks)
        if pad_to_max:
            toks += [t.pad_token] * (self.L - n_real)
            seg += [0] * (self.L - n_real)
        origin = torch.tensor(t.convert_tokens_to_ids(toks), dtype=torch.long)
        n = len(toks)
        out = {}
        if self.is_train:
            if self.mask_b:
                cand = list(range(1, len_a)) + list(range(padded_a, n_real))
                k = min(max(round(self.mask_prob * n_real), 1), self.max_masked)
            elif not self.ignore_sep:
                cand = list(range(1, len_a))
                k = min(max(round(self.mask_prob * len_a), 1), self.max_masked)
            else:
                cand = list(range(1, len_a - 1))
                k = min(max(round(self.mask_prob * (len_a - 1)), 1), self.max_masked)
            if self.mask_prob == 0:
                k = 0
            R.shuffle(cand)
            picked = sorted(cand[:int(k)])
            targets = [toks[i] for i in picked]
            for i in picked:
                if R.random() <= self.p_mask:
                    toks[i] = t.mask_token
                elif R.random() <= self.p_rand / (1 - self.p_mask):
                    toks[i] = t.get_random_token(R) if rng is not None else t.get_random_token()
            masked_pos = torch.zeros(n, dtype=torch.int)
            masked_pos[picked] = 1
            if len(picked) < self.max_masked and pad_to_max:
                targets = targets + [t.pad_token] * (self.max_masked - len(picked))
            out['masked_ids'] = torch.tensor(t.convert_tokens_to_ids(targets), dtype=torch.long)
            out['origin_input_ids'] = origin
        else:
            masked_pos = torch.ones(n, dtype=torch.int)
        if self.mask_type == 'bidirectional':
            am = torch.zeros(n, dtype=torch.long)
            am[:len_a] = 1
            am[padded_a:n_real] = 1
        else:
            am = torch.zeros((n, n), dtype=torch.long)
            tri = torch.tril(torch.ones((len_a, len_a), dtype=torch.long))
            if self.mask_type == 'seq2seq_off':        # a token does not see itself (except [CLS])
                idx = torch.arange(1, len_a)
                tri[idx, idx] = 0
            am[:len_a, :len_a] = tri
            am[padded_a:n_real, padded_a:n_real] = 1    # text_b among itself
            am[:len_a, padded_a:n_real] = 1             # caption sees text_b
        out.update(input_ids=torch.tensor(t.convert_tokens_to_ids(toks), dtype=torch.long), attention_mask=am,
                   segment_ids=torch.tensor(seg, dtype=torch.long), masked_pos=masked_pos)
        return out
