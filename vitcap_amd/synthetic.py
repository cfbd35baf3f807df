"""Synthetic training batches with the layout the reference's training collate produces."""
import torch

VOCAB = 30522
BOS, EOS, PAD, MASK = 101, 102, 0, 103


def synthetic_train_inputs(B, seed=4321, max_seq_a=20, max_seq=70, n_mask=3):
    """What CaptionTensorizer.tensorize_ab(is_train=True, text_b='') emits (dataset.py:206-417), on synthetic captions:
    [CLS] w1..wn [SEP] PAD.., n in [8,18]; 3 caption positions replaced by [MASK] (masked_ids = the originals);
    attention_mask = tril over the seq_a_len caption slots, zero elsewhere; label = multi-hot of 10 random ids."""
    g = torch.Generator().manual_seed(seed)
    input_ids = torch.zeros(B, max_seq, dtype=torch.long)
    attention_mask = torch.zeros(B, max_seq, max_seq)
    masked_pos = torch.zeros(B, max_seq, dtype=torch.int32)
    masked_ids = torch.zeros(B, n_mask, dtype=torch.long)
    label = torch.zeros(B, VOCAB)
    for b in range(B):
        n = int(torch.randint(8, 19, (1,), generator=g))
        toks = torch.randint(1000, 30000, (n,), generator=g)
        seq_a_len = n + 2
        input_ids[b, 0] = BOS
        input_ids[b, 1:n + 1] = toks
        input_ids[b, n + 1] = EOS
        pos = torch.randperm(n, generator=g)[:n_mask].sort().values + 1
        masked_ids[b] = input_ids[b, pos]
        input_ids[b, pos] = MASK
        masked_pos[b, pos] = 1
        attention_mask[b, :seq_a_len, :seq_a_len] = torch.tril(torch.ones(seq_a_len, seq_a_len))
        label[b, torch.randint(1000, 30000, (10,), generator=g)] = 1
    return {'input_ids': input_ids, 'attention_mask': attention_mask, 'masked_pos': masked_pos,
            'masked_ids': masked_ids, 'label': label, 'token_type_ids': torch.zeros(B, max_seq, dtype=torch.long)}


