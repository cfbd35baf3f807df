"""Constrained beam search, host side: the finite-state machines ViTCAP.generate(use_cbs=True) consumes.

Mirrors the interface of the reference's src/tools/captioning/utils_cbs.py for the part the captioning path needs:

* ``load_wordforms``                (utils_cbs.py:446-452)   TSV ``name<TAB>form1,form2`` -> dict
* ``FiniteStateMachineBuilder``     (utils_cbs.py:646-871)   constraints -> adjacency tensor ``fsm[s1, s2, word]`` (uint8)
* ``batch_fsm``                     what a caller has to hand to ``generate``: the machines of a batch trimmed to the sub-states in
                                    use ("dynamically trim unused sub-states", utils_cbs.py:685) and stacked, plus ``num_constraints``

The search itself (``ConstrainedBeamSearch.search``, ``select_best_beam_with_constraints``, utils_cbs.py:26-443) runs on the device:
csrc/cbs.hip behind ``vitcap_gen_opts.use_cbs`` (vitcap_amd.model.ImageCaptioning.generate_cbs).  Not mirrored: ``ConstraintFilter`` /
``ConstraintBoxesReader`` (utils_cbs.py:455-643) -- they turn detector boxes into constraint names with the Open Images class hierarchy
(``anytree``); nothing on the captioning path calls them.

State layout (utils_cbs.py:672-697): main states 0 .. 2**k - 1, bit n-1 of a main state = constraint n satisfied; every word of a
multi-word constraint but the last leads to a sub-state (numbered from 2**k on, in the order they are created) that falls back to the
chain's main state on any other word."""
import numpy as np
import torch


def load_wordforms(wordforms_tsvpath):
    out = {}
    with open(wordforms_tsvpath, 'r') as fp:
        for line in fp:
            parts = line.strip().split('\t')
            if len(parts) >= 2:
                out[parts[0]] = parts[1].split(',')
    return out


class FiniteStateMachineBuilder(object):
    """Same constructor and ``build`` contract as the reference's class; ``tokenizer`` needs ``vocab_size`` and
    ``convert_tokens_to_ids``.  The TSV paths may also be dicts already loaded."""

    def __init__(self, tokenizer, constraint2tokens_tsvpath, tokenforms_tsvpath, max_given_constraints, max_words_per_constraint=4):
        self._tokenizer = tokenizer
        self._max_given_constraints = int(max_given_constraints)
        self._max_words_per_constraint = int(max_words_per_constraint)
        self._num_main_states = 2 ** self._max_given_constraints
        self._num_total_states = self._num_main_states * self._max_words_per_constraint
        as_dict = lambda x: dict(x) if isinstance(x, dict) else load_wordforms(x)
        self._wordforms = as_dict(tokenforms_tsvpath)
        self._constraint2tokens = as_dict(constraint2tokens_tsvpath)

    def _form_ids(self, word):
        return [int(i) for i in self._tokenizer.convert_tokens_to_ids(self._wordforms.get(word, [word]))]

    def build(self, constraints):
        """constraints: up to ``max_given_constraints`` class names (possibly several words each).  Returns (fsm uint8
        (T, T, vocab_size), index of the next unused sub-state)."""
        if len(constraints) > self._max_given_constraints:
            raise AssertionError('%d constraints given, at most %d supported' % (len(constraints), self._max_given_constraints))
        nmain, T, V = self._num_main_states, self._num_total_states, int(self._tokenizer.vocab_size)
        fsm = np.zeros((T, T, V), dtype=np.uint8)
        for s in range(nmain):
            fsm[s, s, :] = 1                       # every word loops on a main state until a constraint claims it
        sub = nmain
        for n, constraint in enumerate(constraints, start=1):
            words = []
            for w in constraint.split():
                words.extend(self._constraint2tokens[w])
            words = words[:self._max_words_per_constraint]
            forms = [self._form_ids(w) for w in words]
            stride = 2 ** (n - 1)
            frm = 0
            while frm < nmain:                     # every main state that lacks bit n-1 ...
                for _ in range(stride):
                    cur = frm
                    for i, ids in enumerate(forms):
                        last = i == len(forms) - 1
                        to = frm + stride if last else sub     # ... reaches the one that has it through one sub-state per extra word
                        self._connect(fsm, cur, to, ids, frm)
                        if not last:
                            cur = sub
                            sub += 1
                    frm += 1
                frm += stride
        return torch.from_numpy(fsm), sub

    @staticmethod
    def _connect(fsm, frm, to, ids, reset):
        # utils_cbs.py:822-871, in its order of assignments (a later constraint re-opens the self-loop of an earlier one's words on
        # the states it touches: the reference does the same)
        for i in ids:
            fsm[frm, to, i] = 1
            fsm[frm, frm, i] = 0
        fsm[frm, frm, :] = 0
        fsm[frm, reset, :] = 1
        for i in ids:
            fsm[frm, reset, i] = 0


def batch_fsm(builder, constraints_per_image, device=None):
    """The ``fsm`` / ``num_constraints`` pair of a batch: every image's machine trimmed to the largest sub-state count in use and
    stacked -> (fsm uint8 (B, S, S, V), num_constraints int64 (B,))."""
    built = [builder.build(list(c)) for c in constraints_per_image]
    S = max(n for _, n in built)
    fsm = torch.stack([f[:S, :S] for f, _ in built]).contiguous()
    num = torch.tensor([len(c) for c in constraints_per_image], dtype=torch.int64)
    if device is not None:
        fsm, num = fsm.to(device), num.to(device)
    return fsm, num
