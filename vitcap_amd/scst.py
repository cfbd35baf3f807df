"""Self-critical sequence training step (BASELINE config 5) on the HIP engines.

Reference behaviour (spec only: src/pipelines/tagger_caption_uni_pipeline_expanding.py:404-478 is not importable as shipped;
generation src/layers/bert/modeling_utils.py:768-886; criterion src/tools/captioning/utils_caption_evaluate.py:162-237):
  1. greedy captions with the model in eval mode, no gradient -> baseline;
  2. `scst_num_return` sampled captions per image (do_sample, temperature 1, no top-k/top-p) WITH gradient through the
     per-sequence mean log-probability;
  3. reward = CIDEr-D(sample) - CIDEr-D(greedy of the same image), computed on the host from the ground-truth captions;
  4. loss = -mean(reward * logprob); backward, clip, AdamW as in the cross-entropy step.

Here 1 and 2 run on the inference engine (sampling = counter-based Gumbel-max draws, csrc/rng.h), 4 on the training engine
in ONE teacher-forced pass over [visual | 20 token rows | 19 [MASK] probe rows] per sampled sequence
(TrainEngine.forward_backward with `sample_ids`), whose loss and gradient equal the generator's 19 differentiated
forwards (tests/test_hip_train_e2e.py::test_scst_logprob_gradient_vs_oracle).

Deviations, stated: the reference samples with the decoder's attention dropout active (train mode) and differentiates the
very same stochastic forwards; here the samples are drawn without dropout and the gradient pass uses the engine's
`attn_dropout` setting.  The CIDEr-D scorer below restates the published algorithm (Vedantam et al. 2015; coco-caption's
pyciderevalcap `CiderD`, n = 4, sigma = 6, x10) because that package is not vendored by the reference either (README:24):
reward VALUES are therefore not pinned against it, only the reward -> loss -> gradient path is."""
import math
from collections import Counter, defaultdict

import torch


# ------------------------------------------------------------------------------------------------------------------
class CiderD(object):
    """CIDEr-D with document frequencies taken from the references passed in ('corpus' mode) or from a given table."""

    def __init__(self, n=4, sigma=6.0, df=None, ref_len=None):
        self.n, self.sigma = n, sigma
        self.df, self.ref_len = df, ref_len

    @staticmethod
    def _ngrams(sentence, n):
        w = sentence.split()
        c = Counter()
        for k in range(1, n + 1):
            for i in range(len(w) - k + 1):
                c[tuple(w[i:i + k])] += 1
        return c

    def _vec(self, counts, df, ref_len):
        vec = [defaultdict(float) for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for ng, tf in counts.items():
            k = len(ng) - 1
            d = math.log(max(1.0, df.get(ng, 0.0)))
            vec[k][ng] = float(tf) * (ref_len - d)
            norm[k] += vec[k][ng] ** 2
            if k == 0:
                length += tf
        return vec, [math.sqrt(v) for v in norm], length

    def _sim(self, vh, vr, nh, nr, lh, lr):
        delta = float(lh - lr)
        val = [0.0] * self.n
        for k in range(self.n):
            for ng, v in vh[k].items():
                val[k] += min(v, vr[k].get(ng, 0.0)) * vr[k].get(ng, 0.0)
            if nh[k] != 0 and nr[k] != 0:
                val[k] /= nh[k] * nr[k]
            val[k] *= math.exp(-(delta ** 2) / (2 * self.sigma ** 2))
        return val

    def compute_score(self, gts, res):
        """gts: list (per hypothesis) of lists of reference strings; res: list of hypothesis strings -> (mean, per-hyp)."""
        refs = [[self._ngrams(r, self.n) for r in rs] for rs in gts]
        if self.df is None:
            df = defaultdict(float)
            for rs in refs:
                for ng in set(ng for r in rs for ng in r):
                    df[ng] += 1
            ref_len = math.log(float(len(refs)))
        else:
            df, ref_len = self.df, self.ref_len
        scores = []
        for hyp, rs in zip(res, refs):
            vh, nh, lh = self._vec(self._ngrams(hyp, self.n), df, ref_len)
            tot = [0.0] * self.n
            for r in rs:
                vr, nr, lr = self._vec(r, df, ref_len)
                s = self._sim(vh, vr, nh, nr, lh, lr)
                tot = [a + b for a, b in zip(tot, s)]
            scores.append(sum(tot) / self.n / max(1, len(rs)) * 10.0)
        return (sum(scores) / max(1, len(scores))), scores


def corpus_bleu(refs, hyps, n=4):
    """Corpus-level BLEU-1..n (Papineni et al. 2002): clipped n-gram counts summed over the corpus, brevity penalty from the
    closest reference length per hypothesis.  refs: per hypothesis a list of reference strings; whitespace tokens.  Used by the
    pipeline's evaluate() report (parity-unpinned, like CiderD: the reference's scorer is the external coco_caption package)."""
    match, total = [0] * n, [0] * n
    hyp_len = ref_len = 0
    for rs, h in zip(refs, hyps):
        hw = h.split()
        rws = [r.split() for r in rs]
        hyp_len += len(hw)
        ref_len += min((abs(len(r) - len(hw)), len(r)) for r in rws)[1]
        for k in range(1, n + 1):
            hc = Counter(tuple(hw[i:i + k]) for i in range(len(hw) - k + 1))
            mx = Counter()
            for r in rws:
                rc = Counter(tuple(r[i:i + k]) for i in range(len(r) - k + 1))
                for ng, c in rc.items():
                    mx[ng] = max(mx[ng], c)
            match[k - 1] += sum(min(c, mx[ng]) for ng, c in hc.items())
            total[k - 1] += max(0, len(hw) - k + 1)
    bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - float(ref_len) / max(1, hyp_len))
    out, logsum = [], 0.0
    for k in range(n):
        p = (match[k] + 1e-15) / (total[k] + 1e-9)
        logsum += math.log(p)
        out.append(bp * math.exp(logsum / (k + 1)))
    return out


def _wrap(s):
    """ScstRewardCriterion._wrap_sentence: strip, drop a final period, append ' <eos>'."""
    r = s.strip()
    if r.endswith('.'):
        r = r[:-1]
    return r + ' <eos>'


def scst_rewards(gt_captions, greedy_captions, sample_captions, scorer=None):
    """reward[i] = score(sample i) - score(greedy caption of its image)  (baseline_type 'greedy'); returns (reward (N,), mean
    sample score).  gt_captions: per image a list of strings; sample_captions: K per image, image-major."""
    B, N = len(gt_captions), len(sample_captions)
    assert len(greedy_captions) == B and N % B == 0
    K = N // B
    scorer = scorer or CiderD()
    gts = [[_wrap(c) for c in gt_captions[i // K]] for i in range(N)] + [[_wrap(c) for c in g] for g in gt_captions]
    res = [_wrap(c) for c in sample_captions] + [_wrap(c) for c in greedy_captions]
    _, sc = scorer.compute_score(gts, res)
    s = torch.tensor(sc[:N], dtype=torch.float32).view(B, K)
    base = torch.tensor(sc[N:], dtype=torch.float32).view(B, 1)
    return (s - base).view(N), float(s.mean())


# ------------------------------------------------------------------------------------------------------------------
class ScstTrainer(object):
    """One self-critical iteration per `step(images, gt_captions)`; `engine` is the TrainEngine of `model`."""

    def __init__(self, model, engine, tokenizer, num_return=5, seed=0, scorer=None):
        self.model, self.eng, self.tok = model, engine, tokenizer
        self.K, self.seed, self.scorer = int(num_return), int(seed), scorer
        self.iter = 0
        self.last_score = None

    def _sync_inference_weights(self):
        """The generator must see the current masters: the inference engine is bound (once, zero-copy) to the training
        engine's own bf16 matrices / fp32 vectors, which every optimizer step refreshes."""
        if not getattr(self, '_bound', False):
            self.eng.bind_inference()
            self._bound = True

    def _decode(self, ids):
        return [self.tok.decode(r.tolist(), skip_special_tokens=True) for r in ids]

    def step(self, images, gt_captions):
        """images (B,3,384,384) on the GPU; gt_captions: per image a list of strings.  Returns {'scst_loss', 'score'}."""
        B, K = images.shape[0], self.K
        self._sync_inference_weights()
        self.model.eval()
        g_ids, _ = self.model.generate(images)
        g_ids = g_ids.clone()
        # the K samples of an image share its encoder pass and visual K/V (generation) and its encoder forward / backward
        # (training): what the reference computes on K-times expanded inputs, without the K-fold repetition of the ViT
        seed = (self.seed + 0x9e3779b1 * self.iter) & 0xffffffff
        samp = dict(temperature=1.0, top_k=0, top_p=1.0, seed=seed)
        if K <= 8:
            s_ids, _, raw_last = self.model.generate_multi(images, K, want_last=True, **samp)
        else:
            s_ids, _, raw_last = self.model.generate_multi(images.repeat_interleave(K, 0).contiguous(), 1, want_last=True, **samp)
        s_ids = s_ids[:, 0].clone()
        reward, score = scst_rewards(gt_captions, self._decode(g_ids[:, 0].cpu()), self._decode(s_ids.cpu()), self.scorer)
        fed = s_ids.clone()
        fed[:, -1] = raw_last            # the token whose log-prob the generator recorded, not the forced [SEP]
        loss, _ = self.eng.forward_backward({'image': images, 'seq_per_image': K, 'sample_ids': fed,
                                             'sample_weight': (reward / float(B * K)).to(images.device)})
        self.eng.all_reduce_grads()
        self.eng.optimizer_step()
        self.iter += 1
        self.last_score = score
        return {'scst_loss': loss, 'score': score}
