"""Scorers: SCST rewards and the validator's metric list.  The reference's scorers (blocks/scorers/**) are CPU text metrics, most
of them wrappers over third-party models that are absent here (bert_score, radgraph, f1chexbert, rouge_score ...: SURVEY §2 row 18,
out of the hot path).  What needs no such package is provided: BLEU with the reference's vendored COCO-caption conventions (pinned
by fixture G15), ROUGE-1 / -2 / -L F-measures with rouge_score's tokenisation but WITHOUT its Porter stemmer (so values differ
from the reference's on inflected words), accuracy / f1-score / auroc for the classifiers; ``compute_scores`` mirrors
scorers/scores.py.  Any callable ``scorer(refs, hyps) -> (mean, per_sample_list)`` can be registered in REWARD_COMPLIANT."""
import re

import numpy as np

_tok = re.compile(r"[a-z0-9]+")


def _lcs(a, b):
    """length of the longest common subsequence, bit-parallel (Allison-Dix / Hyyro: one big-integer add, subtract, and, or per token of
    ``a`` instead of len(b) table cells).  The SCST step computes 2 x batch of these on the host between the rollouts and the
    policy-gradient pass -- with the quadratic table that was ~60 ms of a 200 ms step during which the GPU idles."""
    if not a or not b:
        return 0
    where = {}
    for j, y in enumerate(b):
        where[y] = where.get(y, 0) | (1 << j)
    full = (1 << len(b)) - 1
    v = full
    for x in a:
        u = v & where.get(x, 0)
        v = ((v + u) | (v - u)) & full
    return len(b) - bin(v).count("1")


class RougeL:
    """ROUGE-L F-measure on lower-cased alphanumeric tokens (rouge_score's tokenisation, without its Porter stemmer)."""

    def __call__(self, refs, hyps):
        if len(refs) != len(hyps):
            raise ValueError("Must have equal number of lines across target and prediction.")
        f = []
        for r, h in zip(refs, hyps):
            rt, ht = _tok.findall(r.lower()), _tok.findall(h.lower())
            l = _lcs(rt, ht)
            p, rc = (l / len(ht) if ht else 0.0), (l / len(rt) if rt else 0.0)
            f.append(2 * p * rc / (p + rc) if p + rc > 0 else 0.0)
        return float(np.mean(f)) if f else 0.0, f


def _ngrams(tokens, n):
    from collections import Counter
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


class _RougeN:
    """ROUGE-N F-measure (rouge_score's definition: clipped n-gram overlap; same tokenisation as RougeL above, no stemmer)"""
    n = 1

    def __call__(self, refs, hyps):
        if len(refs) != len(hyps):
            raise ValueError("Must have equal number of lines across target and prediction.")
        f = []
        for r, h in zip(refs, hyps):
            rc, hc = _ngrams(_tok.findall(r.lower()), self.n), _ngrams(_tok.findall(h.lower()), self.n)
            overlap = sum(min(c, hc.get(g, 0)) for g, c in rc.items())
            p, rc_ = (overlap / max(1, sum(hc.values()))), (overlap / max(1, sum(rc.values())))
            f.append(2 * p * rc_ / (p + rc_) if p + rc_ > 0 else 0.0)
        return float(np.mean(f)) if f else 0.0, f


class Rouge1(_RougeN):
    n = 1


class Rouge2(_RougeN):
    n = 2


class Bleu:
    """corpus BLEU-n with the COCO-caption scorer's conventions, as the reference calls it (blocks/scorers/NLG/bleu/bleu.py:24-46:
    one reference per hypothesis, whitespace tokens, effective reference length "closest", 1e-9 / 1e-15 smoothing constants,
    brevity penalty exp(1 - 1/ratio)); returns (corpus BLEU-n, per-sentence BLEU-n).  Pinned by tests/golden/g15_bleu.pt."""

    def __init__(self, n=4, **kwargs):
        self.n = n

    def __call__(self, refs, hyps):
        import math
        n, small, tiny = self.n, 1e-9, 1e-15
        tot_guess, tot_correct, tot_test, tot_ref, per = [0] * n, [0] * n, 0, 0, []
        for r, h in zip(refs, hyps):
            rw, hw = r.split(), h.split()
            guess = [max(0, len(hw) - k) for k in range(n)]
            correct = []
            for k in range(n):
                rc, hc = _ngrams(rw, k + 1), _ngrams(hw, k + 1)
                correct.append(sum(min(c, rc.get(g, 0)) for g, c in hc.items()))
            tot_test, tot_ref = tot_test + len(hw), tot_ref + len(rw)
            b = 1.0
            for k in range(n):
                tot_guess[k] += guess[k]
                tot_correct[k] += correct[k]
                b *= (correct[k] + tiny) / (guess[k] + small)
            b = b ** (1.0 / n)
            ratio = (len(hw) + tiny) / (len(rw) + small)
            per.append(b * math.exp(1 - 1 / ratio) if ratio < 1 else b)
        b = 1.0
        for k in range(n):
            b *= (tot_correct[k] + tiny) / (tot_guess[k] + small)
        b = b ** (1.0 / n)
        ratio = (tot_test + tiny) / (tot_ref + small)
        return (b * math.exp(1 - 1 / ratio) if ratio < 1 else b), per


REWARD_COMPLIANT = {"rougel": [RougeL, 1], "rouge1": [Rouge1, 1], "rouge2": [Rouge2, 1], "bleu": [Bleu, 1]}


def compute_scores(metrics, refs, hyps, split, seed, ckpt_dir, epoch, logger, dump=True):
    """ref: blocks/scorers/scores.py:34-151 -- the metrics that need no third-party model (BLEU, ROUGE-1/2/L, accuracy, f1-score,
    auroc); refs / hyps and the scores are dumped to ``{ckpt_dir}/{split}_{seed}_{refs,hyps,metrics}.txt`` like the reference does
    next to its log file.  Unknown or unavailable metrics are reported and skipped (the reference's own behaviour)."""
    import json
    import os
    scores = {}
    if not metrics:
        return scores
    assert refs is not None and hyps is not None, "You specified metrics but your evaluation does not return hyps nor refs"
    assert len(refs) == len(hyps), "refs and hyps must have same length: {} vs {}".format(len(refs), len(hyps))
    base = os.path.join(ckpt_dir or ".", "{}_{}_{{}}".format(split, seed))
    if dump and isinstance(refs, list):
        with open(base.format("refs.txt"), "w") as f:
            f.write("\n".join(map(str, refs)))
        with open(base.format("hyps.txt"), "w") as f:
            f.write("\n".join(map(str, hyps)))
    for metric in metrics:
        name = list(metric.keys())[0] if isinstance(metric, dict) else metric
        low = str(name).lower()
        try:
            if low == "bleu":
                scores["BLEU"] = Bleu()(refs, hyps)[0]
            elif low in ("rouge1", "rouge2", "rougel"):
                scores[str(name).upper()] = {"rouge1": Rouge1, "rouge2": Rouge2, "rougel": RougeL}[low]()(refs, hyps)[0]
            elif low == "accuracy":
                scores["accuracy"] = round(float(np.mean(np.array(refs) == np.argmax(hyps, axis=-1))) * 100, 2)
            elif low == "f1-score":
                from sklearn.metrics import classification_report
                scores["f1-score"] = classification_report(refs, np.argmax(hyps, axis=-1))
            elif low == "auroc":
                from scipy.special import softmax
                from sklearn.metrics import roc_auc_score
                scores["auroc"] = roc_auc_score(refs, softmax(np.asarray(hyps), axis=-1), multi_class="ovr")
            else:
                logger.warning("Metric not implemented: {}".format(name))
        except Exception as e:                   # a failing metric must not end the run (scores.py:141-143)
            logger.error("Error computing metric {}: {}".format(name, e))
            scores[str(name)] = None
    if dump:
        with open(base.format("metrics.txt"), "a+") as f:
            f.write(json.dumps({"split": split, "epoch": epoch, "scores": scores}, indent=4, sort_keys=False, default=str))
    return scores
