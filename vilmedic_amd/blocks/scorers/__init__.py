"""Reward scorers for SCST.  The reference's scorers are CPU text metrics over third-party packages that are absent
here (rouge_score, bert_score, radgraph ...: SURVEY §2 row 18, OUT OF SCOPE); a dependency-free ROUGE-L F-measure is
provided as the default reward and any callable ``scorer(refs, hyps) -> (mean, per_sample_list)`` can be registered."""
import re

import numpy as np

_tok = re.compile(r"[a-z0-9]+")


def _lcs(a, b):
    if not a or not b:
        return 0
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b):
            cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[-1]


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


REWARD_COMPLIANT = {"rougel": [RougeL, 1]}
