"""Mirror of the parts of ecg_byte/utils/model_utils.py the generate/eval path uses.
BLEU is restated from NLTK's published `corpus_bleu` / `SmoothingFunction.method1` (nltk==3.9.1 is pinned in
the reference's requirements.txt but neither vendored nor installed here: parity unpinned, pinned only by
hand-computed known answers in tests/test_host_eval.py).  METEOR (WordNet), ROUGE (`rouge` package) and BERTScore
(a downloaded model) are out of scope: `evaluate_strings` reports BLEU and leaves the other keys to a caller-supplied hook."""
from __future__ import annotations

import math
from collections import Counter
from fractions import Fraction

import numpy as np


def count_parameters(model) -> int:
    """model_utils.py:14-15"""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def early_stopping(validation_losses, patience=5, delta=0):
    """model_utils.py:17-27"""
    if len(validation_losses) < patience + 1:
        return False
    best_loss = min(validation_losses[:-patience])
    return validation_losses[-1] > best_loss + delta


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25), epsilon=0.1):
    """NLTK `corpus_bleu(..., smoothing_function=SmoothingFunction().method1)`: clipped n-gram counts summed over the
    corpus, closest reference length, brevity penalty, zero numerators replaced by `epsilon` (method1), geometric mean."""
    num, den = Counter(), Counter()
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        for n in range(1, len(weights) + 1):
            counts = _ngrams(hyp, n)
            max_counts = {}
            for ref in refs:
                rc = _ngrams(ref, n)
                for g in counts:
                    max_counts[g] = max(max_counts.get(g, 0), rc[g])
            num[n] += sum(min(c, max_counts.get(g, 0)) for g, c in counts.items())
            den[n] += max(1, sum(counts.values()))
        hyp_len += len(hyp)
        ref_len += min((len(r) for r in refs), key=lambda rl: (abs(rl - len(hyp)), rl))
    if num[1] == 0:
        return 0
    if hyp_len > ref_len:
        bp = 1.0
    elif hyp_len == 0:
        bp = 0.0
    else:
        bp = math.exp(1 - ref_len / hyp_len)
    p_n = [Fraction(num[n], den[n]) if num[n] else (num[n] + epsilon) / den[n] for n in range(1, len(weights) + 1)]
    s = math.fsum(w * math.log(p) for w, p in zip(weights, p_n) if p > 0)
    return bp * math.exp(s)


def calculate_bleu(references, hypotheses):
    """model_utils.py:29-31: whitespace tokens, one reference per hypothesis."""
    return corpus_bleu([[r.split()] for r in references], [h.split() for h in hypotheses])


def evaluate_strings(references, hypotheses, device=None, extra_metrics=None):
    """model_utils.py:56-64.  `extra_metrics(references, hypotheses) -> dict` may add METEOR / ROUGE / BERTSCORE
    entries in the reference's shapes; without it only BLEU is reported."""
    if len(references) != len(hypotheses):
        raise ValueError("The number of references and hypotheses must be the same.")
    out = {"BLEU": calculate_bleu(references, hypotheses)}
    if extra_metrics is not None:
        out.update(extra_metrics(references, hypotheses))
    return out


def run_statistical_analysis(all_seeds_results):
    """model_utils.py:68-92: per metric over the seeds, values x100: mean, sample std, 95 % Student-t interval."""
    from scipy import stats
    metrics = list(all_seeds_results[0]["metrics"].keys())
    statistical_results = {}
    for metric in metrics:
        values = [result["metrics"][metric] * 100 for result in all_seeds_results]
        mean = np.mean(values)
        std = np.std(values, ddof=1)
        t_value = stats.t.ppf((1 + 0.95) / 2, len(values) - 1)
        margin = t_value * (std / np.sqrt(len(values)))
        statistical_results[metric] = {"mean": mean, "std": std, "conf_interval": (mean - margin, mean + margin),
                                       "raw_values": values}
    return statistical_results
