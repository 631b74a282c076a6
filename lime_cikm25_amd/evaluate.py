"""MIND-style ranking metrics from rank files: a restatement of the reference's evaluate.py:7-89 (SURVEY.md section 8f
row 1).  Host-side logic (numpy); the scores it consumes come from the HIP path.

Pinned by tests/golden/eval_*.json, produced by running the reference's own ``util.compute_scores`` and
``evaluate.scoring`` (tools/make_eval_goldens.py).  AUC is computed the way sklearn's ``roc_auc_score`` does for a
binary target (ROC curve at the distinct score thresholds, trapezoidal area), without importing sklearn.
"""
import json

import numpy as np


def _by_score(y_true, y_score, k=None):
    """Labels reordered by descending score (numpy's default sort reversed, as the reference does: evaluate.py:8, :22)."""
    ranked = np.take(y_true, np.argsort(y_score)[::-1][:k])
    return ranked, np.arange(ranked.shape[0])


def dcg_score(y_true, y_score, k=10):
    """sum_i (2^rel_i - 1) / log2(i + 2) over the k best-scored items (evaluate.py:7-12)."""
    rel, pos = _by_score(y_true, y_score, k)
    return np.sum((2 ** rel - 1) / np.log2(pos + 2))


def ndcg_score(y_true, y_score, k=10):
    """DCG normalised by the DCG of the ideal order (evaluate.py:15-18)."""
    ideal = dcg_score(y_true, y_true, k)
    return dcg_score(y_true, y_score, k) / ideal


def mrr_score(y_true, y_score):
    """Sum of label / rank over the label mass (evaluate.py:21-25)."""
    rel, pos = _by_score(y_true, y_score)
    return np.sum(rel / (pos + 1)) / np.sum(y_true)


def roc_auc_score(y_true, y_score):
    """Binary ROC AUC as sklearn computes it (evaluate.py:77): thresholds at the distinct scores, trapezoidal rule."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_score = np.asarray(y_score, dtype=np.float64)
    if np.unique(y_true).size != 2:
        raise ValueError('Only one class present in y_true. ROC AUC score is not defined in that case.')
    pos = y_true == y_true.max()
    order = np.argsort(y_score, kind='mergesort')[::-1]
    y_score, pos = y_score[order], pos[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, pos.size - 1]
    tps = np.cumsum(pos)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    tps, fps = np.r_[0.0, tps], np.r_[0.0, fps]
    tpr, fpr = tps / tps[-1], fps / fps[-1]
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))          # trapezoidal rule


def parse_line(line):
    """'<impression id> [v1,v2,...]' -> (id, list)  (evaluate.py:27-30)."""
    impid, payload = line.strip('\n').split()
    return impid, json.loads(payload)


def _rank_scores(ranks, n_labels, line_no):
    """A rank file stores 1-based ranks; the metrics consume 1 / rank (evaluate.py:60-66)."""
    out = []
    for r in ranks:
        v = 1. / r
        if not 0 <= v <= 1:
            raise ValueError('Line-{}: score_rslt should be int from 0 to {}'.format(line_no, float(n_labels)))
        out.append(v)
    return out


def scoring(truth_f, sub_f):
    """(AUC, MRR, nDCG@5, nDCG@10) averaged over the impressions of a truth file and a rank file read in lock step
    (evaluate.py:32-89): impressions with an empty label list are skipped, a missing submission line counts as all-ones."""
    per_impression = []
    line_no = 1
    for truth_line in truth_f:
        sub_line = sub_f.readline()
        impid, labels = parse_line(truth_line)
        if not labels:                          # masked impression (the submission line is consumed all the same)
            continue
        if sub_line == '':
            sub_id, ranks = impid, [1] * len(labels)
        else:
            try:
                sub_id, ranks = parse_line(sub_line)
            except Exception:
                raise ValueError('line-{}: Invalid Input Format!'.format(line_no))
        if sub_id != impid:
            raise ValueError('line-{}: Inconsistent Impression Id {} and {}'.format(line_no, sub_id, impid))
        y_true = np.array(labels, dtype='float32')
        y_score = _rank_scores(ranks, len(labels), line_no)
        per_impression.append((roc_auc_score(y_true, y_score), mrr_score(y_true, y_score), ndcg_score(y_true, y_score, 5),
                               ndcg_score(y_true, y_score, 10)))
        line_no += 1
    cols = list(zip(*per_impression)) if per_impression else [[], [], [], []]
    return tuple(np.mean(c) for c in cols)
