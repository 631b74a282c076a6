"""MIND-style ranking metrics from rank files: a restatement of the reference's evaluate.py:7-89 (SURVEY.md section 8f
row 1).  Host-side logic (numpy); the scores it consumes come from the HIP path.

Pinned by tests/golden/eval_*.json, produced by running the reference's own ``util.compute_scores`` and
``evaluate.scoring`` (tools/make_eval_goldens.py).  AUC is computed the way sklearn's ``roc_auc_score`` does for a
binary target (ROC curve at the distinct score thresholds, trapezoidal area), without importing sklearn.
"""
import json

import numpy as np


def dcg_score(y_true, y_score, k=10):
    """evaluate.py:7-12."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    gains = 2 ** y_true - 1
    discounts = np.log2(np.arange(len(y_true)) + 2)
    return np.sum(gains / discounts)


def ndcg_score(y_true, y_score, k=10):
    """evaluate.py:15-18."""
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true, y_score):
    """evaluate.py:21-25."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    rr_score = y_true / (np.arange(len(y_true)) + 1)
    return np.sum(rr_score) / np.sum(y_true)


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
    """evaluate.py:27-30."""
    impid, ranks = line.strip('\n').split()
    return impid, json.loads(ranks)


def scoring(truth_f, sub_f):
    """evaluate.py:32-89: (AUC, MRR, nDCG@5, nDCG@10) averaged over the impressions of a truth file and a rank file."""
    aucs, mrrs, ndcg5s, ndcg10s = [], [], [], []
    line_index = 1
    for lt in truth_f:
        ls = sub_f.readline()
        impid, labels = parse_line(lt)
        if labels == []:                       # masked impression
            continue
        if ls == '':
            sub_impid, sub_ranks = impid, [1] * len(labels)
        else:
            try:
                sub_impid, sub_ranks = parse_line(ls)
            except Exception:
                raise ValueError('line-{}: Invalid Input Format!'.format(line_index))
        if sub_impid != impid:
            raise ValueError('line-{}: Inconsistent Impression Id {} and {}'.format(line_index, sub_impid, impid))
        y_true = np.array(labels, dtype='float32')
        y_score = []
        for rank in sub_ranks:
            score_rslt = 1. / rank
            if score_rslt < 0 or score_rslt > 1:
                raise ValueError('Line-{}: score_rslt should be int from 0 to {}'.format(line_index, float(len(labels))))
            y_score.append(score_rslt)
        aucs.append(roc_auc_score(y_true, y_score))
        mrrs.append(mrr_score(y_true, y_score))
        ndcg5s.append(ndcg_score(y_true, y_score, 5))
        ndcg10s.append(ndcg_score(y_true, y_score, 10))
        line_index += 1
    return np.mean(aucs), np.mean(mrrs), np.mean(ndcg5s), np.mean(ndcg10s)
