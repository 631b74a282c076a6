"""Drop-in for the scoring-path part of the reference's util.py: RemainingLifetimeWeighting (util.py:15-52) and the
eval harness around the model, compute_scores (util.py:77-129)."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .evaluate import scoring


class RemainingLifetimeWeighting(nn.Module):
    """Dot-product interest match x sigmoid remaining-lifetime weight (util.py:23-49) as one HIP kernel."""

    def __init__(self, config):
        super().__init__()
        self.alpha = config.sigmoid_scaling_alpha
        self.beta = config.penalty_scaling_beta
        self.use_expired_penalty = config.use_expired_penalty
        self.use_remaining_lifetime_weighting = config.use_remaining_lifetime_weighting

    def forward(self, user_embedding, news_embedding, remaining_lifetime):
        if torch.is_grad_enabled() and (user_embedding.requires_grad or news_embedding.requires_grad):
            from . import training
            return training.lifetime_weighted_logits(self, user_embedding, news_embedding, remaining_lifetime)
        return ops.lifetime_score(user_embedding, news_embedding, remaining_lifetime, self.alpha, self.beta,
                                  self.use_remaining_lifetime_weighting, self.use_expired_penalty)

    def initialize(self):
        pass


def rank_impressions(scores, indices):
    """util.py:113-123: per impression, the 1-based rank of every candidate under a stable descending sort (ties keep
    the candidate order; +0.0 and -0.0 tie).  ``scores``: one per row; ``indices``: the impression of each row
    (non-decreasing, 0-based; impressions without rows get an empty list).  Returns a list of rank lists."""
    scores = np.asarray(scores, dtype=np.float64)
    indices = np.asarray(indices, dtype=np.int64)
    n_imp = int(indices[-1]) + 1 if indices.size else 0
    pos = np.arange(scores.size)
    order = np.lexsort((pos, -scores, indices))             # by impression, then score descending, then candidate order
    starts = np.searchsorted(indices, np.arange(n_imp), side='left')
    counts = np.bincount(indices, minlength=n_imp)
    rank = np.empty(scores.size, dtype=np.int64)
    rank[order] = pos - np.repeat(starts, counts) + 1       # position inside the impression's sorted block
    return [rank[starts[i]:starts[i] + counts[i]].tolist() for i in range(n_imp)]


def write_rank_file(result_file, ranks):
    """util.py:117-123: ``<impression> [r1,r2,...]`` lines, 1-based impression ids, no spaces, no trailing newline."""
    with open(result_file, 'w', encoding='utf-8') as f:
        for i, r in enumerate(ranks):
            f.write(('' if i == 0 else '\n') + str(i + 1) + ' ' + str(r).replace(' ', ''))


def compute_scores(model, batches, indices, result_file, truth_file=None):
    """The reference's dev / test pass (util.py:77-129) over an iterable of 25-tensor batches (what DevTest_Dataset
    yields, dataset.py:216-227): score every (impression, candidate) row with the model in eval mode, write the rank
    file, and -- given the truth file -- return (AUC, MRR, nDCG@5, nDCG@10), else four Nones.  The remaining lifetime is
    derived per ``config.lifetime_type`` exactly as util.py:98-106."""
    config = model.config
    dev = getattr(model, 'device', None) or next(model.parameters()).device
    scores = []
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for batch in batches:
            batch = [x.to(dev, non_blocking=True) for x in batch]
            news_category, news_freshness, news_user_topic_lifetime = batch[15], batch[23], batch[24]
            if config.lifetime_type == 'fixed':
                remaining_lifetime = config.fixed_lifetime - news_freshness
            elif config.lifetime_type == 'topic_wise':
                remaining_lifetime = config.category_lifetime_map.to(news_category.device)[news_category.long()] - news_freshness
            elif config.lifetime_type == 'user_topic':
                remaining_lifetime = news_user_topic_lifetime - news_freshness
            else:
                raise ValueError('Invalid lifetime_type')
            scores.append(model(*batch, remaining_lifetime).squeeze(dim=1).float().cpu())
    model.train(was_training)
    scores = torch.cat(scores).tolist() if scores else []
    assert len(scores) == len(indices), 'one score per (impression, candidate) row'
    write_rank_file(result_file, rank_impressions(scores, indices))
    if truth_file is None:
        return None, None, None, None
    with open(truth_file, 'r', encoding='utf-8') as truth_f, open(result_file, 'r', encoding='utf-8') as result_f:
        return scoring(truth_f, result_f)


def compute_scores_cached(model, behaviors, indices, result_file, truth_file=None, rows_per_forward=None):
    """The same dev / test pass from a per-news content cache (Model.build_news_cache + Model.score_behaviors): every news goes
    through the token encoders ONCE instead of once per (row, slot) -- the reference re-encodes all 51 news of every row
    (util.py:86-111).  ``behaviors``: a dev / test ``DeviceBehaviors``.  Rows are scored in chunks of ``rows_per_forward`` (default:
    config.batch_size, as trainer.py:153 calls compute_scores; main.py:50,67 pass twice that, which only fits while
    2 x batch_size <= H + config.batch_size node slots) with the chunk's row count as the GraphSAGE source count, so the scores
    are those of ``compute_scores`` over the same batches; the remaining lifetime follows ``config.lifetime_type``
    ('fixed' / 'topic_wise' / 'user_topic', util.py:98-106)."""
    config = model.config
    if config.lifetime_type not in ('fixed', 'topic_wise', 'user_topic'):
        raise ValueError('Invalid lifetime_type')
    per = rows_per_forward or config.batch_size
    slots = behaviors.hist_index.shape[1] + model.user_encoder.user_node_embedding.shape[0]
    if per > slots:
        raise ValueError('rows_per_forward = %d exceeds the H + config.batch_size = %d GraphSAGE node slots (SURVEY Q7): the '
                         'reference raises an index error there' % (per, slots))
    was_training = model.training
    model.eval()
    cache = model.build_news_cache(behaviors.corpus)
    scores = []
    for r0 in range(0, behaviors.num, per):
        rows = list(range(r0, min(behaviors.num, r0 + per)))
        scores.append(model.score_behaviors(behaviors, rows, cache, n_src=len(rows)).float().cpu())
    model.train(was_training)
    scores = torch.cat(scores).tolist() if scores else []
    assert len(scores) == len(indices), 'one score per (impression, candidate) row'
    write_rank_file(result_file, rank_impressions(scores, indices))
    if truth_file is None:
        return None, None, None, None
    with open(truth_file, 'r', encoding='utf-8') as truth_f, open(result_file, 'r', encoding='utf-8') as result_f:
        return scoring(truth_f, result_f)
