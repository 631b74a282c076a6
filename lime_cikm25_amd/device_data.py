"""Device-side batch assembly: the reference's Train_Dataset / DevTest_Dataset
(dataset.py:105-141, :192-227) are pure gathers from the corpus tables (corpus.py:360-367) by news index, so the tables
stay resident in HBM and a batch is two launches of ``lime_gather_rows_multi`` (behaviour rows by behaviour index, then the
eight per-news arrays of every candidate and history slot by news index) instead of B x (H + K) host-side numpy gathers,
a collate and a host->device copy per step.  Negative sampling (dataset.py:42-77) stays on the host, as in the reference:
its result (the sampled candidate tables) is an input here.

The output is the reference's 25-tuple (+ remaining_lifetime, which the caller derives: trainer.py:126-127), with each
candidate tensor laid out directly in front of its history counterpart so that the news encoder's cat is a view.
"""
import numpy as np
import torch

from . import ops

_NEWS_FIELDS = ('news_category', 'news_subCategory', 'news_title_text', 'news_title_mask', 'news_title_entity',
                'news_abstract_text', 'news_abstract_mask', 'news_abstract_entity')


def _pad_history_list(values, H):
    """dataset.py:125-128: the LAST H entries, then zeros up to H."""
    values = list(values)
    return values[-H:] + [0] * max(0, H - len(values))


def negative_sampling(train_behaviors, negative_sample_num, randint=None):
    """Train_Dataset.negative_sampling (dataset.py:42-77) on the host: per train record [positive, negatives ...] news indices,
    the candidate freshness repeated, and [positive lifetime, negative lifetimes ...].  A record with at most
    ``negative_sample_num`` non-clicked news cycles through them; otherwise distinct ones are drawn with ``randint(0, n - 1)``.
    The reference imports ``randint`` from numpy.random (dataset.py:6), whose upper bound is EXCLUSIVE: the last non-clicked
    news of such an impression is never drawn.  That is the default here too (same ``np.random.seed`` -> same samples, pinned
    by tests/golden/dataset_train.npz); pass ``randint=lambda lo, hi: random.randint(lo, hi)`` for the inclusive draw."""
    if randint is None:
        from numpy.random import randint
    samples, freshness, lifetime = [], [], []
    for rec in train_behaviors:
        neg_indices, fresh, neg_lifetimes = rec[4], rec[6], rec[8]
        s, f, l = [rec[3]], [fresh], [rec[7]]
        n = len(neg_indices)
        used = set()
        for j in range(negative_sample_num):
            if n <= negative_sample_num:
                k = j % n
            else:
                while True:
                    k = randint(0, n - 1)
                    if k not in used:
                        used.add(k)
                        break
            s.append(neg_indices[k])
            f.append(fresh)
            l.append(neg_lifetimes[k])
        samples.append(s)
        freshness.append(f)
        lifetime.append(l)
    return samples, freshness, lifetime


class DeviceCorpus:
    """The eight per-news arrays of the reference's Corpus (corpus.py:360-367) on the device."""

    def __init__(self, corpus, device='cuda'):
        self.device = torch.device(device)
        for name in _NEWS_FIELDS:
            arr = np.ascontiguousarray(getattr(corpus, name))
            setattr(self, name, torch.from_numpy(arr).to(self.device))
        self.max_history_num = corpus.max_history_num
        self.category_num = corpus.config.category_num

    def fields(self):
        return [getattr(self, n) for n in _NEWS_FIELDS]


class DeviceBehaviors:
    """One split's behaviour table on the device: per behaviour row the user id, the history (news indices, mask, padded
    freshness / lifetime lists) and the candidates (news indices, freshness, lifetime)."""

    def __init__(self, corpus, user_id, hist_index, hist_mask, user_fr, user_lt, cand_index, cand_fr, cand_lt, eval_shape):
        dev = corpus.device
        self.corpus = corpus
        self.eval_shape = eval_shape
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dt))).to(dev)
        self.user_id = t(user_id, np.int64).view(-1, 1)
        self.hist_index = t(hist_index, np.int32)
        self.hist_mask = t(hist_mask, bool)
        self.user_freshness = t(user_fr, np.float32)
        self.user_lifetime = t(user_lt, np.float32)
        self.cand_index = t(cand_index, np.int32)
        self.cand_freshness = t(cand_fr, np.float32)
        self.cand_lifetime = t(cand_lt, np.float32)
        self.num = self.user_id.shape[0]
        self._plans, self._turn = {}, {}

    @classmethod
    def from_train(cls, dev_corpus, corpus, train_samples, train_freshness, train_user_topic_lifetime):
        """corpus.train_behaviors (corpus.py:539-552) + the tables negative_sampling filled (dataset.py:42-77)."""
        H = corpus.max_history_num
        beh = corpus.train_behaviors
        return cls(dev_corpus, [b[0] for b in beh], np.stack([np.asarray(b[1]) for b in beh]), np.stack([np.asarray(b[2]) for b in beh]),
                   [_pad_history_list(b[9], H) for b in beh], [_pad_history_list(b[10], H) for b in beh],
                   train_samples, train_freshness, train_user_topic_lifetime, eval_shape=False)

    @classmethod
    def from_devtest(cls, dev_corpus, corpus, mode):
        """corpus.dev_behaviors / test_behaviors (corpus.py:590-600): one candidate per row."""
        assert mode in ('dev', 'test')
        H = corpus.max_history_num
        beh = corpus.dev_behaviors if mode == 'dev' else corpus.test_behaviors
        return cls(dev_corpus, [b[0] for b in beh], np.stack([np.asarray(b[1]) for b in beh]), np.stack([np.asarray(b[2]) for b in beh]),
                   [_pad_history_list(b[7], H) for b in beh], [_pad_history_list(b[8], H) for b in beh],
                   np.asarray([b[3] for b in beh]).reshape(-1, 1), np.asarray([b[5] for b in beh]).reshape(-1, 1),
                   np.asarray([b[6] for b in beh]).reshape(-1, 1), eval_shape=True)

    def assemble(self, rows):
        """The collated batch of behaviour rows `rows` (int32 / int64 tensor or sequence): the reference's 25 tensors in
        ``__getitem__`` order, on the device.  Train split: candidates [B, 1 + neg, ...]; dev / test: without the N axis.

        The output tensors come from a ring of two pre-built workspaces per batch size (allocations, views and the two
        descriptor tables are made once): a returned batch stays valid until the call after the next one.
        """
        dev = self.corpus.device
        rows = torch.as_tensor(rows, device=dev)
        B = rows.numel()
        if B not in self._plans:
            self._plans[B] = [self._plan(B), self._plan(B)]
        ring = self._plans[B]
        plan = ring[self._turn.get(B, 0)]
        self._turn[B] = 1 - self._turn.get(B, 0)
        plan['rows'].copy_(rows.reshape(-1))                                 # int64 -> int32 on the way if need be
        ops.gather_rows_multi_run(plan['rows'], plan['level1'])
        ops.gather_rows_multi_run(plan['idx_all'], plan['level2'])
        return list(plan['out'])

    def _plan(self, B):
        c = self.corpus
        dev = c.device
        H, N = self.hist_index.shape[1], self.cand_index.shape[1]
        rows = torch.empty(B, dtype=torch.int32, device=dev)
        # level 1: behaviour rows.  The news indices go into ONE vector, candidates first: the order of the level-2 outputs.
        idx_all = torch.empty(B * N + B * H, dtype=torch.int32, device=dev)
        user_id = torch.empty((B, 1), dtype=torch.int64, device=dev)
        hist_mask = torch.empty((B, H), dtype=torch.bool, device=dev)
        ufr, ult = torch.empty((B, H), dtype=torch.float32, device=dev), torch.empty((B, H), dtype=torch.float32, device=dev)
        cfr, clt = torch.empty((B, N), dtype=torch.float32, device=dev), torch.empty((B, N), dtype=torch.float32, device=dev)
        level1 = [(self.cand_index, idx_all[:B * N].view(B, N)), (self.hist_index, idx_all[B * N:].view(B, H)),
                  (self.user_id, user_id), (self.hist_mask, hist_mask), (self.user_freshness, ufr), (self.user_lifetime, ult),
                  (self.cand_freshness, cfr), (self.cand_lifetime, clt)]
        # level 2: the eight per-news arrays for candidates and history in one launch; every output is one buffer whose
        # first B * N rows are the candidates and the rest the history (adjacent -> the encoder's cat is a view)
        outs = [torch.empty((B * N + B * H,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in c.fields()]
        news = [o[:B * N].view((B, N) + tuple(o.shape[1:])) for o in outs]
        user = [o[B * N:].view((B, H) + tuple(o.shape[1:])) for o in outs]
        if self.eval_shape:                                                  # DevTest_Dataset: candidate tensors without the N axis
            news = [t.squeeze(1) for t in news]
            cfr, clt = cfr.squeeze(1), clt.squeeze(1)
        out = [user_id.view(B)] + user + [ufr, ult, hist_mask] + self._zeros(B, H) + news + [cfr, clt]
        return {'rows': rows, 'idx_all': idx_all, 'level1': ops.gather_rows_multi_prepare(B, level1),
                'level2': ops.gather_rows_multi_prepare(B * N + B * H, list(zip(c.fields(), outs))), 'out': out}

    def _zeros(self, B, H):
        """dataset.py:119-121: the SUE-only tensors are zeros for every other user encoder."""
        dev = self.corpus.device
        return [torch.zeros((B, H, H), dtype=torch.float32, device=dev),
                torch.zeros((B, self.corpus.category_num + 1), dtype=torch.bool, device=dev),
                torch.zeros((B, H), dtype=torch.int64, device=dev)]
