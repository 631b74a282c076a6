"""Counter-based synthetic weights and MIND-shaped impression batches.

Everything here is a pure function of (name, seed, index) through splitmix64, so
the GPU box, this container and the golden-vector generator (tools/make_goldens.py)
produce bit-identical weights and inputs without sharing any RNG state, torch/numpy
RNG version, or module construction order.

Input layout follows the 26-tensor ``Model.forward`` signature of the reference
(model.py:151-154) with dtypes from dataset.py:118-141 / corpus.py:361-368.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(s):
    h = 0xCBF29CE484222325
    for b in s.encode('utf-8'):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z):
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over='ignore'):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream_u64(name, seed, n):
    """n 64-bit words of the stream identified by (name, seed)."""
    base = np.uint64(_fnv1a64(name) ^ ((int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over='ignore'):
        ctr = base + (np.arange(1, n + 1, dtype=np.uint64) * _GOLD)
    return _mix(ctr)


def uniform01(name, seed, n):
    """float64 uniforms in [0, 1)."""
    return (stream_u64(name, seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def randint(name, seed, n, lo, hi):
    """int64 uniform in [lo, hi)."""
    return lo + np.floor(uniform01(name, seed, n) * (hi - lo)).astype(np.int64)


# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------

def canonical_name(key):
    """state_dict lists the shared news encoder twice; generate it once."""
    if key.startswith('user_encoder.news_encoder.'):
        return key[len('user_encoder.'):]
    return key


def synth_tensor(key, shape, seed=0):
    """Non-degenerate deterministic value for the state_dict entry ``key``.

    Rules are by name/shape only: LayerNorm gains in (0.5, 1.5); biases in (-0.1, 0.1);
    2-D ``weight`` of a Linear [out, in] uniform with unit gain (a = sqrt(3 / in));
    embedding tables std ~ 0.87; the word table std ~ 0.35 with row 0 zero
    (corpus.py:193-196 starts the pad row at zero).
    """
    name = canonical_name(key)
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, seed, n)
    leaf = name.split('.')[-1]
    parent = name.split('.')[-2] if '.' in name else ''
    is_ln = parent in ('norm1', 'norm2', 'layernorm', 'ln0', 'ln1')
    if is_ln and leaf == 'weight':
        v = 0.5 + u
    elif leaf == 'bias' or leaf == 'in_proj_bias':
        v = (u - 0.5) * 0.2
    elif parent == 'word_embedding':
        v = (u - 0.5) * 1.2
        v = v.reshape(shape)
        v[0, :] = 0.0
    elif parent.endswith('_embedding') or parent.endswith('Category_embedding'):
        v = (u - 0.5) * 3.0
    elif leaf == 'user_node_embedding':
        v = (u - 0.5) * 1.0
    elif len(shape) == 2:
        a = math.sqrt(3.0 / shape[1])
        v = (u - 0.5) * 2.0 * a
    else:
        v = (u - 0.5) * 0.2
    return torch.from_numpy(np.asarray(v, dtype=np.float64).reshape(shape).astype(np.float32))


def fill_state_dict(module, seed=0, skip_suffixes=('.pe',)):
    """Overwrite every parameter/buffer of ``module`` (except positional tables) by name."""
    sd = module.state_dict()
    new = OrderedDict()
    for k, v in sd.items():
        if any(k.endswith(s) for s in skip_suffixes) or not v.dtype.is_floating_point:
            new[k] = v
        else:
            new[k] = synth_tensor(k, v.shape, seed).to(v.dtype)
    module.load_state_dict(new)
    return module


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------

INPUT_NAMES = (
    'user_ID', 'user_category', 'user_subCategory', 'user_title_text', 'user_title_mask', 'user_title_entity',
    'user_content_text', 'user_content_mask', 'user_content_entity', 'user_freshness', 'user_user_topic_lifetime',
    'user_history_mask', 'user_history_graph', 'user_history_category_mask', 'user_history_category_indices',
    'news_category', 'news_subCategory', 'news_title_text', 'news_title_mask', 'news_title_entity',
    'news_content_text', 'news_content_mask', 'news_content_entity', 'news_freshness', 'news_user_topic_lifetime',
    'remaining_lifetime',
)


def _texts(tag, seed, rows, length, vocab, min_len):
    """Token ids in [1, vocab) for the first len positions, 0 after; len ~ U{min_len..length}."""
    lens = randint(tag + '.len', seed, rows, min(min_len, length), length + 1)
    ids = randint(tag + '.ids', seed, rows * length, 1, vocab).reshape(rows, length)
    pos = np.arange(length)[None, :]
    mask = pos < lens[:, None]
    ids = np.where(mask, ids, 0)
    return ids.astype(np.int32), mask


def _log_uniform(tag, seed, n, lo, hi):
    u = uniform01(tag, seed, n)
    return np.exp(math.log(lo) + u * (math.log(hi) - math.log(lo))).astype(np.float32)


def make_batch(config, B, N, seed=0, eval_shape=False, expired_fraction=0.5):
    """One synthetic impression batch as an OrderedDict of the 26 ``Model.forward`` inputs.

    ``eval_shape=True`` drops the candidate axis (N must be 1) as DevTest_Dataset does
    (dataset.py:192-227); ``Model.forward`` unsqueezes it again in eval mode (model.py:158-169).
    History rows beyond a row's history length are padding: news id 0 = all-pad tokens with
    mask[0] = 1 (corpus.py:476-477), freshness/lifetime 0 (dataset.py:125,128).
    Candidates of a row share one freshness (dataset.py:53,73).  ``remaining_lifetime`` is the
    caller-side ``news_user_topic_lifetime - news_freshness`` (trainer.py:126-127).
    """
    H, T, L = config.max_history_num, config.max_title_length, config.max_abstract_length
    V, C, SC = config.vocabulary_size, config.category_num, config.subCategory_num
    s = int(seed)
    d = OrderedDict()
    d['user_ID'] = torch.from_numpy(randint('user_ID', s, B, 0, config.user_num))

    hist_len = randint('hist_len', s, B, 0, H + 1)
    hmask = np.arange(H)[None, :] < hist_len[:, None]                                   # [B, H]

    def hist(a, pad):
        shape = (B, H) + a.shape[1:]
        a = a.reshape(shape)
        m = hmask.reshape((B, H) + (1,) * (a.ndim - 2))
        return np.where(m, a, pad)

    ucat = hist(randint('user_category', s, B * H, 0, C), 0).astype(np.int32)
    usub = hist(randint('user_subCategory', s, B * H, 0, SC), 0).astype(np.int32)
    ut, utm = _texts('user_title', s, B * H, T, V, 4)
    uc, ucm = _texts('user_content', s, B * H, L, V, 8)
    ut = hist(ut, 0).astype(np.int32)
    uc = hist(uc, 0).astype(np.int32)
    utm = hist(utm, False)
    ucm = hist(ucm, False)
    # the pad news keeps mask[0] = 1 (corpus.py:476-477)
    utm[..., 0] = True
    ucm[..., 0] = True
    d['user_category'] = torch.from_numpy(ucat)
    d['user_subCategory'] = torch.from_numpy(usub)
    d['user_title_text'] = torch.from_numpy(ut)
    d['user_title_mask'] = torch.from_numpy(utm)
    d['user_title_entity'] = torch.zeros(B, H, T, dtype=torch.int32)
    d['user_content_text'] = torch.from_numpy(uc)
    d['user_content_mask'] = torch.from_numpy(ucm)
    d['user_content_entity'] = torch.zeros(B, H, L, dtype=torch.int32)
    d['user_freshness'] = torch.from_numpy(hist(_log_uniform('user_freshness', s, B * H, 60.0, 30 * 86400.0), 0.0).astype(np.float32))
    d['user_user_topic_lifetime'] = torch.from_numpy(
        hist(_log_uniform('user_user_topic_lifetime', s, B * H, 600.0, 14 * 86400.0), 0.0).astype(np.float32))
    d['user_history_mask'] = torch.from_numpy(hmask.copy())
    d['user_history_graph'] = torch.zeros(B, H, H, dtype=torch.float32)
    d['user_history_category_mask'] = torch.zeros(B, C + 1, dtype=torch.bool)
    d['user_history_category_indices'] = torch.zeros(B, H, dtype=torch.int64)

    d['news_category'] = torch.from_numpy(randint('news_category', s, B * N, 0, C).reshape(B, N).astype(np.int32))
    d['news_subCategory'] = torch.from_numpy(randint('news_subCategory', s, B * N, 0, SC).reshape(B, N).astype(np.int32))
    nt, ntm = _texts('news_title', s, B * N, T, V, 4)
    nc, ncm = _texts('news_content', s, B * N, L, V, 8)
    d['news_title_text'] = torch.from_numpy(nt.reshape(B, N, T))
    d['news_title_mask'] = torch.from_numpy(ntm.reshape(B, N, T))
    d['news_title_entity'] = torch.zeros(B, N, T, dtype=torch.int32)
    d['news_content_text'] = torch.from_numpy(nc.reshape(B, N, L))
    d['news_content_mask'] = torch.from_numpy(ncm.reshape(B, N, L))
    d['news_content_entity'] = torch.zeros(B, N, L, dtype=torch.int32)
    fresh = np.repeat(_log_uniform('news_freshness', s, B, 60.0, 30 * 86400.0)[:, None], N, axis=1)
    life = _log_uniform('news_user_topic_lifetime', s, B * N, 600.0, 14 * 86400.0).reshape(B, N)
    # a share of the candidates get a lifetime within a few seconds of their freshness so that the
    # sigmoid weight of util.py:42 is not saturated to exactly 0/1
    near = uniform01('near', s, B * N).reshape(B, N) < (1.0 - expired_fraction) * 0.5
    delta = ((uniform01('near.delta', s, B * N).reshape(B, N) - 0.5) * 16.0).astype(np.float32)
    life = np.where(near, fresh + delta, life).astype(np.float32)
    d['news_freshness'] = torch.from_numpy(fresh.astype(np.float32))
    d['news_user_topic_lifetime'] = torch.from_numpy(life)
    d['remaining_lifetime'] = d['news_user_topic_lifetime'] - d['news_freshness']
    if eval_shape:
        assert N == 1, 'the eval path carries one candidate per row (dataset.py:192-227)'
        for k in INPUT_NAMES[15:]:
            d[k] = d[k].squeeze(1)
    assert tuple(d.keys()) == INPUT_NAMES
    return d


def synth_word_embedding(config, seed=0):
    """The tensor the reference unpickles at newsEncoders.py:173-174."""
    return synth_tensor('news_encoder.base_news_encoder.word_embedding.weight',
                        (config.vocabulary_size, config.word_embedding_dim), seed)


def synth_corpus(config, n_news=200, n_train=40, n_dev=30, seed=0, n_neg_max=9):
    """A toy corpus object with the attributes the reference's datasets read (corpus.py:355-368 arrays, the behaviour lists of
    corpus.py:539-552 / 590-600), filled from the counter-based generator.  News 0 is the <PAD> news (corpus.py:476-477).
    Histories have 0 .. H clicked news; the per-behaviour freshness / lifetime lists are deliberately shorter or LONGER
    than H so that dataset.py:125-128's ``[-H:] + [0] * pad`` is exercised on both sides."""
    from types import SimpleNamespace
    H, T, L = config.max_history_num, config.max_title_length, config.max_abstract_length
    V = config.vocabulary_size
    c = SimpleNamespace()
    c.config = config
    c.negative_sample_num = config.negative_sample_num
    c.max_history_num, c.max_title_length, c.max_abstract_length = H, T, L
    c.news_category = randint('corpus.cat', seed, n_news, 0, config.category_num).astype(np.int32)
    c.news_subCategory = randint('corpus.sub', seed, n_news, 0, config.subCategory_num).astype(np.int32)

    def texts(tag, length):
        ids = randint('corpus.%s' % tag, seed, n_news * length, 1, V).reshape(n_news, length).astype(np.int32)
        lens = randint('corpus.%s.len' % tag, seed, n_news, 1, length + 1)
        mask = np.arange(length)[None, :] < lens[:, None]
        ent = randint('corpus.%s.ent' % tag, seed, n_news * length, 0, 50).reshape(n_news, length).astype(np.int32)
        return np.where(mask, ids, 0).astype(np.int32), mask, np.where(mask, ent, 0).astype(np.int32)
    c.news_title_text, c.news_title_mask, c.news_title_entity = texts('title', T)
    c.news_abstract_text, c.news_abstract_mask, c.news_abstract_entity = texts('abstract', L)
    for arr in (c.news_title_text, c.news_title_entity, c.news_abstract_text, c.news_abstract_entity):
        arr[0] = 0
    c.news_title_mask[0] = False
    c.news_abstract_mask[0] = False
    c.news_title_mask[0][0] = True                                         # corpus.py:476-477
    c.news_abstract_mask[0][0] = True
    c.news_category[0] = 0
    c.news_subCategory[0] = 0

    def history(tag, i):
        n = int(randint('%s.hn' % tag, seed + i, 1, 0, H + 1)[0])
        idx = np.zeros(H, dtype=np.int32)
        idx[:n] = randint('%s.h' % tag, seed + i, n, 1, n_news)
        mask = np.zeros(H, dtype=bool)
        mask[:n] = True
        n_list = int(randint('%s.ln' % tag, seed + i, 1, 0, H + 4)[0])      # list length independent of n: shorter and longer than H
        fr = (_log_uniform('%s.fr' % tag, seed + i, n_list, 60.0, 30 * 86400.0)).astype(np.float64).tolist()
        lt = (_log_uniform('%s.lt' % tag, seed + i, n_list, 600.0, 14 * 86400.0)).astype(np.float64).tolist()
        return idx, mask, fr, lt
    c.train_behaviors = []
    for i in range(n_train):
        idx, mask, fr, lt = history('train', i)
        n_neg = int(randint('train.nn', seed + i, 1, 1, n_neg_max + 1)[0])
        neg = randint('train.neg', seed + i, n_neg, 1, n_news).astype(np.int64).tolist()
        neg_lt = _log_uniform('train.neglt', seed + i, n_neg, 600.0, 14 * 86400.0).astype(np.float64).tolist()
        c.train_behaviors.append([int(randint('train.uid', seed + i, 1, 0, config.user_num)[0]), idx, mask,
                                  int(randint('train.pos', seed + i, 1, 1, n_news)[0]), neg, i,
                                  float(_log_uniform('train.cfr', seed + i, 1, 60.0, 30 * 86400.0)[0]),
                                  float(_log_uniform('train.plt', seed + i, 1, 600.0, 14 * 86400.0)[0]), neg_lt, fr, lt])
    for split, n in (('dev', n_dev), ('test', max(1, n_dev // 2))):
        beh = []
        for i in range(n):
            idx, mask, fr, lt = history(split, i)
            beh.append([int(randint('%s.uid' % split, seed + i, 1, 0, config.user_num)[0]), idx, mask,
                        int(randint('%s.cand' % split, seed + i, 1, 1, n_news)[0]), i,
                        float(_log_uniform('%s.cfr' % split, seed + i, 1, 60.0, 30 * 86400.0)[0]),
                        float(_log_uniform('%s.clt' % split, seed + i, 1, 600.0, 14 * 86400.0)[0]), fr, lt])
        setattr(c, '%s_behaviors' % split, beh)
    for split in ('train', 'dev', 'test'):                                  # SUE-only tables (dataset.py:21-28): absent for CROWN
        for name in ('user_history_graph', 'user_history_category_mask', 'user_history_category_indices'):
            setattr(c, '%s_%s' % (split, name), None)
    return c
