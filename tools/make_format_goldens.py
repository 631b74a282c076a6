"""Generate tests/golden/formats.json by running the IMPORTED REFERENCE's ``Corpus`` (corpus.py) on a small synthetic dataset
directory (build container only): the behaviour records it parses out of behaviors.tsv / news.tsv are the pin for
lime_cikm25_amd/formats.py.

    python tools/make_format_goldens.py

The dataset is written to a temp dir in the reference's own file layout ('adressa' flavour: no knowledge-graph files; the
'MIND' regex tokenizer: no nltk); ``torchtext.vocab.GloVe`` is replaced by an in-memory table (the word vectors do not
matter here).  Stored: the tsv lines (synthetic data), the dictionaries Corpus built, and its train / dev / test records.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness  # noqa: E402

TOPICS = ['sports', 'news', 'finance', 'travel']
WORDS = ['alpha', 'beta', 'gamma', 'delta', 'epsilon', 'zeta', 'eta', 'theta', 'iota', 'kappa', '42', '3.5', 'Rare1', 'rare2', ',', '!',
         'MiXed', 'x_y']


def news_lines(ids, rng):
    out = []
    for nid in ids:
        cat = TOPICS[rng.integers(len(TOPICS))]
        title = ' '.join(WORDS[i] for i in rng.integers(len(WORDS), size=rng.integers(2, 7)))
        body = ' '.join(WORDS[i] for i in rng.integers(len(WORDS), size=rng.integers(3, 12)))
        out.append('\t'.join([nid, cat, cat + '-sub%d' % rng.integers(2), title, body, '2020-01-01 00:00:00', '[]', '[]']) + '\n')
    return out


def behavior_lines(n, users, pool, rng, start_id):
    out = []
    for i in range(n):
        hist = list(rng.choice(pool, size=rng.integers(0, 7), replace=False))
        k = len(hist)
        fresh = [float(x) for x in np.round(rng.uniform(60, 2e6, size=k), 3)]
        life = [float(x) for x in np.round(rng.uniform(600, 1e6, size=k), 3)]
        cands = list(rng.choice(pool, size=rng.integers(2, 6), replace=False))
        labels = ['0'] * len(cands)
        for j in rng.choice(len(cands), size=min(len(cands) - 1, rng.integers(1, 3)), replace=False):      # both classes present
            labels[j] = '1'
        seen = {t: float(np.round(rng.uniform(1e3, 1e5), 2)) for t in rng.choice(TOPICS, size=rng.integers(0, 3), replace=False)}
        unseen = {t: float(np.round(rng.uniform(1e3, 1e5), 2)) for t in rng.choice(TOPICS, size=rng.integers(0, 3), replace=False)}
        out.append('\t'.join([str(start_id + i), users[rng.integers(len(users))],
                              repr([fresh, life, [float(np.round(rng.uniform(60, 1e5), 3))]]), ' '.join(hist),
                              ' '.join('%s-%s' % (c, l) for c, l in zip(cands, labels)),
                              json.dumps([seen, unseen, float(np.round(rng.uniform(1e3, 1e5), 2))])]) + '\n')
    return out


def main():
    rng = np.random.default_rng(7)
    ref_harness._install_stubs()

    class GloVe:                                                      # stand-in for torchtext.vocab.GloVe
        def __init__(self, name=None, dim=50, cache=None, max_vectors=None):
            self.stoi = {w: i for i, w in enumerate(WORDS[:6])}
            self.vectors = torch.zeros(6, dim)
    sys.modules['torchtext.vocab'].GloVe = GloVe
    sys.path.insert(0, ref_harness.REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    work = tempfile.mkdtemp(prefix='lime_fmt_')
    roots = {s: os.path.join(work, s) for s in ('train', 'dev', 'test')}
    train_ids = ['N%d' % i for i in range(1, 13)]
    dev_ids = train_ids[:8] + ['N%d' % i for i in range(13, 17)]
    test_ids = train_ids[4:] + ['N%d' % i for i in range(17, 20)]
    files = {}
    for split, ids in (('train', train_ids), ('dev', dev_ids), ('test', test_ids)):
        os.makedirs(roots[split])
        files[split + '_news'] = news_lines(ids, rng) if split == 'train' else None
    # a news keeps ONE line across the splits (the reference takes the first occurrence)
    all_news = {l.split('\t')[0]: l for l in files['train_news']}
    for split, ids in (('dev', dev_ids), ('test', test_ids)):
        fresh = [i for i in ids if i not in all_news]
        for l in news_lines(fresh, rng):
            all_news[l.split('\t')[0]] = l
        files[split + '_news'] = [all_news[i] for i in ids]
    users = ['U%d' % i for i in range(1, 6)]
    files['train_behaviors'] = behavior_lines(7, users, train_ids, rng, 1)
    files['dev_behaviors'] = behavior_lines(5, users + ['U99'], dev_ids, rng, 100)
    files['test_behaviors'] = behavior_lines(4, users + ['U98'], test_ids, rng, 200)
    for split in ('train', 'dev', 'test'):
        open(os.path.join(roots[split], 'news.tsv'), 'w', encoding='utf-8').writelines(files[split + '_news'])
        open(os.path.join(roots[split], 'behaviors.tsv'), 'w', encoding='utf-8').writelines(files[split + '_behaviors'])
    cfg = types.SimpleNamespace(
        dataset='adressa', tokenizer='MIND', word_threshold=12, word_embedding_dim=50, max_title_length=6, max_abstract_length=10,
        max_history_num=4, negative_sample_num=2, user_encoder='CROWN', no_self_connection=False, no_adjacent_normalization=False,
        gcn_normalization_type='symmetric', train_root=roots['train'], dev_root=roots['dev'], test_root=roots['test'],
        entity_embedding_dim=100, context_embedding_dim=100)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        from corpus import Corpus
        c = Corpus(cfg)
    finally:
        os.chdir(cwd)

    def rec(r):
        return [x.tolist() if isinstance(x, np.ndarray) else x for x in r]
    out = {
        'max_history_num': cfg.max_history_num,
        'lines': files,
        'news_ID_dict': c.news_ID_dict, 'user_ID_dict': c.user_ID_dict, 'category_dict': c.category_dict,
        'news_category': c.news_category.tolist(), 'news_subCategory': c.news_subCategory.tolist(),
        'subCategory_dict': c.subCategory_dict, 'word_dict': c.word_dict,
        'max_title_length': cfg.max_title_length, 'max_abstract_length': cfg.max_abstract_length,
        'news_title_text': c.news_title_text.tolist(), 'news_title_mask': c.news_title_mask.astype(int).tolist(),
        'news_abstract_text': c.news_abstract_text.tolist(), 'news_abstract_mask': c.news_abstract_mask.astype(int).tolist(),
        'category_index_to_name': {str(k): v for k, v in c.category_index_to_name.items()},
        'train_behaviors': [rec(r) for r in c.train_behaviors],
        'dev_behaviors': [rec(r) for r in c.dev_behaviors], 'dev_indices': list(c.dev_indices),
        'test_behaviors': [rec(r) for r in c.test_behaviors], 'test_indices': list(c.test_indices),
    }
    path = os.path.join(ROOT, 'tests', 'golden', 'formats.json')
    json.dump(out, open(path, 'w'), indent=0)
    print('%s: %d train / %d dev / %d test records, %.1f KB' % (path, len(out['train_behaviors']), len(out['dev_behaviors']),
                                                              len(out['test_behaviors']), os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
