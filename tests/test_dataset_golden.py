"""The oracle's batch assembly (oracle.lime_oracle.assemble_*) against the tuples the IMPORTED reference datasets
produced (tests/golden/dataset_*.npz, tools/make_dataset_goldens.py): integer / byte / float work, so bit-exact."""
import os

import numpy as np
import pytest

import dataset_cases
from oracle import lime_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _same(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, (what, got.dtype, want.dtype, got.shape, want.shape)
    assert np.array_equal(got, want), what


def test_train_assembly_matches_reference_dataset():
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_train.npz'))
    out = O.assemble_train(corpus, g['train_samples'], g['train_freshness'], g['train_user_topic_lifetime'], dataset_cases.TRAIN_INDICES)
    assert len(out) == 25
    for k, arr in enumerate(out):
        _same(arr, g['out%02d' % k], 'train output %d' % k)
    # the fixture exercises both sides of dataset.py:125-128
    lens = [len(corpus.train_behaviors[i][9]) for i in dataset_cases.TRAIN_INDICES]
    assert min(lens) < cfg.max_history_num < max(lens)


@pytest.mark.parametrize('mode,rows', [('dev', dataset_cases.DEV_INDICES), ('test', dataset_cases.TEST_INDICES)])
def test_devtest_assembly_matches_reference_dataset(mode, rows):
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_%s.npz' % mode))
    out = O.assemble_devtest(corpus, mode, rows)
    assert len(out) == 25
    for k, arr in enumerate(out):
        _same(arr, g['out%02d' % k], '%s output %d' % (mode, k))


def test_negative_sampling_reproduces_the_reference_draws():
    """device_data.negative_sampling under the seed the goldens were drawn with (tools/make_dataset_goldens.py) gives the
    reference's sampled candidate tables, including numpy's exclusive upper bound in randint (dataset.py:6,64)."""
    from lime_cikm25_amd.device_data import negative_sampling
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_train.npz'))
    np.random.seed(dataset_cases.SAMPLING_SEED)
    samples, fresh, life = negative_sampling(corpus.train_behaviors, cfg.negative_sample_num)
    assert np.array_equal(np.asarray(samples, dtype=np.int64), g['train_samples'])
    assert np.array_equal(np.asarray(fresh, dtype=np.float64), g['train_freshness'])
    assert np.array_equal(np.asarray(life, dtype=np.float64), g['train_user_topic_lifetime'])
    many = [r for r in corpus.train_behaviors if len(r[4]) > cfg.negative_sample_num]
    assert many, 'the fixture must exercise the random branch'
