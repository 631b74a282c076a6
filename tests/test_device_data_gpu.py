"""Device-side batch assembly (lime_gather_rows_multi behind DeviceCorpus / DeviceBehaviors) against the tuples of the
imported reference datasets (tests/golden/dataset_*.npz) and the oracle restatement: integer / byte / float gathers,
bit-exact.  Then the assembled batch drives the model."""
import os

import numpy as np
import pytest
import torch

import dataset_cases
from oracle import lime_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _check(batch, golden_arrays, what):
    assert len(batch) == len(golden_arrays) == 25
    for k, (t, want) in enumerate(zip(batch, golden_arrays)):
        got = t.cpu().numpy()
        assert got.dtype == want.dtype and got.shape == want.shape, (what, k, got.dtype, want.dtype, got.shape, want.shape)
        assert np.array_equal(got, want), '%s output %d' % (what, k)


def test_train_batches_match_reference_dataset():
    from lime_cikm25_amd import DeviceBehaviors, DeviceCorpus
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_train.npz'))
    dc = DeviceCorpus(corpus)
    beh = DeviceBehaviors.from_train(dc, corpus, g['train_samples'], g['train_freshness'], g['train_user_topic_lifetime'])
    _check(beh.assemble(dataset_cases.TRAIN_INDICES), [g['out%02d' % k] for k in range(25)], 'train vs reference')
    rows = list(range(beh.num))[::-1]                                       # every behaviour, reversed
    want = O.assemble_train(corpus, g['train_samples'], g['train_freshness'], g['train_user_topic_lifetime'], rows)
    _check(beh.assemble(torch.tensor(rows)), want, 'train vs oracle')


@pytest.mark.parametrize('mode,rows', [('dev', dataset_cases.DEV_INDICES), ('test', dataset_cases.TEST_INDICES)])
def test_devtest_batches_match_reference_dataset(mode, rows):
    from lime_cikm25_amd import DeviceBehaviors, DeviceCorpus
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_%s.npz' % mode))
    beh = DeviceBehaviors.from_devtest(DeviceCorpus(corpus), corpus, mode)
    _check(beh.assemble(rows), [g['out%02d' % k] for k in range(25)], '%s vs reference' % mode)
    every = list(range(beh.num))
    _check(beh.assemble(every), O.assemble_devtest(corpus, mode, every), '%s vs oracle' % mode)


def test_assembled_batch_drives_the_model():
    """Assembled on the device -> Model.forward (training shape and eval shape) == the oracle on the same tuple."""
    from lime_cikm25_amd import DeviceBehaviors, DeviceCorpus, Model, synth
    from helpers import rel_err
    cfg, corpus = dataset_cases.build()
    g = np.load(os.path.join(GOLD, 'dataset_train.npz'))
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, 5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    dc = DeviceCorpus(corpus)
    names = list(synth.make_batch(cfg, 2, 2, seed=0).keys())               # the 26 forward arguments, in order
    for beh, eval_shape in ((DeviceBehaviors.from_train(dc, corpus, g['train_samples'], g['train_freshness'],
                                                        g['train_user_topic_lifetime']), False),
                            (DeviceBehaviors.from_devtest(dc, corpus, 'dev'), True)):
        batch = beh.assemble(list(range(8)))
        remaining = batch[24] - batch[23]                                   # trainer.py:126-127 (lifetime_type user_topic)
        model.training = not eval_shape
        with torch.no_grad():
            got = model(*batch, remaining).cpu()
        cpu = {n: t.cpu() for n, t in zip(names, batch + [remaining])}
        want = O.model_forward(sd, cfg, cpu, eval_shape=eval_shape)
        assert rel_err(got.numpy().reshape(-1), want.numpy().reshape(-1)) < 1e-3


def test_cached_scoring_equals_the_eval_forward():
    """Model.score_behaviors (per-news content cache, no token encoder at scoring time) == Model.forward in eval mode on the
    assembled rows == the oracle."""
    from lime_cikm25_amd import DeviceBehaviors, DeviceCorpus, Model, synth
    from helpers import rel_err
    cfg, corpus = dataset_cases.build()
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, 9)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    dc = DeviceCorpus(corpus)
    beh = DeviceBehaviors.from_devtest(dc, corpus, 'dev')
    cache = model.build_news_cache(dc, rows_per_pass=25)                   # several passes over the 60 news
    assert cache.shape == (60, cfg.lime_output_dim)
    rows = list(range(beh.num))
    got = model.score_behaviors(beh, rows, cache).cpu()
    batch = beh.assemble(rows)
    remaining = batch[24] - batch[23]
    model.use_graph = False
    with torch.no_grad():
        ref = model(*batch, remaining).cpu().reshape(-1)
    assert rel_err(got.numpy(), ref.numpy()) < 2e-5
    names = list(synth.make_batch(cfg, 2, 2, seed=0).keys())
    want = O.model_forward(sd, cfg, {n: t.cpu() for n, t in zip(names, batch + [remaining])}, eval_shape=True).reshape(-1)
    assert rel_err(got.numpy(), want.numpy()) < 1e-3
