"""The eval harness against goldens produced by the reference's own util.compute_scores and
evaluate.scoring (tools/make_eval_goldens.py).  CPU only."""
import io
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR
from lime_cikm25_amd import evaluate as E
from lime_cikm25_amd import util as U


def golden(name):
    with open(os.path.join(GOLDEN_DIR, name + '.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('name', ['eval_plain', 'eval_ties'])
def test_rank_file_is_byte_identical(name, tmp_path):
    g = golden(name)
    ranks = U.rank_impressions(g['scores'], g['indices'])
    path = tmp_path / 'ranks.txt'
    U.write_rank_file(str(path), ranks)
    assert path.read_text() == g['rank_file']


@pytest.mark.parametrize('name', ['eval_plain', 'eval_ties'])
def test_metrics_match_reference_scoring(name):
    g = golden(name)
    got = E.scoring(io.StringIO(g['truth_file']), io.StringIO(g['rank_file']))
    assert np.allclose(got, g['metrics'], rtol=0, atol=1e-12)


def test_ties_keep_candidate_order_and_signed_zeros_tie():
    scores = [0.0, -0.0, 1.0, 0.0, 1.0, -3.0]
    assert U.rank_impressions(scores, [0] * 6) == [[3, 4, 1, 5, 2, 6]]
    assert U.rank_impressions([2.0, 1.0, 5.0], [0, 2, 2]) == [[1], [], [2, 1]]       # impression 1 has no rows


def test_auc_equals_sklearn():
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(2, 30))
        y = rng.integers(0, 2, n)
        if y.min() == y.max():
            y[0] = 1 - y[0]
        s = np.round(rng.normal(size=n), int(rng.integers(0, 3)))      # plenty of ties
        assert abs(E.roc_auc_score(y, s) - roc_auc_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        E.roc_auc_score([1, 1], [0.1, 0.2])


def test_compute_scores_end_to_end_with_a_stub_model(tmp_path):
    g = golden('eval_ties')
    scores = torch.tensor(g['scores'], dtype=torch.float32)

    class Stub:
        config = type('C', (), dict(lifetime_type='user_topic', fixed_lifetime=0))
        device = torch.device('cpu')
        training = True

        def __init__(self):
            self.pos = 0

        def eval(self):
            self.training = False

        def train(self, mode=True):
            self.training = mode

        def __call__(self, *args):
            assert len(args) == 26 and torch.equal(args[25], args[24] - args[23])      # remaining lifetime, util.py:104
            n = args[0].shape[0]
            out = scores[self.pos:self.pos + n].unsqueeze(1)
            self.pos += n
            return out

    rows = len(g['scores'])
    batches = [[torch.arange(lo, min(lo + 50, rows), dtype=torch.float32)] * 25 for lo in range(0, rows, 50)]
    truth = tmp_path / 'truth.txt'
    truth.write_text(g['truth_file'])
    model = Stub()
    got = U.compute_scores(model, batches, g['indices'], str(tmp_path / 'r.txt'), str(truth))
    assert (tmp_path / 'r.txt').read_text() == g['rank_file']
    assert np.allclose(got, g['metrics'], rtol=0, atol=1e-12)
    assert model.training is True                                                      # mode restored
