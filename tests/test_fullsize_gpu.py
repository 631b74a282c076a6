"""Parity at the FULL sizes of BASELINE.json configs[2], [3] (per-GPU shape) and [4] on the MI355X, against the CPU oracle:

  configs[2]  batch 256, history 50, title 32 + body 128, K = 1+4, config.batch_size 256 -- fp32 AND the bf16 token encoders.
              n_src = 256 rows > H = 50: the GraphSAGE mean spills 206 slots deep into user_node_embedding at real size.
  configs[3]  the per-GPU shape of the 8-GPU data-parallel run: batch 32, body 512, config.batch_size 256 -- scoring forward.
  configs[4]  1024 impressions x K = 100, scoring only: a 64-impression slice against the oracle (every news encoded once,
              tests/oracle_impressions.py, tied to the oracle's expanded-row forward by a CPU test), and the full 1024 x 100
              through size-independent properties: bitwise determinism, impression-permutation equivariance, agreement of the
              slice with the same impressions inside the full run.

fp32 tolerance 1e-3 relative (north star; observed ~1e-6..1e-5 is printed).  bf16: max |dlogit| <= 1e-2 x mean |logit| and
|dAUC| <= 0.01 (bf16 keeps 8 mantissa bits; this is what the bf16 arithmetic delivers, observed values printed)."""
import numpy as np
import pytest
import torch

import oracle_impressions
from helpers import rel_err
from lime_cikm25_amd import Model, make_config, synth
from oracle import lime_oracle as O
from test_model_gpu import gpu_model, run

pytestmark = pytest.mark.gpu
TOL = 1e-3
BF16_TOL = 6e-3     # observed 4e-3; tests/test_bf16_budget.py attributes it: the bf16 WEIGHTS (a rounding shared by every token, so the
                    # token mean does not average it out) are 3.3e-3 of an emulated 3.5e-3, every activation rounding together 1.0e-3


def _auc(scores):
    from sklearn.metrics import roc_auc_score
    labels = np.zeros_like(scores)
    labels[:, 0] = 1
    return float(np.mean([roc_auc_score(labels[r], scores[r]) for r in range(scores.shape[0])]))


@pytest.fixture(scope='module')
def cfg3():
    cfg = make_config(vocabulary_size=50000, batch_size=256)
    model, sd = gpu_model(cfg, seed=61)
    batch = synth.make_batch(cfg, 256, 5, seed=62)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    want = O.model_forward(sd, cfg, batch)                    # ~25 s on 16 cores
    return cfg, model, sd, batch, want


def test_config3_shape_fp32_against_oracle(cfg3):
    cfg, model, sd, batch, want = cfg3
    got = run(model, batch, False)
    e = rel_err(got.numpy(), want.numpy())
    print('configs[2] shape fp32 (B=256, n_src=256 > H=50): max rel err vs oracle %.2e' % e)
    assert got.shape == (256, 5) and e < TOL
    assert torch.equal(run(model, batch, False), got)
    assert abs(_auc(got.numpy()) - _auc(want.numpy())) <= 1e-3
    # the spill is live at this size: with user_node_embedding zeroed the oracle's logits move by far more than the tolerance
    sd0 = dict(sd)
    sd0['user_encoder.user_node_embedding'] = torch.zeros_like(sd['user_encoder.user_node_embedding'])
    assert rel_err(O.model_forward(sd0, cfg, batch).numpy(), want.numpy()) > 10 * TOL


def test_config3_bf16_against_oracle(cfg3):
    cfg, model32, sd, batch, want = cfg3
    model = Model(make_config(vocabulary_size=50000, batch_size=256, compute_dtype='bf16'))
    model.load_state_dict(sd)
    model = model.cuda()
    got = run(model, batch, False)
    scale = float(want.abs().mean())
    err = float((got - want).abs().max()) / scale
    dauc = abs(_auc(got.numpy()) - _auc(want.numpy()))
    print('configs[2] bf16 (B=256): max |dlogit| / mean |logit| = %.3e, |dAUC| = %.4f' % (err, dauc))
    assert torch.isfinite(got).all() and err < BF16_TOL and dauc <= 0.01
    assert torch.equal(run(model, batch, False), got)


def test_config4_per_gpu_shape_forward_against_oracle():
    cfg = make_config(vocabulary_size=50000, max_abstract_length=512, batch_size=256)
    model, sd = gpu_model(cfg, seed=71)
    batch = synth.make_batch(cfg, 32, 5, seed=72)
    want = O.model_forward(sd, cfg, batch)                    # ~15 s: 1,760 news with 512-token bodies
    got = run(model, batch, False)
    e = rel_err(got.numpy(), want.numpy())
    print('configs[3] per-GPU shape (B=32, L=512): max rel err vs oracle %.2e' % e)
    assert e < TOL and torch.equal(run(model, batch, False), got)


_KEYS = ('user_category', 'user_subCategory', 'user_title_text', 'user_title_mask', 'user_content_text', 'user_freshness',
         'user_user_topic_lifetime', 'user_history_mask', 'news_category', 'news_subCategory', 'news_title_text', 'news_title_mask',
         'news_content_text', 'news_freshness', 'news_user_topic_lifetime', 'remaining_lifetime')


def _score(model, batch, rows=None, n_src=None):
    c = [batch[k] if rows is None else batch[k][rows] for k in _KEYS]
    out = model.score_impressions(*[t.cuda() for t in c], n_src=n_src)
    torch.cuda.synchronize()
    return out.cpu()


def test_config5_slice_against_oracle_and_full_size_properties():
    cfg = make_config(vocabulary_size=50000, batch_size=1024)
    model, sd = gpu_model(cfg, seed=81)
    model.eval()
    B, K = 1024, 100
    batch = synth.make_batch(cfg, B, K, seed=82)
    n_src = min(B * K, cfg.max_history_num + cfg.batch_size)          # what one reference forward over all pairs would use
    full = _score(model, batch)
    assert full.shape == (B, K) and torch.isfinite(full).all()
    # (1) a 64-impression slice against the oracle (6,400 candidate rows; 9,600 news encoded once on the CPU)
    sl = {k: v[:64] for k, v in batch.items()}
    want = oracle_impressions.score_impressions(sd, cfg, sl, n_src=n_src)
    got = _score(model, batch, rows=slice(0, 64), n_src=n_src)
    e = rel_err(got.numpy(), want.numpy())
    print('configs[4] 64 x 100 slice: max rel err vs oracle %.2e' % e)
    assert e < TOL
    # the slice scored alone and the same impressions inside the full run (different pass boundaries and GEMM shapes)
    assert rel_err(full[:64].numpy(), got.numpy()) < 2e-5
    # (2) full size: bitwise determinism and impression-permutation equivariance
    assert torch.equal(_score(model, batch), full)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(B))
    assert torch.equal(_score(model, {k: v[perm] for k, v in batch.items()}), full[perm])
    # (3) ranking agrees with the oracle on the slice wherever its scores are separated beyond the tolerance
    agree = 0
    for r in range(64):
        o = want[r].numpy()
        if np.min(np.abs(o[:, None] - o[None, :])[~np.eye(K, dtype=bool)]) > 1e-3 * (np.abs(o).max() + 1e-6):
            assert np.array_equal(np.argsort(-got[r].numpy(), kind='stable'), np.argsort(-o, kind='stable'))
            agree += 1
    print('rankings compared on %d of 64 impressions' % agree)
