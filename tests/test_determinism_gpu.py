"""Bitwise reproducibility of the training step (round 2): the word-table gradient is a sort + segmented sum in a fixed order
(lime_embed_bwd_sorted_f32) instead of float atomics, so two runs of the same step from the same state give the same bits --
for sequences of at most 128 tokens (the blocked attention backward of longer bodies still adds dq with atomics)."""
import numpy as np
import pytest
import torch

from lime_cikm25_amd import Model, make_config, ops, synth
from lime_cikm25_amd.training import TrainStep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,vocab,dim,pad_share', [(5000, 300, 300, 0.5), (70000, 20000, 300, 0.45), (257, 5, 64, 0.0), (1000, 40, 320, 0.9),
                                                      (256, 1000, 300, 0.0), (513, 2, 300, 0.0)])
def test_sorted_scatter_equals_index_add_and_is_reproducible(rows, vocab, dim, pad_share):
    g = torch.Generator().manual_seed(rows + vocab)
    ids = torch.randint(1 if vocab > 1 else 0, vocab, (rows,), generator=g, dtype=torch.int32)
    ids[torch.rand(rows, generator=g) < pad_share] = 0
    dx = torch.randn(rows, dim, generator=g)
    want = torch.zeros(max(vocab, 33), dim, dtype=torch.float64).index_add_(0, ids.long(), dx.double())
    outs = []
    for _ in range(2):
        dt = torch.zeros(max(vocab, 33), dim, device='cuda')
        ops.embed_bwd(ids.cuda(), dx.cuda(), dt)
        torch.cuda.synchronize()
        outs.append(dt.cpu())
    assert torch.equal(outs[0], outs[1])
    scale = float(want.abs().max())
    assert float((outs[0].double() - want).abs().max()) < 2e-6 * scale * max(1.0, np.sqrt(rows / max(1, vocab)))
    untouched = torch.ones(max(vocab, 33), dtype=torch.bool)
    untouched[ids.long().unique()] = False
    assert bool((outs[0][untouched] == 0).all())


def test_training_step_is_bitwise_reproducible():
    cfg = make_config(vocabulary_size=5000, max_history_num=20, max_title_length=32, max_abstract_length=128, batch_size=16)
    batch = None
    flats, losses = [], []
    for _ in range(2):
        torch.manual_seed(0)
        model = Model(cfg)
        model.initialize()
        synth.fill_state_dict(model, seed=12)
        model = model.cuda()
        model.eval()
        model.training = True                      # [B, K] shape, dropout off
        ts = TrainStep(model, lr=1e-3, gradient_clip_norm=4.0)
        if batch is None:
            batch = [v.cuda() for v in synth.make_batch(cfg, 16, 5, seed=13).values()]
        for _ in range(3):
            losses.append(float(ts.step(*batch)))
        torch.cuda.synchronize()
        flats.append(ts.flat.clone())
    assert losses[:3] == losses[3:]
    assert torch.equal(flats[0], flats[1])
