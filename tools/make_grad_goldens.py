"""Generate tests/golden/grad_*.npz: loss and parameter gradients of one training step's forward + backward, by running
the IMPORTED REFERENCE on CPU (build container only) -- the pin of SURVEY.md section 8f row 2.

    python tools/make_grad_goldens.py

The reference model is built as in tools/make_goldens.py; ``model.eval(); model.training = True`` keeps every child in
eval mode (no dropout: the gradients are deterministic) while ``Model.forward`` takes the [B, K] training shape.  The loss
is the trainer's ``negative_log_softmax`` (trainer.py:71-73).  Stored per parameter that received a gradient: the whole
tensor when it has at most 2048 elements, else its 2048 largest-magnitude entries (flat indices + values), plus the L2 norm
and the sum; and the names whose ``.grad`` stayed None.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import ref_harness  # noqa: E402
from lime_cikm25_amd import synth  # noqa: E402
import golden_cases  # noqa: E402

CASES = ('cfg1_crown', 'cfg1_mhsa', 'spill', 'empty_history', 'full_len', 'long_body', 'two_layers')
KEEP = 2048


def run_case(name):
    cfg, batch, case = golden_cases.build_case(name)
    assert not case['eval_shape']
    torch.manual_seed(0)
    model = ref_harness.build_reference_model(cfg, synth.synth_word_embedding(cfg, golden_cases.WEIGHT_SEED))
    model.initialize()
    synth.fill_state_dict(model, golden_cases.WEIGHT_SEED)
    model.eval()
    model.training = True
    logits = model(*batch.values())
    loss = (-torch.log_softmax(logits, dim=1).select(dim=1, index=0)).mean()           # trainer.py:71-73
    loss.backward()
    out = {'loss': loss.detach().numpy(), 'logits': logits.detach().numpy()}
    with_grad, without = [], []
    seen = set()
    for k, p in model.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        if p.grad is None:
            without.append(k)
            continue
        with_grad.append(k)
        g = p.grad.detach().reshape(-1)
        out['norm:' + k] = g.double().norm().numpy()
        out['sum:' + k] = g.double().sum().numpy()
        if g.numel() <= KEEP:
            out['full:' + k] = p.grad.detach().numpy()
        else:
            idx = torch.topk(g.abs(), KEEP).indices.sort().values
            out['idx:' + k] = idx.numpy()
            out['val:' + k] = g[idx].numpy()
    out['with_grad'] = np.array(json.dumps(with_grad))
    out['without_grad'] = np.array(json.dumps(without))
    return out


def main():
    outdir = os.path.join(ROOT, 'tests', 'golden')
    for name in sys.argv[1:] or CASES:
        arrays = run_case(name)
        path = os.path.join(outdir, 'grad_' + name + '.npz')
        np.savez_compressed(path, **arrays)
        print('%-14s %7.1f KB  loss %.6f  %d tensors with grad, %d without' % (
            name, os.path.getsize(path) / 1024.0, float(arrays['loss']), len(json.loads(str(arrays['with_grad']))),
            len(json.loads(str(arrays['without_grad'])))))


if __name__ == '__main__':
    main()
