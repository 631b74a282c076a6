"""Golden vectors for the eval harness: runs the REFERENCE's own util.compute_scores (util.py:77-129) and
evaluate.scoring (evaluate.py:32-89) on CPU in the build container, with a fake model that returns prescribed scores,
the DataLoader / DevTest_Dataset names patched to in-memory batches and Tensor.cuda patched to the identity.
Writes tests/golden/eval_*.json: inputs (scores, impression indices, labels) and the reference's outputs (rank-file
text, the four metrics)."""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness  # noqa: E402


def make_case(seed, n_imp, ties):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(2, 40, size=n_imp)
    indices, scores, labels = [], [], []
    for i, n in enumerate(sizes):
        s = rng.normal(size=n).astype(np.float32)
        if ties:                                   # saturated lifetime weights give exact +-0.0 and repeated values
            s[rng.random(n) < 0.4] = 0.0
            s[rng.random(n) < 0.1] = -0.0
            s = np.where(rng.random(n) < 0.2, np.float32(0.5), s)
        lab = np.zeros(n, dtype=np.int64)
        lab[rng.choice(n, size=rng.integers(1, max(2, n // 3)), replace=False)] = 1
        if lab.all():
            lab[0] = 0
        indices += [i] * n
        scores += s.tolist()
        labels.append(lab.tolist())
    return indices, scores, labels


def run_reference(indices, scores, labels):
    ref_harness.import_reference()
    import util as ref_util              # the reference's util.py
    rows = len(scores)
    bs = 7
    score_t = torch.tensor(scores, dtype=torch.float32)

    class FakeModel:
        config = SimpleNamespace(category_lifetime_map=None, lifetime_type='user_topic', fixed_lifetime=0)

        def __init__(self):
            self.pos = 0

        def eval(self):
            return self

        def __call__(self, *args):
            n = args[0].size(0)
            out = score_t[self.pos:self.pos + n].unsqueeze(1)
            self.pos += n
            return out

    def fake_batches():
        for lo in range(0, rows, bs):
            n = min(bs, rows - lo)
            yield [torch.zeros(n)] * 25
    corpus = SimpleNamespace(dev_indices=indices, test_indices=indices)
    old = (ref_util.DataLoader, ref_util.DevTest_Dataset, torch.Tensor.cuda, torch.cuda.empty_cache)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'dev', 'ref'))
        with open(os.path.join(tmp, 'dev', 'ref', 'truth-synth.txt'), 'w') as f:
            f.write('\n'.join('%d %s' % (i + 1, str(l).replace(' ', '')) for i, l in enumerate(labels)))
        try:
            ref_util.DataLoader = lambda ds, **kw: fake_batches()
            ref_util.DevTest_Dataset = lambda c, mode: None
            torch.Tensor.cuda = lambda self, *a, **k: self
            torch.cuda.empty_cache = lambda: None
            os.chdir(tmp)
            metrics = ref_util.compute_scores(FakeModel(), corpus, bs, 'dev', 'ranks.txt', 'synth')
            rank_text = open('ranks.txt').read()
            truth_text = open(os.path.join('dev', 'ref', 'truth-synth.txt')).read()
        finally:
            os.chdir(cwd)
            ref_util.DataLoader, ref_util.DevTest_Dataset, torch.Tensor.cuda, torch.cuda.empty_cache = old
    return rank_text, truth_text, [float(m) for m in metrics]


def main():
    out = os.path.join(ROOT, 'tests', 'golden')
    for name, seed, n_imp, ties in (('eval_plain', 1, 60, False), ('eval_ties', 2, 80, True)):
        indices, scores, labels = make_case(seed, n_imp, ties)
        rank_text, truth_text, metrics = run_reference(indices, scores, labels)
        with open(os.path.join(out, name + '.json'), 'w') as f:
            json.dump({'indices': indices, 'scores': scores, 'labels': labels, 'rank_file': rank_text,
                       'truth_file': truth_text, 'metrics': metrics}, f)
        print(name, len(scores), 'rows', metrics)


if __name__ == '__main__':
    main()
