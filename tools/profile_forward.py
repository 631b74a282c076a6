"""Per-op timing of one eager forward at a bench workload: wraps every lime_cikm25_amd.ops function with a HIP event
pair and prints the ops in call order with their shapes (run on the GPU box)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import Model, make_config, ops, synth  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg2b')
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    overrides, B, N, desc = bench.WORKLOADS[args.workload]
    cfg = make_config(**overrides)
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, seed=1)
    model = model.cuda().eval()
    model.training = True
    model.use_graph = False
    batch = [v.cuda() for v in synth.make_batch(cfg, B, N, seed=100).values()]
    records = []
    names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith('_') and n not in ('check', 'inv_sqrt', 'linear_kernel_name')
             and getattr(getattr(ops, n), '__module__', '') == ops.__name__]

    def wrap(name, fn):
        def w(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            shp = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:3]
            records.append((name, shp, k.get('act'), e0, e1))
            return out
        return w
    for n in names:
        setattr(ops, n, wrap(n, getattr(ops, n)))
    torch.set_grad_enabled(False)
    for _ in range(2):
        model(*batch)
    torch.cuda.synchronize()
    records.clear()
    for _ in range(args.iters):
        model(*batch)
    torch.cuda.synchronize()
    per = len(records) // args.iters
    tot = 0.0
    for i in range(per):
        us = sum(records[i + j * per][3].elapsed_time(records[i + j * per][4]) for j in range(args.iters)) * 1e3 / args.iters
        tot += us
        name, shp, act, _, _ = records[i]
        print('%3d %-20s %8.1f us  %s %s' % (i, name, us, shp, act or ''))
    print('sum of op times %.1f us over %d ops' % (tot, per))


if __name__ == '__main__':
    main()
