"""lime_mean_pool_f32 / _count_f32 over the block rows pool32 leaves.   python tools/exp/mean_pool_bench.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from lime_cikm25_amd import ops


def timed(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n_seq, S in ((5013, 4), (2517, 4), (14081, 4), (14081, 16), (100000, 4)):
    x = torch.randn(n_seq * S, 304, device='cuda')[:, :300]
    out = torch.empty(n_seq, 300, device='cuda')
    cnt = torch.tensor([n_seq // 2], dtype=torch.int32, device='cuda')
    a = timed(lambda: ops.mean_pool(x, n_seq, S, out=out))
    b = timed(lambda: ops.mean_pool(x, n_seq, S, out=out, n_seq_dev=cnt))
    print('n_seq %6d S %2d  %7.1f us (%.2f TB/s)   half live %7.1f us' % (n_seq, S, a, n_seq * S * 1200 / a * 1e-6, b), flush=True)
