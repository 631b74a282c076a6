import sys, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from lime_cikm25_amd import ops, newsEncoders
w = sys.argv[1]
run = bench.Run(w, 0, 1) if hasattr(bench, 'Run') else None
for _ in range(2): run.step()
torch.cuda.synchronize()
prof = []
ops.PROFILE = prof
newsEncoders.SERIAL_STREAMS = True
run.step(); torch.cuda.synchronize()
ops.PROFILE = None
agg = collections.OrderedDict()
for (name, m, n, k, n_alg, e0, e1) in prof:
    key = (name[:60], m, n, k)
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3
for k, v in agg.items():
    print('%-62s M=%-7d N=%-5d K=%-5d x%d  %.1f us  %.1f TF' % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0], 2.0 * k[1] * k[2] * k[3] / (v[1] / v[0]) / 1e6))
