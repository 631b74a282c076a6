"""Every lime_linear_f32 / wgrad launch of one step of a bench workload with its shape and HIP-event duration.
    python tools/exp/linear_shapes.py cfg3"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from lime_cikm25_amd import ops, newsEncoders

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
if name.startswith('B'):                              # e.g. B128: the config-2b shape at another batch size
    bench.WORKLOADS[name] = (dict(batch_size=max(int(name[1:]), 64)), int(name[1:]), 5, 'batch sweep')
run = bench.Run(name, 0, 1)
run.step(); run.step()
prof = []
ops.PROFILE = prof
newsEncoders.SERIAL_STREAMS = True
if run.train:
    from lime_cikm25_amd.training import negative_log_softmax
    b = run.batches[0]
    run.ts.backward(negative_log_softmax(run.model(*b))); run.ts.update()
else:
    run.step()
torch.cuda.synchronize()
ops.PROFILE = None
for rec in prof:
    nm, m, n, k, n_alg, e0, e1 = rec[:7]
    us = e0.elapsed_time(e1) * 1e3
    print('%-70s M %7d N %5d K %5d  %8.1f us  %6.1f TF' % (nm[:70], m, n, k, us, 2.0 * m * n * k / us * 1e-6))
