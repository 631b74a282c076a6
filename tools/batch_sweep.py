"""Throughput of the scoring forward (fp32 split-product path and bf16 encoders) over the batch size: the same MIND-shaped synthetic
impressions (H = 50, T = 32, L = 128, K = 5) as BASELINE configs[1] / [2], B = 8 .. 512.  One line per point.
    python tools/batch_sweep.py [logfile]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

LOG = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None


def say(msg):
    print(msg, flush=True)
    if LOG:
        LOG.write(msg + '\n')
        LOG.flush()


say('# impressions/s of Model.forward (HIP-graph replay, inputs resident), 1 x MI355X; B impressions x (5 candidates + 50 history slots)')
for dtype in ('fp32', 'bf16'):
    for B in (8, 16, 32, 64, 128, 256, 512):
        name = 'sweep_%s_%d' % (dtype, B)
        over = dict(batch_size=max(B, 64))
        if dtype == 'bf16':
            over['compute_dtype'] = 'bf16'
        bench.WORKLOADS[name] = (over, B, 5, 'batch sweep')
        run = bench.Run(name, 0, 1)
        for _ in range(5):
            run.step()
        torch.cuda.synchronize()
        steps = max(20, int(1.0 / (0.0006 * B ** 0.8)))
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        say('%s  B %4d   %8.3f ms/step   %9.0f impressions/s   (%d steps)' % (dtype, B, dt * 1e3, B / dt, steps))
        del run
        torch.cuda.empty_cache()
