"""Soak run: thousands of replays of the scoring forward (fp32 split-product path, bf16 path, MHSA, 1024 x 100 layout) on FIXED inputs --
every result must equal the first one bit for bit -- and two independent training runs from the same initial state (config 2b without and
with the reference's dropout, the configs[3] shape), whose final parameters must agree bit for bit.  A sporadic wrong lane (the round-2 event recorded in profiles/r02_notes.md) would show up here as a
mismatch.    python tools/soak.py [scale [logfile]]      -> one line per workload"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
LOG = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None       # progress also into a file (a long run behind a pipe looks hung to gpurun)


def say(msg):
    print(msg, flush=True)
    if LOG:
        LOG.write(msg + '\n')
        LOG.flush()


for name, reps in (('cfg2b', 4000), ('cfg3', 1500), ('cfg2a', 4000), ('cfg5', 24)):
    run = bench.Run(name, 0, 1)
    run.batches = run.batches[:1]                    # one fixed batch
    ref = run.step().clone()
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    t0 = time.time()
    n = max(2, int(reps * scale))
    for i in range(n):
        out = run.step()
        bad += (out != ref).any()                     # EVERY replay is compared, on the device; one read-back at the end
        if i % 5000 == 4999:
            say('  %s: %d replays, %d mismatches so far' % (name, i + 1, int(bad)))
    torch.cuda.synchronize()
    say('%-6s %6d replays in %6.1f s: %d mismatching results (every replay compared bit for bit), finite: %s' % (
        name, n, time.time() - t0, int(bad), bool(torch.isfinite(ref).all())))
    del run
    torch.cuda.empty_cache()

for wl, base in (('train2b', 60), ('train2b_dropout', 30), ('train4', 8)):
    finals = []
    for attempt in range(2):
        torch.manual_seed(0)
        run = bench.Run(wl, 0, 1)
        steps = max(2, int(base * scale))
        for _ in range(steps):
            loss = run.step()
        torch.cuda.synchronize()
        finals.append((float(loss), torch.cat([p.detach().reshape(-1) for p in run.model.parameters()]).clone()))
        del run
        torch.cuda.empty_cache()
    same = torch.equal(finals[0][1], finals[1][1])
    say('%s two runs of %d steps from the same state: final loss %.6f / %.6f, %d parameters bitwise equal: %s' % (
        wl, steps, finals[0][0], finals[1][0], finals[0][1].numel(), same))
