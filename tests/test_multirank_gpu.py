"""The N > 1 path on the GPU box (reference main.py:28 + trainer.py:246-256): fresh rank processes, one
process group, the native TrainStep on every rank -- identical buckets after the all-reduce, equal to the mean of the local
gradients -- and ``python bench.py --gpus 2`` launching its own ranks from a plain shell.

The 1-GPU box has one device, so the two ranks share it and the group is gloo (RCCL refuses two ranks on one GPU); the code path
(distributed.init -> TrainStep.broadcast -> allreduce_mean_ on the CUDA bucket) is the one an N-GPU node runs over RCCL.

These tests start child processes, which must not happen from a process that has initialised the GPU: conftest.py moves them to
the FRONT of the session and this module never touches the device itself."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(args, world, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=timeout)[0].decode(errors='replace') for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs


def test_trainstep_two_ranks_end_with_identical_buckets(tmp_path):
    _launch([os.path.join(ROOT, 'tests', '_rank_worker.py'), 'trainstep', str(tmp_path)], 2)
    r0 = torch.load(tmp_path / 't0.pt', weights_only=True)
    r1 = torch.load(tmp_path / 't1.pt', weights_only=True)
    assert torch.equal(r0['start'], r1['start'])                             # constructor broadcast: rank 0's parameters everywhere
    assert not torch.equal(r0['local'], r1['local'])                         # different rows -> different local gradients
    assert torch.equal(r0['reduced'], r1['reduced'])                         # one all-reduce -> the same bucket on every rank
    want = (r0['local'].double() + r1['local'].double()) / 2
    err = float((r0['reduced'].double() - want).abs().max() / want.abs().max())
    assert err < 1e-6, err                                                   # ... equal to the mean
    assert torch.equal(r0['after1'], r1['after1']) and torch.equal(r0['after2'], r1['after2'])   # replicas stay in lock step
    assert not torch.equal(r0['after1'], r0['start'])
    assert r0['loss'] != r1['loss'] and all(map(lambda v: v == v, (r0['loss2'], r1['loss2'])))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 --workload train2b` from a plain shell (no torchrun): rc 0, one JSON line, n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'train2b', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors='replace')[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['world_size'] == 2 and out['value'] > 0
    assert out['metric'] == 'impressions trained/sec' and out['config']['backend'] in ('nccl', 'gloo')
