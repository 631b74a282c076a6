"""One-process-per-GPU helpers.

Scoring: impression rows are independent, so ranks score disjoint row shards with no data-path collective; the only
collectives are the barrier / max-time reduction of the benchmark and an optional all_gather of the scores for
metrics.  Training: the one exchange step of the path -- a single all-reduce of the flat gradient bucket per step
(``allreduce_mean_``), between backward and clip_grad_norm_, which is what DistributedDataParallel does for the reference
(trainer.py:250-255).  Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def default_backend():
    """"nccl" (= RCCL over xGMI on ROCm) when every local rank has a GPU of its own; "gloo" on CPU and when ranks share a device
    (RCCL refuses two ranks on one GPU: the 2-rank rehearsal on a 1-GPU box)."""
    if not torch.cuda.is_available():
        return 'gloo'
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
    return 'nccl' if torch.cuda.device_count() >= local_world else 'gloo'


def init(backend=None, device_id=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*); returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or default_backend()
        kw = {}
        if device_id is None and backend == 'nccl':
            device_id = torch.device('cuda', torch.cuda.current_device())
        if device_id is not None and backend == 'nccl':
            kw['device_id'] = device_id
        dist.init_process_group(backend=backend, **kw)
    return rank, world


def shard_rows(n_rows, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`: scoring keeps the row order, remainders go to the first ranks."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device='cpu'):
    """The slowest rank's time (what bench.py divides the total work by)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scores(local_scores, n_rows):
    """all_gather of per-shard score vectors back into row order (shards may differ by one row)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_scores
    world = dist.get_world_size()
    sizes = [shard_rows(n_rows, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local_scores.shape[1:]), dtype=local_scores.dtype, device=local_scores.device)
    pad[:local_scores.shape[0]] = local_scores
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def allreduce_mean_(flat, group=None):
    """In-place mean over ranks of a flat gradient bucket (one collective per training step; a no-op for one process)."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat.mul_(1.0 / world)


def broadcast_(flat, src=0, group=None):
    """Every rank takes rank ``src``'s values (what DistributedDataParallel does with the parameters at construction, so
    that replicas start identical whatever their local initialisation was); a no-op for one process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def sampler_rows(n_rows, rank, world, epoch_perm=None):
    """Row indices of rank `rank` for one training epoch: rows rank, rank + world, ... of the (optionally permuted) order,
    padded by wrapping around so every rank takes the same number of steps -- torch's DistributedSampler rule
    (trainer.py:293-295)."""
    order = list(range(n_rows)) if epoch_perm is None else list(epoch_perm)
    per = (n_rows + world - 1) // world
    order = [order[i % n_rows] for i in range(per * world)] if n_rows else []
    return order[rank::world]
