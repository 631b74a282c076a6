"""The N > 1 plumbing on CPU: two gloo ranks shard the impression rows, score their shard with the oracle (the reference's
own DDP semantics: the GraphSAGE source count is the per-rank row count, trainer.py:255 / SURVEY Q7), gather the scores
back in row order, and reduce the step time with MAX -- exactly what bench.py does on GPUs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases
from helpers import synth_state_dict
from lime_cikm25_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import json
    from helpers import load_golden
    from oracle import lime_oracle as O
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    r, w = D.init(backend='gloo')
    assert (r, w) == (rank, world)
    cfg, batch, c = golden_cases.build_case('cfg1_crown')
    g = load_golden('cfg1_crown')
    sd = synth_state_dict([(k, s) for k, s in json.loads(str(g['state_dict_spec'])) if not k.endswith('.pe')])
    B = batch['user_ID'].shape[0]
    lo, hi = D.shard_rows(B, rank, world)
    shard = type(batch)((k, v[lo:hi]) for k, v in batch.items())
    local = O.model_forward(sd, cfg, shard)
    full = D.gather_scores(local, B)
    t = D.max_over_ranks(1.0 + rank)
    dist.barrier()
    torch.save({'full': full, 'local': local, 'lo': lo, 'hi': hi, 't': t}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.destroy_process_group()


def test_shard_rows_partition():
    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [D.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_two_gloo_ranks_score_disjoint_shards(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'r0.pt', weights_only=True)
    r1 = torch.load(tmp_path / 'r1.pt', weights_only=True)
    assert (r0['lo'], r0['hi'], r1['lo'], r1['hi']) == (0, 4, 4, 8)
    assert torch.equal(r0['full'], r1['full'])                               # both ranks hold the gathered scores
    assert torch.equal(r0['full'], torch.cat([r0['local'], r1['local']]))    # in row order
    assert r0['t'] == r1['t'] == 2.0                                         # MAX over ranks


def _bucket_worker(rank, world, port, out_dir):
    """Two ranks build the model from DIFFERENT seeds; TrainStep's constructor must leave both with rank 0's parameters
    (what DistributedDataParallel does at construction, trainer.py:256)."""
    from lime_cikm25_amd import Model, make_config
    from lime_cikm25_amd.training import TrainStep
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    D.init(backend='gloo')
    cfg = make_config(max_history_num=4, max_title_length=8, max_abstract_length=8, batch_size=4, vocabulary_size=50)
    torch.manual_seed(100 + rank)
    model = Model(cfg)
    model.initialize()
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    ts = TrainStep(model)
    dist.barrier()
    named = dict(model.named_parameters())
    torch.save({'before': before, 'flat': ts.flat.clone(), 'probe': named['news_encoder.project.weight'].detach().clone()},
               os.path.join(out_dir, 'b%d.pt' % rank))
    dist.destroy_process_group()


def test_trainstep_starts_every_rank_from_rank0_parameters(tmp_path):
    port = _free_port()
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'b0.pt', weights_only=True)
    r1 = torch.load(tmp_path / 'b1.pt', weights_only=True)
    assert not torch.equal(r0['before'], r1['before'])                       # the local initialisations differed
    assert torch.equal(r0['flat'], r1['flat'])                               # ... the buckets do not
    assert torch.equal(r0['probe'], r1['probe'])                             # and the parameters are views of them
