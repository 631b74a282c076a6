"""One rank of the multi-process GPU tests (tests/test_multirank_gpu.py starts it as a fresh child process; RANK / WORLD_SIZE /
MASTER_* come from the environment).  Ranks share cuda:0 on the 1-GPU box, so the group is gloo there (RCCL refuses two ranks
on one device); on a node with a GPU per rank distributed.default_backend() picks nccl."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def trainstep(out_dir):
    import torch.distributed as dist
    from lime_cikm25_amd import Model, make_config, synth
    from lime_cikm25_amd import distributed as D
    from lime_cikm25_amd.training import TrainStep, negative_log_softmax
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    D.init()
    cfg = make_config(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8, vocabulary_size=3000)
    torch.manual_seed(50 + rank)                                  # replicas start from DIFFERENT local initialisations
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, seed=7 + rank)
    model = model.cuda().train()
    ts = TrainStep(model, lr=1e-3, gradient_clip_norm=4.0)        # broadcasts rank 0's parameters
    start = ts.flat.clone()
    batch = [v.cuda() for v in synth.make_batch(cfg, 8, 3, seed=200 + rank).values()]     # every rank its own rows
    # step 1 by hand, to see the bucket on both sides of the collective
    ts.grad.zero_()
    loss = negative_log_softmax(model(*batch))
    loss.backward()
    local = ts.grad.clone()
    D.allreduce_mean_(ts.grad)
    reduced = ts.grad.clone()
    ts.update()
    after1 = ts.flat.clone()
    # step 2 through the public entry point
    loss2 = ts.step(*batch)
    torch.cuda.synchronize()
    dist.barrier()
    torch.save({'start': start.cpu(), 'local': local.cpu(), 'reduced': reduced.cpu(), 'after1': after1.cpu(), 'after2': ts.flat.cpu(),
                'loss': float(loss), 'loss2': float(loss2), 'backend': dist.get_backend()}, os.path.join(out_dir, 't%d.pt' % rank))
    dist.destroy_process_group()


if __name__ == '__main__':
    {'trainstep': trainstep}[sys.argv[1]](sys.argv[2])
