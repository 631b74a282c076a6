"""bench.py -- impressions scored / sec on MI355X for LIME's candidate-scoring path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2b|cfg2a|cfg1|cfg3shape]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one ``Model.forward`` (news encoder over the K candidates and H history news of every row,
CROWN user encoder, dot product x remaining-lifetime weight) over one synthetic MIND-shaped batch
that is already resident in HBM.  Default workload = BASELINE.json configs[1] with the reference's
default body length: LIME-CROWN-CROWN, batch 32, history 50, title 32 + body 128, K = 1+4, 300-d, fp32.
Impression rows are independent, so N GPUs score N batches with no data-path collective (weak scaling).

Rank 0 prints ONE JSON line: metric/value/unit ..., plus
  "roofline"     the dominant kernel (the 128x128 fp32-MFMA GEMM behind in_proj / linear1), algorithmic FLOPs per
                 launch / average launch duration from HIP events recorded in the timed region on the launch stream,
                 against the 157.3 TFLOP/s fp32 matrix peak of MI355X_MICROARCH.md;
  "cpu_baseline" the CPU oracle (oracle/lime_oracle.py, a torch-CPU port of the reference forward) timed on this
                 host's cores on the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (config overrides, B, N, description)
    'cfg2b': (dict(), 32, 5, 'MIND-shape synthetic, LIME-CROWN-CROWN, batch=32, history=50, title_len=32, body_len=128, '
                             'K=1+4, 300-d, fp32 (BASELINE.json configs[1], reference default body length)'),
    'cfg2a': (dict(content_encoder='MHSA'), 32, 5, 'MIND-shape synthetic, LIME-MHSA-CROWN (title only), batch=32, history=50, '
                                                   'title_len=32, K=1+4, 300-d, fp32'),
    'cfg1': (dict(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8), 8, 2,
             'MIND-small synthetic, batch=8, history=10, title_len=16, K=1+1 (BASELINE.json configs[0])'),
    'cfg3shape': (dict(batch_size=256), 256, 5, 'MIND-shape synthetic, batch=256, history=50, title 32 + body 128, K=1+4, '
                                                'fp32 arithmetic (configs[2] shape)'),
    'cfg3': (dict(batch_size=256, compute_dtype='bf16'), 256, 5,
             'MIND-shape synthetic, batch=256, history=50, title 32 + body 128, K=1+4, bf16 MFMA token encoders with fp32 '
             'accumulate / softmax / LayerNorm (BASELINE.json configs[2])'),
    'cfg2b_bf16': (dict(compute_dtype='bf16'), 32, 5, 'configs[1] shape (batch=32) with the bf16 token encoders of configs[2]'),
    # one training step = forward + backward + gradient all-reduce (N > 1) + clip_grad_norm_ + Adam
    'train2b': (dict(), 32, 5, 'training step (forward + backward + gradient all-reduce + clip + Adam), LIME-CROWN-CROWN, '
                               'batch=32 per GPU, history=50, title 32 + body 128, K=1+4, fp32, dropout off'),
    'train2a': (dict(content_encoder='MHSA'), 32, 5, 'training step, LIME-MHSA-CROWN (title only), batch=32 per GPU, history=50, '
                                                     'title_len=32, K=1+4, fp32, dropout off'),
    'train2b_dropout': (dict(dropout_rate=0.2), 32, 5,
                        'training step as train2b with the dropout_rate = 0.2 of the reference (config.py:78) in every encoder dropout '
                        'site (model.train()), counter-based masks'),
    'train4': (dict(max_abstract_length=512, batch_size=256), 32, 5,
               'training step at the BASELINE.json configs[3] shape per GPU (Adressa-shape: batch=32 per GPU, history=50, '
               'title 32 + body 512, K=1+4, config.batch_size=256), forward + backward + gradient all-reduce + clip + Adam, fp32, '
               'dropout off'),
    'cfg5': (dict(batch_size=1024), 1024, 100,
             'MIND-shape inference, 1024 impressions x K=100 candidates, history=50, title 32 + body 128, scoring only, '
             'eval-mode (per-candidate) semantics with every history encoded once (BASELINE.json configs[4]), fp32'),
}


def flops_per_impression(cfg, N, per_candidate_user_side=False):
    """Algorithmic FLOPs of one impression row (GEMM FLOPs = 2 m n k)."""
    H, T, L = cfg.max_history_num, cfg.max_title_length, cfg.max_abstract_length

    def seq(S):
        return S * 1334400 + 1200 * S * S
    tail = 6854800
    if cfg.content_encoder == 'CROWN':
        news = seq(T) + seq(L) + tail
    else:   # MHSA, title only
        news = T * 360000 + T * T * 800 + T * 160800 + T * 400 + 600000 + 480000
    user = lambda n: 50000 * (H + n) + 2400 * n * H + 960000 * H + 320800 * n + 320000
    if per_candidate_user_side:            # eval semantics: every candidate is its own N = 1 row of the user encoder
        return (H + N) * news + N * user(1)
    return (H + N) * news + user(N)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg2b', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--overlap-streams', action='store_true',
                    help='fork the title / body / freshness / attention-weight branches onto side streams in the timed region')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (the product path has no CPU fallback)'
    local_rank %= max(1, torch.cuda.device_count())      # (a 2-rank rehearsal on a 1-GPU box shares the device)
    torch.cuda.set_device(local_rank)
    from lime_cikm25_amd import distributed as D
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('LIME_BENCH_BACKEND', 'nccl')                  # "nccl" is RCCL on ROCm; gloo only to rehearse
        D.init(backend=backend, device_id=torch.device('cuda', local_rank) if backend == 'nccl' else None)

    from lime_cikm25_amd import Model, make_config, newsEncoders, ops, synth
    if args.overlap_streams:
        newsEncoders.OVERLAP_BRANCHES = frozenset((0, 1, 2))
    overrides, B, N, desc = WORKLOADS[args.workload]
    cfg = make_config(**overrides)
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, seed=1)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.cuda()
    model.eval()
    model.training = True                        # [B, K] candidates; every child in eval mode
    batch_cpu = synth.make_batch(cfg, B, N, seed=100 + rank)
    batch = [v.cuda() for v in batch_cpu.values()]
    step = torch.no_grad()(lambda: model(*batch))      # scoring: no autograd graph (grad mode on would take the training path)
    train = args.workload.startswith('train')
    if train:                                    # trainer.py:131-148 on the native step (flat buckets, one all-reduce)
        from lime_cikm25_amd.training import TrainStep
        if cfg.dropout_rate > 0:
            model.train()                        # every dropout active, as under trainer.py:87
        ts = TrainStep(model, lr=1e-5, gradient_clip_norm=4.0)
        step = lambda: ts.step(*batch)
    if args.workload == 'cfg5':                  # Model.score_impressions: eval semantics, histories encoded once (eager)
        model.training = False
        c = {k: v.cuda() for k, v in batch_cpu.items()}
        sargs = [c[k] for k in ('user_category', 'user_subCategory', 'user_title_text', 'user_title_mask', 'user_content_text',
                                'user_freshness', 'user_user_topic_lifetime', 'user_history_mask', 'news_category',
                                'news_subCategory', 'news_title_text', 'news_title_mask', 'news_content_text', 'news_freshness',
                                'news_user_topic_lifetime', 'remaining_lifetime')]
        step = lambda: model.score_impressions(*sargs)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The forward's ~75 launches are replayed from a HIP graph captured on the first call (Model.use_graph).
    first_loss = None
    for i in range(args.warmup):
        logits = step()
        if i == 0 and train:
            first_loss = float(logits)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step()
    barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, device='cuda')                     # the slowest rank's time
    assert torch.isfinite(logits).all()
    # Per-kernel durations: the same K steps again, launched eagerly with a HIP event pair recorded on the launch
    # stream around every lime_linear_f32 launch (events cannot be recorded inside a graph replay).  Every branch of the
    # forward runs on ONE stream here (in the timed region only the two small head branches are forked, the token-encoder
    # GEMMs are alone on the device there as well), so these durations are the quantity rocprofv3 --kernel-trace reports
    # for the same command (profiles/).
    prof = []
    if rank == 0:
        ops.PROFILE = prof
        newsEncoders.SERIAL_STREAMS = True
        if train:                                  # rank 0 alone: the same step without the collective
            from lime_cikm25_amd.training import negative_log_softmax

            def local_step():
                ts.grad.zero_()
                negative_log_softmax(model(*batch)).backward()
                ts.update()
            prof_step = local_step
        else:
            prof_step = step
        for _ in range(args.steps):
            prof_step()
        torch.cuda.synchronize()
        newsEncoders.SERIAL_STREAMS = False
        ops.PROFILE = None
    if dist is not None:
        dist.barrier()

    if rank == 0:
        value = world * B * args.steps / dt
        fimp = flops_per_impression(cfg, N, per_candidate_user_side=args.workload == 'cfg5')
        if train:
            fimp *= 3                                   # backward = input gradients + weight gradients: 2 x the forward GEMMs
        # dominant kernel = the gemm_f32_kernel instantiation with the largest total time (out_proj + linear2 of both
        # encoders: 128x320 tiles, residual in the accumulators, LayerNorm epilogue)
        by_kernel = {}
        for (name, m, n, k, e0, e1) in prof:
            d = by_kernel.setdefault(name, [0.0, 0.0, 0])
            d[0] += 2.0 * m * n * k
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
        roof = None
        if by_kernel:
            name, (fl, sec, cnt) = max(by_kernel.items(), key=lambda kv: kv[1][1])
            ach = fl / sec / 1e12
            traffic = None
            tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tfile) and args.workload == 'cfg2b':      # the PMC passes were taken on this workload's launches
                traffic = json.load(open(tfile)).get(name, {}).get('hbm_bytes_per_launch')
            is_bf16 = name.startswith('gemm_pp_kernel<') and name.split(', ')[4].startswith('true')   # gemm_pp_kernel<NTL, LN, RELU, RES, BF, PING>
            peak = PEAK_BF16_MFMA_TFLOPS if is_bf16 else PEAK_F32_MFMA_TFLOPS
            roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak,
                    'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': traffic,
                    'launches': cnt, 'avg_launch_us': round(sec / cnt * 1e6, 1), 'flops_per_launch': fl / cnt,
                    'measured': 'HIP events on the launch stream, eager pass of the same %d steps, branches on one stream' % args.steps,
                    'all_gemm_kernels': {k: {'tflops': round(v[0] / v[1] / 1e12, 2), 'avg_launch_us': round(v[1] / v[2] * 1e6, 1),
                                             'launches': v[2]} for k, v in by_kernel.items()}}
        out = {
            'metric': 'impressions trained/sec' if train else 'impressions scored/sec', 'value': round(value, 2), 'unit': 'impressions/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if getattr(cfg, 'compute_dtype', 'fp32') == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'batch_per_gpu': B, 'history': cfg.max_history_num, 'candidates': N,
                       'title_len': cfg.max_title_length, 'body_len': cfg.max_abstract_length,
                       'parallelism': ('data parallel over %d GPU(s), one all-reduce of the flat gradient bucket per step' % world) if train
                       else 'rows sharded over %d GPU(s), no data-path collective' % world,
                       'streams': 'title / body / freshness / attention-weight branches forked' if args.overlap_streams else
                       'token encoders on one stream; freshness and attention-weight branches forked beside the head'},
            'roofline': roof,
            'end_to_end': {'flops_per_impression': fimp, 'achieved_tflops': round(value * fimp / 1e12, 2),
                           'frac_of_f32_mfma_peak': round(value * fimp / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4)},
        }
        if world == 1 and not args.no_cpu_baseline and train and cfg.dropout_rate == 0:
            from oracle import lime_oracle
            ncore = min(len(os.sched_getaffinity(0)), 16)
            torch.set_num_threads(ncore)
            sd_g = {k: v.clone() for k, v in sd_cpu.items()}
            for k in ts.names:
                sd_g[k].requires_grad_(True)
            for k in list(sd_g):
                if k.startswith('user_encoder.news_encoder.'):
                    sd_g[k] = sd_g[k[len('user_encoder.'):]]
            c0 = time.perf_counter()
            lg = lime_oracle.model_forward(sd_g, cfg, batch_cpu, grad=True)
            closs = (-torch.log_softmax(lg, dim=1).select(dim=1, index=0)).mean()
            closs.backward()
            cdt = time.perf_counter() - c0
            out['cpu_baseline'] = {'value': round(B / cdt, 2), 'unit': 'impressions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                   'sample': '1 forward + backward of the same %d-impression batch (torch CPU fp32 oracle with '
                                             'autograd, no optimizer step, no warm-up)' % B,
                                   'first_step_loss_cpu': float(closs), 'first_step_loss_gpu': first_loss}
        elif world == 1 and not args.no_cpu_baseline and args.workload != 'cfg5' and not train:
            from oracle import lime_oracle
            # the GPU box gives one GPU's share of the host: 16 cores (more threads only oversubscribe)
            ncore = min(len(os.sched_getaffinity(0)), 16)
            torch.set_num_threads(ncore)
            n_warm, n_timed = 1, 3
            for _ in range(n_warm):
                lime_oracle.model_forward(sd_cpu, cfg, batch_cpu)
            c0 = time.perf_counter()
            for _ in range(n_timed):
                want = lime_oracle.model_forward(sd_cpu, cfg, batch_cpu)
            cdt = time.perf_counter() - c0
            err = float(((logits.cpu() - want).abs() / (want.abs() + want.abs()[want != 0].mean())).max())
            out['cpu_baseline'] = {'value': round(B * n_timed / cdt, 2), 'unit': 'impressions/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'sample': '%d forwards of the same %d-impression batch (torch CPU fp32 oracle, '
                                   '%d warm-up)' % (n_timed, B, n_warm), 'max_rel_err_gpu_vs_cpu': err}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
