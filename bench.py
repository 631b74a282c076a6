"""bench.py -- impressions scored / sec on MI355X for LIME's candidate-scoring path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2b|cfg2a|cfg1|cfg3|cfg5|train2b|...]

With ``--gpus N`` (N > 1) from a plain shell the script launches its N ranks itself: N fresh child processes, spawned
BEFORE this process makes any GPU call, each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what the reference does
with ``mp.spawn`` in main.py:28 and ``init_process_group(backend='nccl', init_method='env://')`` in trainer.py:246-256).
Under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` the ranks already exist and are used as
they are.  One process per GPU, backend "nccl" (= RCCL over xGMI on ROCm).

A step = one ``Model.forward`` (news encoder over the K candidates and H history news of every row, CROWN user encoder,
dot product x remaining-lifetime weight) over one synthetic MIND-shaped batch that is already resident in HBM.  Default
workload = BASELINE.json configs[1] with the reference's default body length: LIME-CROWN-CROWN, batch 32, history 50,
title 32 + body 128, K = 1+4, 300-d, fp32.  Impression rows are independent, so N GPUs score N batches with no data-path
collective (weak scaling); the training workloads add the one exchange step of the path, the all-reduce of the flat
gradient bucket.

Rank 0 prints ONE JSON line: metric / value / unit ... (value = exactly K timed steps between two barriers), plus
  "roofline"        the dominant kernel: ALGORITHMIC FLOPs per launch (2 m n k on the unpadded problem: in_proj counts its
                    900 useful columns, not the 960 the kernel computes) / average launch duration from HIP events recorded
                    on the launch stream, against the 157.3 TFLOP/s fp32-matrix peak of MI355X_MICROARCH.md; "frac_padded"
                    counts the padded columns as well; "traffic" is NOT measured in this run -- it is the PMC figure of the
                    rocprofv3 pass named in "traffic_source";
  "sustained"       the same step over a >= 2 s region (>= 100 steps), batches rotating;
  "value_with_h2d"  the same K steps with the 26 input tensors copied from pinned host memory inside every step
                    (trainer.py:93-118, util.py:94);
  "cpu_baseline"    the CPU oracle (oracle/lime_oracle.py, a torch-CPU port of the reference forward) timed on this
                    host's cores on the same workload (rank 0, N = 1 only): 3 warm-ups + 5 forwards;
  "also"            (default workload, N = 1) the other BASELINE.json configurations in the same run: cfg2a (configs[1]
                    read literally: title only), cfg3 (configs[2]: batch 256, bf16), cfg5 (configs[4]: 1024 x 100 scoring),
                    train2b (one training step at the configs[1] shape), train4 (configs[3] per-GPU shape).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 (the 5 PF headline figure includes 2:1 sparsity)
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6     # fp32-level products as six bf16 MFMAs each (csrc/gemm_sp_f32.hip)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (config overrides, B, N, description)
    'cfg2b': (dict(), 32, 5, 'MIND-shape synthetic, LIME-CROWN-CROWN, batch=32, history=50, title_len=32, body_len=128, '
                             'K=1+4, 300-d, fp32 (BASELINE.json configs[1], reference default body length)'),
    'cfg2a': (dict(content_encoder='MHSA'), 32, 5, 'MIND-shape synthetic, LIME-MHSA-CROWN (title only), batch=32, history=50, '
                                                   'title_len=32, K=1+4, 300-d, fp32 (BASELINE.json configs[1] read literally)'),
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
    'cfg4fwd': (dict(max_abstract_length=512, batch_size=256), 32, 5,
                'scoring forward at the BASELINE.json configs[3] shape per GPU (batch=32, history=50, title 32 + body 512, K=1+4), fp32'),
    'cfg5': (dict(batch_size=1024), 1024, 100,
             'MIND-shape inference, 1024 impressions x K=100 candidates, history=50, title 32 + body 128, scoring only, '
             'eval-mode (per-candidate) semantics with every history encoded once (BASELINE.json configs[4]), fp32'),
}
# (workload, steps, warmup) run beside the default line so that the driver's own run times them
ALSO = (('cfg2a', 100, 10), ('cfg3', 30, 5), ('cfg5', 3, 1), ('train2b', 30, 5), ('train2b_dropout', 10, 3), ('train4', 8, 2))
N_BATCHES = 4           # distinct resident batches a scoring run rotates through


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


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh ranks, spawned before this process touches the GPU
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Spawn ``n`` child processes of this script (rank r on GPU r) and return the worst exit code.  The parent never
    initialises the GPU (no torch import even): children are plain ``subprocess`` spawns, nothing is exec'ed over a process
    that holds the device.  Rank 0's stdout (the JSON line) is passed through; every rank shares stderr."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), LIME_BENCH_SPAWNED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# one workload on this rank
# ---------------------------------------------------------------------------------------------------------------------
class Run:
    """Model + resident batches + the step closure of one workload."""

    def __init__(self, name, rank, world, overlap_streams=False):
        import torch
        from lime_cikm25_amd import Model, make_config, newsEncoders, synth
        self.name, self.rank, self.world = name, rank, world
        overrides, self.B, self.N, self.desc = WORKLOADS[name]
        if overlap_streams:
            newsEncoders.OVERLAP_BRANCHES = frozenset((0, 1, 2))
        self.cfg = cfg = make_config(**overrides)
        self.train = name.startswith('train')
        model = Model(cfg)
        model.initialize()
        synth.fill_state_dict(model, seed=1)
        self.sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
        self.model = model = model.cuda()
        model.eval()
        model.training = True                        # [B, K] candidates; every child in eval mode
        nb = 1 if (self.train or name == 'cfg5') else N_BATCHES
        self.batches_cpu = [synth.make_batch(cfg, self.B, self.N, seed=100 + rank + 1000 * i) for i in range(nb)]
        self.batches = [[v.cuda() for v in b.values()] for b in self.batches_cpu]
        self.i = 0
        self.ts = None
        if self.train:                                # trainer.py:131-148 on the native step (flat buckets, one all-reduce)
            from lime_cikm25_amd.training import TrainStep
            if cfg.dropout_rate > 0:
                model.train()                         # every dropout active, as under trainer.py:87
            self.ts = TrainStep(model, lr=1e-5, gradient_clip_norm=4.0)
            self._step = lambda b: self.ts.step(*b)
        elif name == 'cfg5':                          # Model.score_impressions: eval semantics, histories encoded once (eager)
            model.training = False
            keys = ('user_category', 'user_subCategory', 'user_title_text', 'user_title_mask', 'user_content_text',
                    'user_freshness', 'user_user_topic_lifetime', 'user_history_mask', 'news_category', 'news_subCategory',
                    'news_title_text', 'news_title_mask', 'news_content_text', 'news_freshness', 'news_user_topic_lifetime',
                    'remaining_lifetime')
            c = {k: v.cuda() for k, v in self.batches_cpu[0].items()}
            sargs = [c[k] for k in keys]
            self._step = lambda b: model.score_impressions(*sargs)
        else:
            nograd = torch.no_grad()
            self._step = nograd(lambda b: model(*b))  # scoring: no autograd graph (grad mode on would take the training path)

    def step(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return self._step(b)

    def step_h2d(self, pinned):
        """The step with the host->device copy of the batch inside it (all 26 tensors, as trainer.py:93-118 moves them)."""
        p = pinned[self.i % len(pinned)]
        self.i += 1
        return self._step([t.cuda(non_blocking=True) for t in p])


def executed_flops_per_impression(run):
    """FLOPs the DEFAULT path executes per impression on this run's batches: as flops_per_impression, with the token encoders
    counted over what newsEncoders.encode_tokens_compact really runs -- the live sequences + one all-padding representative
    through the layer, in_proj over the live tokens only.  Also returns the batches' padding statistics."""
    cfg = run.cfg
    if cfg.content_encoder != 'CROWN' or getattr(cfg, 'compute_dtype', 'fp32') != 'fp32' or run.name == 'cfg5':
        return None, None
    import torch
    H, T, L = cfg.max_history_num, cfg.max_title_length, cfg.max_abstract_length
    tot, stats = 0.0, {'title': [0, 0, 0, 0], 'body': [0, 0, 0, 0]}
    for b in run.batches_cpu:
        for key, S, (ck, uk) in (('title', T, ('news_title_text', 'user_title_text')), ('body', L, ('news_content_text', 'user_content_text'))):
            ids = torch.cat([b[ck].reshape(-1, S), b[uk].reshape(-1, S)])
            live_seq = int((ids != 0).any(dim=1).sum())
            live_tok = int((ids != 0).sum())
            n_c = live_seq + 1
            tot += n_c * (S * (2 * 300 * 300 + 4 * 300 * 512) + 1200 * S * S) + live_tok * 2 * 300 * 900
            st = stats[key]
            st[0] += ids.shape[0]; st[1] += live_seq; st[2] += ids.numel(); st[3] += live_tok
    nb = len(run.batches_cpu)
    N = run.N
    user = 50000 * (H + N) + 2400 * N * H + 960000 * H + 320800 * N + 320000
    per_imp = tot / nb / run.B + (H + N) * 6854800 + user
    pad = {k: {'sequences': v[0] // nb, 'live_sequences': round(v[1] / nb, 1), 'live_token_fraction': round(v[3] / v[2], 4)} for k, v in stats.items()}
    return per_imp, pad


def timed(run, steps, warmup, barrier, step=None):
    step = step or run.step
    out = None
    for _ in range(warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    return time.perf_counter() - t0, out


def kernel_profile(run, steps):
    """Per-kernel durations: ``steps`` more steps launched eagerly with a HIP event pair recorded on the launch stream around
    every lime_linear_f32 launch (events cannot be recorded inside a graph replay).  Every branch of the forward runs on ONE
    stream here (in the timed region only the two small head branches are forked; the token-encoder GEMMs are alone on the
    device there as well), so these durations are what rocprofv3 --kernel-trace reports for the same command (profiles/)."""
    import torch
    from lime_cikm25_amd import newsEncoders, ops
    prof = []
    ops.PROFILE = prof
    newsEncoders.SERIAL_STREAMS = True
    try:
        if run.train:                                  # rank 0 alone: the same step without the collective
            from lime_cikm25_amd.training import negative_log_softmax
            ts, model, b = run.ts, run.model, run.batches[0]
            for _ in range(steps):
                ts.backward(negative_log_softmax(model(*b)))
                ts.update()
        else:
            for _ in range(steps):
                run.step()
        torch.cuda.synchronize()
    finally:
        newsEncoders.SERIAL_STREAMS = False
        ops.PROFILE = None
    by_kernel = {}
    for rec in prof:
        name, m, n, k, n_alg, e0, e1 = rec[:7]
        gathered = rec[7] if len(rec) > 7 else 0
        d = by_kernel.setdefault(name, [0.0, 0.0, 0.0, 0, 0.0, 0.0])
        d[0] += 2.0 * m * n_alg * k                    # algorithmic: the useful output columns
        d[1] += 2.0 * m * n * k                        # what the kernel computes (head padding included)
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += 1
        d[5] += gathered                               # bytes of table rows the launch gathered as its A operand (a_ids)
    return by_kernel


def gather_figures(run, by_kernel):
    """The gather-bound embedding lookup of the path (newsEncoders.py:311-312; SURVEY.md 8d "K1"), against the HBM roofline:
    (1) the stand-alone gather kernel (lime_embed_pe_f32: table rows by id + positional rows -> [tokens, E]; what the training path
    with dropout runs) timed here on this workload's body ids with HIP events; (2) on the scoring path the gather is FUSED into the
    in_proj GEMM's A fetch (an LDS-DMA per table row): bytes gathered per launch over the launch's duration -- a lower bound on the
    rate the fetch sustains, the launch itself being bound by the matrix pipe."""
    import torch
    from lime_cikm25_amd import ops
    cfg = run.cfg
    if cfg.content_encoder != 'CROWN':
        return None
    enc = run.model.news_encoder.base_news_encoder
    table, pe = enc.word_embedding.weight, enc.body_pos_encoder.table()
    b = run.batches_cpu[0]
    ids = torch.cat([b['news_content_text'].reshape(-1), b['user_content_text'].reshape(-1)]).to(torch.int32).cuda()
    L, E = cfg.max_abstract_length, table.shape[1]
    out = torch.empty((ids.numel(), E), dtype=torch.float32, device='cuda')
    for _ in range(3):
        ops.embed_pe(ids, table, pe, L, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.embed_pe(ids, table, pe, L, out=out)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    nbytes = ids.numel() * (4 + 2 * E * 4)                    # id + gathered row read, result row written (the positional rows stay in cache)
    res = {'embed_pe_kernel': {'tokens': ids.numel(), 'bytes': nbytes, 'us': round(sec * 1e6, 1), 'GB_per_s': round(nbytes / sec / 1e9, 1),
                               'frac_of_hbm_peak': round(nbytes / sec / 1e9 / PEAK_HBM_GBS, 4), 'peak_GB_per_s': PEAK_HBM_GBS,
                               'note': 'stand-alone gather + positional add (lime_embed_pe_f32) over every body token of one batch, padding '
                                       'included; bytes = ids + gathered rows + result rows'}}
    fused = [(k, v) for k, v in by_kernel.items() if v[5] > 0]
    if fused:
        k, v = max(fused, key=lambda kv: kv[1][2])
        res['fused_in_proj_gather'] = {'kernel': k, 'gathered_bytes_per_launch': int(v[5] / v[3]), 'avg_launch_us': round(v[2] / v[3] * 1e6, 1),
                                       'GB_per_s': round(v[5] / v[2] / 1e9, 1), 'frac_of_hbm_peak': round(v[5] / v[2] / 1e9 / PEAK_HBM_GBS, 4),
                                       'note': 'word rows gathered by the in_proj GEMM itself (a_ids: one LDS-DMA per row and k chunk, three column '
                                               'blocks re-gather a row panel, mostly from L2); the launch is bound by the matrix pipe, so this is '
                                               'the rate the gather NEEDS there, not what the fetch path can sustain'}
    return res


def roofline(by_kernel, workload, steps):
    if not by_kernel:
        return None
    name, (fl, flp, sec, cnt, _x, _g) = max(by_kernel.items(), key=lambda kv: kv[1][2])
    ach = fl / sec / 1e12
    traffic, source = None, None
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json' if workload == 'cfg2b' else 'traffic_%s.json' % workload)
    if os.path.exists(tfile):                               # the PMC passes were taken on this workload's launches
        tj = json.load(open(tfile))
        traffic = tj.get(name, {}).get('hbm_bytes_per_launch')
        source = tj.get('_source', 'profiles/traffic.json')
    is_bf16 = ((name.startswith('gemm_pp_kernel<') and name.split(', ')[4].startswith('true')) or     # <NTL, LN, RELU, RES, BF, ...>
               name.startswith('ffn_bf16_kernel<'))                                                    # the fused bf16 encoder block
    is_split = name.startswith('gemm_sp_kernel<')
    peak = PEAK_BF16_MFMA_TFLOPS if is_bf16 else (PEAK_SPLIT_TFLOPS if is_split else PEAK_F32_MFMA_TFLOPS)
    extra = {}
    if is_split:
        extra = {'peak_note': 'fp32 operands, each product formed as six bf16 MFMAs (csrc/gemm_sp_f32.hip): the pipe that bounds the '
                              'kernel is the bf16 matrix pipe, whose %.0f TFLOP/s dense peak delivers %.1f TFLOP/s of fp32-level products; '
                              'achieved counts the algorithmic 2 m n k once' % (PEAK_BF16_MFMA_TFLOPS, PEAK_SPLIT_TFLOPS),
                 'achieved_bf16_mfma_tflops': round(6 * flp / sec / 1e12, 1),
                 'frac_of_f32_mfma_peak': round(ach / PEAK_F32_MFMA_TFLOPS, 4)}
    return {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(ach / peak, 4), 'frac_padded': round(flp / sec / 1e12 / peak, 4), **extra,
            'traffic': traffic, 'traffic_source': (source + ' -- a rocprofv3 --pmc pass of an earlier run of this command, NOT measured '
                                                   'in this run') if traffic is not None else None,
            'launches': cnt, 'avg_launch_us': round(sec / cnt * 1e6, 1), 'flops_per_launch': fl / cnt,
            'flops_per_launch_padded': flp / cnt,
            'measured': 'HIP events on the launch stream, eager pass of %d steps, branches on one stream' % steps,
            'all_gemm_kernels': {k: {'tflops': round(v[0] / v[2] / 1e12, 2), 'avg_launch_us': round(v[2] / v[3] * 1e6, 1),
                                     'launches': v[3]} for k, v in by_kernel.items()}}


def cpu_baseline(run, logits, first_loss):
    """The oracle on this host's cores: 3 warm-ups + 5 timed forwards of the workload's batches (training: one forward +
    backward, no warm-up -- a step takes ~6 s)."""
    import torch
    from oracle import lime_oracle
    cfg, B = run.cfg, run.B
    ncore = min(len(os.sched_getaffinity(0)), 16)         # the GPU box gives one GPU's share of the host: 16 cores
    torch.set_num_threads(ncore)
    if run.train:
        sd_g = {k: v.clone() for k, v in run.sd_cpu.items()}
        for k in run.ts.names:
            sd_g[k].requires_grad_(True)
        for k in list(sd_g):
            if k.startswith('user_encoder.news_encoder.'):
                sd_g[k] = sd_g[k[len('user_encoder.'):]]
        c0 = time.perf_counter()
        lg = lime_oracle.model_forward(sd_g, cfg, run.batches_cpu[0], grad=True)
        closs = (-torch.log_softmax(lg, dim=1).select(dim=1, index=0)).mean()
        closs.backward()
        cdt = time.perf_counter() - c0
        return {'value': round(B / cdt, 2), 'unit': 'impressions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                'sample': '1 forward + backward of the same %d-impression batch (torch CPU fp32 oracle with autograd, no optimizer '
                          'step, no warm-up)' % B, 'first_step_loss_cpu': float(closs), 'first_step_loss_gpu': first_loss}
    n_warm, n_timed = 3, 5
    nb = len(run.batches_cpu)
    for i in range(n_warm):
        lime_oracle.model_forward(run.sd_cpu, cfg, run.batches_cpu[i % nb])
    c0 = time.perf_counter()
    for i in range(n_timed):
        want = lime_oracle.model_forward(run.sd_cpu, cfg, run.batches_cpu[i % nb])
    cdt = time.perf_counter() - c0
    with torch.no_grad():
        got = run.model(*run.batches[(n_timed - 1) % nb]).cpu()
    err = float(((got - want).abs() / (want.abs() + want.abs()[want != 0].mean())).max())
    return {'value': round(B * n_timed / cdt, 2), 'unit': 'impressions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d forwards over %d distinct %d-impression batches of the workload (torch CPU fp32 oracle, %d warm-ups)'
                      % (n_timed, nb, B, n_warm), 'max_rel_err_gpu_vs_cpu': err}


def bench_workload(name, steps, warmup, rank, world, dist, D, args, full=True):
    """Time one workload; returns the result dict on rank 0 (None elsewhere)."""
    import torch
    run = Run(name, rank, world, overlap_streams=args.overlap_streams)
    cfg, B, N, train = run.cfg, run.B, run.N, run.train

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = run.step()                                   # captures the HIP graph / builds the buckets
    first_loss = float(first) if train else None
    dt, logits = timed(run, steps, max(0, warmup - 1), barrier)
    dt = D.max_over_ranks(dt, device='cuda')             # the slowest rank's time
    assert torch.isfinite(logits).all()
    sustained = h2d = None
    if full and name != 'cfg5':
        # >= 2 s and >= 100 steps, batches rotating (the K-step region above is what `value` reports)
        n_sus = int(min(2000, max(100, 2.2 / (dt / steps))))
        sdt, _ = timed(run, n_sus, 0, barrier)
        sdt = D.max_over_ranks(sdt, device='cuda')
        sustained = {'steps': n_sus, 'seconds': round(sdt, 3), 'ms_per_step': round(sdt / n_sus * 1e3, 4),
                     'value': round(world * B * n_sus / sdt, 2)}
        pinned = [[t.pin_memory() for t in b.values()] for b in run.batches_cpu]
        hdt, _ = timed(run, steps, 2, barrier, step=lambda: run.step_h2d(pinned))
        hdt = D.max_over_ranks(hdt, device='cuda')
        nbytes = sum(t.numel() * t.element_size() for t in pinned[0])
        h2d = {'value_with_h2d': round(world * B * steps / hdt, 2), 'ms_per_step_with_h2d': round(hdt / steps * 1e3, 4),
               'h2d_bytes_per_step': nbytes,
               'h2d_note': 'all 26 input tensors copied from pinned host memory with .cuda(non_blocking=True) inside every step'}
    dense = None
    fexec, pad_stats = executed_flops_per_impression(run)
    if not train:
        # the same K steps with the repetition shortcuts OFF: every token of every slot through the encoder layer, as the
        # reference computes it (newsEncoders.py:311-321) -- what LIME_DENSE_TOKENS=1 runs
        from lime_cikm25_amd import newsEncoders
        newsEncoders.DEDUP = False
        run.model._graphs.clear()
        try:
            ddt, dlogits = timed(run, steps, 3, barrier)
            ddt = D.max_over_ranks(ddt, device='cuda')
            dense_batch = (run.i - 1) % len(run.batches)        # the batch the last dense step scored
            dlogits = dlogits.float().cpu()
        finally:
            newsEncoders.DEDUP = True
            run.model._graphs.clear()
        run.i = dense_batch                                  # the default path on that very batch
        default_logits = run.step().float().cpu()
        dense = {'value': round(world * B * steps / ddt, 2), 'ms_per_step': round(ddt / steps * 1e3, 4),
                 'note': 'LIME_DENSE_TOKENS=1: every token of every history / candidate slot goes through the encoder layer, padding '
                         'news and padding tokens included (what the reference computes); `value` is the default path, which encodes '
                         'all-padding sequences once and runs in_proj over the live tokens -- same logits',
                 'max_abs_logit_difference_vs_default': float((dlogits - default_logits).abs().max()),
                 'mean_abs_logit': float(default_logits.abs().mean())}
    by_kernel = kernel_profile(run, min(steps, 20) if full else min(steps, 5)) if rank == 0 else {}
    if dist is not None:
        dist.barrier()
    if rank != 0:
        return None
    value = world * B * steps / dt
    fimp = flops_per_impression(cfg, N, per_candidate_user_side=name == 'cfg5')
    if train:
        fimp *= 3                                       # backward = input gradients + weight gradients: 2 x the forward GEMMs
    bf16 = getattr(cfg, 'compute_dtype', 'fp32') == 'bf16'
    out = {
        'metric': 'impressions trained/sec' if train else 'impressions scored/sec', 'value': round(value, 2), 'unit': 'impressions/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
        'config': {'workload': run.desc, 'batch_per_gpu': B, 'history': cfg.max_history_num, 'candidates': N,
                   'title_len': cfg.max_title_length, 'body_len': cfg.max_abstract_length,
                   'parallelism': ('data parallel over %d GPU(s), one all-reduce of the flat gradient bucket per step' % world) if train
                   else 'rows sharded over %d GPU(s), no data-path collective' % world,
                   'backend': (dist.get_backend() if dist is not None else None), 'world_size': world,
                   'distinct_batches': len(run.batches),
                   'streams': 'title / body / freshness / attention-weight branches forked' if args.overlap_streams else
                   'token encoders on one stream; freshness and attention-weight branches forked beside the head'},
        'roofline': roofline(by_kernel, name, min(steps, 20) if full else min(steps, 5)),
        'gather': gather_figures(run, by_kernel) if (rank == 0 and full and not train) else None,
        'end_to_end': {'flops_per_impression': fimp, 'achieved_tflops': round(value * fimp / 1e12, 2),
                       'frac_of_f32_mfma_peak': round(value * fimp / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4)},
    }
    if not train:
        out['dense'] = dense
    if fexec is not None and not train:
        e2e = out['end_to_end']
        e2e['note'] = ('flops_per_impression is the DENSE algorithmic count of BASELINE.md section 3 (every slot, every token); the '
                       'default path executes flops_per_impression_executed on these batches (padding statistics in input_padding), '
                       'so achieved_tflops is a dense-EQUIVALENT rate and may exceed the MFMA peak; the hardware rate is '
                       'achieved_tflops_executed')
        e2e['flops_per_impression_executed'] = round(fexec)
        e2e['achieved_tflops_executed'] = round(value * fexec / 1e12, 2)
        e2e['frac_of_f32_mfma_peak_executed'] = round(value * fexec / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4)
        out['input_padding'] = pad_stats
    if sustained:
        out['sustained'] = sustained
    if h2d:
        out.update(h2d)
    if full and world == 1 and not args.no_cpu_baseline and name != 'cfg5' and (not train or cfg.dropout_rate == 0):
        out['cpu_baseline'] = cpu_baseline(run, logits, first_loss)
    return out


def brief(res):
    """The part of a workload's result kept under "also"."""
    r = res['roofline'] or {}
    return {'metric': res['metric'], 'value': res['value'], 'unit': res['unit'], 'steps': res['steps'], 'warmup': res['warmup'],
            'ms_per_step': res['ms_per_step'], 'dtype': res['dtype'], 'workload': res['config']['workload'],
            'dense': res.get('dense'),
            'end_to_end_tflops': res['end_to_end']['achieved_tflops'],
            'dominant_kernel': {k: r.get(k) for k in ('kernel', 'achieved', 'peak', 'frac', 'frac_padded', 'avg_launch_us', 'launches')}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='cfg2b', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='skip the other BASELINE configurations beside the default line')
    ap.add_argument('--plain', action='store_true',
                    help='only the warm-up and the K timed steps of the default path (for rocprofv3 --pmc / --kernel-trace passes: no dense '
                         'run, no sustained / H2D regions, no instrumented pass, no CPU baseline)')
    ap.add_argument('--overlap-streams', action='store_true',
                    help='fork the title / body / freshness / attention-weight branches onto side streams in the timed region')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # a plain `python bench.py --gpus N`: become the launcher.  Nothing above imported torch or touched the GPU.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (the product path has no CPU fallback)'
    ndev = torch.cuda.device_count()
    shared = world > ndev                                 # a rehearsal of N ranks on fewer devices (1-GPU box)
    torch.cuda.set_device(local_rank % max(1, ndev))
    from lime_cikm25_amd import distributed as D
    dist = None
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  RCCL refuses two ranks on one device, so a rehearsal on a 1-GPU box falls back to gloo
        # (and says so in config.backend); a real N-GPU node always takes nccl.
        backend = os.environ.get('LIME_BENCH_BACKEND', 'gloo' if shared else 'nccl')
        D.init(backend=backend, device_id=torch.device('cuda', local_rank % ndev) if backend == 'nccl' else None)

    if args.plain:
        run = Run(args.workload, rank, world)
        sync = torch.cuda.synchronize
        dt, _ = timed(run, args.steps, args.warmup, sync)
        print(json.dumps({'workload': args.workload, 'steps': args.steps, 'ms_per_step': round(dt / args.steps * 1e3, 4),
                          'value': round(run.B * args.steps / dt, 2)}), flush=True)
        return
    res = bench_workload(args.workload, args.steps, args.warmup, rank, world, dist, D, args, full=True)
    if rank == 0 and world == 1 and args.workload == 'cfg2b' and not args.no_also:
        also = {}
        for name, steps, warmup in ALSO:
            try:
                torch.cuda.empty_cache()
                also[name] = brief(bench_workload(name, steps, warmup, 0, 1, None, D, args, full=False))
            except Exception as e:                         # the headline line must survive a failing side workload
                also[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        res['also'] = also
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
