"""Encoding only what differs in a padded batch (csrc/compact.hip, newsEncoders.encode_tokens_compact): the index lists against a
numpy statement, the compacted in_proj GEMM (device-side row count, scattered result rows) and the row-map attention against
the dense kernels on the same data, the compacted token encoder against the dense one, and the model with DEDUP on against
DEDUP off (every token of every slot through the layer, as the reference computes it, newsEncoders.py:311-321) -- including a
batch without any padding and one that is nothing but padding."""
import math

import numpy as np
import pytest
import torch

from helpers import rel_err
from lime_cikm25_amd import Model, make_config, newsEncoders, ops, synth
from oracle import lime_oracle as O
from test_model_gpu import gpu_model, run

pytestmark = pytest.mark.gpu


def random_ids(n_seq, S, seed, p_empty=0.4, vocab=1000):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, S + 1, size=n_seq)
    lens[rng.random(n_seq) < p_empty] = 0
    ids = rng.integers(1, vocab, size=(n_seq, S))
    ids[np.arange(S)[None, :] >= lens[:, None]] = 0
    holes = rng.random((n_seq, S)) < 0.02                       # padding words INSIDE a text are handled too
    ids[holes] = 0
    return ids.astype(np.int32)


def np_compact(ids):
    n_seq, S = ids.shape
    live = (ids != 0).any(axis=1)
    src = np.flatnonzero(live)
    n_live = len(src)
    inv = np.full(n_seq, n_live, dtype=np.int64)
    inv[src] = np.arange(n_live)
    ids_c = np.concatenate([ids[src], np.zeros((1, S), np.int32)])
    rows = np.arange((n_live + 1) * S).reshape(n_live + 1, S)
    cap = (n_seq + 1) * S
    row_map = np.where(ids_c != 0, rows, cap + np.arange(S)[None, :])
    tok = np.flatnonzero(ids_c.reshape(-1) != 0)
    return inv, ids_c, row_map, ids_c.reshape(-1)[tok], tok, n_live


@pytest.mark.parametrize('n_seq,S,p_empty', [(300, 32, 0.4), (77, 128, 0.5), (40, 512, 0.3), (64, 32, 0.0), (50, 64, 1.0)])
def test_compact_sequences_against_numpy(n_seq, S, p_empty):
    ids = random_ids(n_seq, S, seed=n_seq + S, p_empty=p_empty)
    c = ops.compact_sequences(torch.from_numpy(ids).cuda())
    inv, ids_c, row_map, tok_ids, tok_rows, n_live = np_compact(ids)
    counts = c.counts.cpu().numpy()
    assert counts.tolist() == [n_live + 1, (n_live + 1) * S, len(tok_ids), n_live, len(tok_ids) + S]
    # behind the live tokens: the S padding rows themselves (id 0 -> row pad_base + t)
    assert (c.tok_ids.cpu().numpy()[len(tok_ids):len(tok_ids) + S] == 0).all()
    assert np.array_equal(c.tok_rows.cpu().numpy()[len(tok_ids):len(tok_ids) + S], c.cap + np.arange(S))
    assert np.array_equal(c.seq_inv.cpu().numpy(), inv)
    n = (n_live + 1) * S
    assert np.array_equal(c.ids_c.cpu().numpy()[:n], ids_c.reshape(-1))
    assert np.array_equal(c.row_map.cpu().numpy()[:n], row_map.reshape(-1))
    assert np.array_equal(c.tok_ids.cpu().numpy()[:len(tok_ids)], tok_ids)
    assert np.array_equal(c.tok_rows.cpu().numpy()[:len(tok_ids)], tok_rows)


def test_compacted_inproj_and_row_map_attention_equal_the_dense_kernels():
    """in_proj over the live tokens (m_dev, c_ids) writes exactly the rows the dense in_proj writes for them (same kernel, same
    k order: bitwise), and attention through row_map equals attention over the materialised rows (bitwise)."""
    n_seq, S, E, nhead = 200, 32, 300, 10
    hd, Wd = E // nhead, nhead * 32
    prev = ops.set_split_gemm(True, force=True)     # both launches on ONE kernel family (the dense one has too few tiles for the default rules)
    try:
        _compacted_inproj_body(n_seq, S, E, nhead, hd, Wd)
    finally:
        ops.set_split_gemm(prev)


def _compacted_inproj_body(n_seq, S, E, nhead, hd, Wd):
    g = torch.Generator().manual_seed(3)
    ids_np = random_ids(n_seq, S, seed=5, p_empty=0.0, vocab=700)
    ids = torch.from_numpy(ids_np).cuda()
    table = (torch.randn(700, E, generator=g) * 0.3).cuda()
    w = ops.pad_heads((torch.randn(3 * E, E, generator=g) * 0.05).cuda(), 3 * nhead, hd, 32)
    pew = torch.randn(S, 3 * Wd, generator=g).cuda()
    dense = ops.linear(table, w, None, a_ids=ids.reshape(-1), res=pew, res_mod=S)          # [n_seq * S, 960], big-M kernel
    c = ops.compact_sequences(ids)
    cap = c.cap
    qkv = torch.full((cap + S, 3 * Wd), float('nan'), device='cuda')
    ops.linear(table, w, None, a_ids=c.tok_ids, res=pew, res_mod=S, out=qkv[:cap], m_dev=c.n_live_tokens, c_ids=c.tok_rows)
    torch.cuda.synchronize()
    live = torch.from_numpy(ids_np.reshape(-1) != 0).cuda()
    assert torch.equal(qkv[:n_seq * S][live], dense[live])                                   # p_empty = 0: compact order = original
    assert torch.isnan(qkv[:n_seq * S][~live]).all() and torch.isnan(qkv[(n_seq + 1) * S:cap]).all()   # nothing else was written
    # padding rows from the dense result of a padding token at each position (any sequence: they only depend on t)
    pad_rows = ops.linear(table, w, None, a_ids=torch.zeros(S, dtype=torch.int32, device='cuda'), res=pew, res_mod=S)
    qkv[cap:] = pad_rows
    full = qkv[:(n_seq + 1) * S].clone()
    rm = c.row_map[:(n_seq + 1) * S].long()
    full = qkv[rm]                                                                           # materialised rows
    want = ops.token_attention(full[:, :Wd], full[:, Wd:2 * Wd], full[:, 2 * Wd:], n_seq + 1, S, nhead, hd, 1.0 / math.sqrt(hd), head_stride=32)
    got = ops.token_attention_rows(qkv[:, :Wd], qkv[:, Wd:2 * Wd], qkv[:, 2 * Wd:], c.row_map, c.n_compact, n_seq + 1, S, nhead, hd,
                                   1.0 / math.sqrt(hd))
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize('S', [32, 128])
def test_compact_token_encoder_equals_the_dense_one(S):
    cfg = make_config(vocabulary_size=3000, max_title_length=S, max_abstract_length=128)
    model, sd = gpu_model(cfg, seed=91)
    enc = model.news_encoder.base_news_encoder
    tr, pos = (enc.title_transformer, enc.title_pos_encoder)
    n_seq = 300
    ids = torch.from_numpy(random_ids(n_seq, S, seed=17, p_empty=0.45, vocab=3000)).cuda()
    table = enc.word_embedding.weight
    dense = torch.empty(n_seq, 300, device='cuda')
    comp = torch.empty(n_seq, 300, device='cuda')
    with torch.no_grad():
        newsEncoders.encode_tokens(ids, table, pos.table(), tr, enc.head_num, pooled_out=dense)
        assert newsEncoders.compact_applicable(ids, table, tr, enc.head_num)
        newsEncoders.encode_tokens_compact(ids, table, pos.table(), tr, enc.head_num, pooled_out=comp)
    torch.cuda.synchronize()
    e = rel_err(comp.cpu().numpy(), dense.cpu().numpy())
    print('S=%d compact vs dense pooled: %.2e' % (S, e))
    assert e < 2e-6
    # all-padding sequences: one vector, bit for bit
    empty = (ids == 0).all(dim=1)
    assert int(empty.sum()) > 10 and bool((comp[empty] == comp[empty][0]).all())


@pytest.mark.parametrize('mode', ['mind_shaped', 'no_padding', 'only_padding'])
def test_model_with_and_without_dedup(mode, monkeypatch):
    """Config-2 shape.  DEDUP off = every token of every slot through the layer (the reference's way); DEDUP on must give the
    same logits (<= 2e-6 relative; the oracle is 1e-3 away from neither)."""
    cfg = make_config(vocabulary_size=50000)
    model, sd = gpu_model(cfg, seed=23)
    batch = synth.make_batch(cfg, 32, 5, seed=24)
    if mode == 'no_padding':
        rng = np.random.default_rng(0)
        for k in ('user_title_text', 'user_content_text', 'news_title_text', 'news_content_text'):
            batch[k] = torch.from_numpy(rng.integers(1, cfg.vocabulary_size, size=tuple(batch[k].shape)).astype(np.int32))
    if mode == 'only_padding':
        for k in ('user_title_text', 'user_content_text', 'news_title_text', 'news_content_text'):
            batch[k] = torch.zeros_like(batch[k])
    monkeypatch.setattr(newsEncoders, 'DEDUP', True)
    got = run(model, batch, False)
    model._graphs.clear()
    monkeypatch.setattr(newsEncoders, 'DEDUP', False)
    dense = run(model, batch, False)
    model._graphs.clear()
    e = rel_err(got.numpy(), dense.numpy())
    print('%s: dedup vs dense %.2e' % (mode, e))
    assert torch.isfinite(got).all() and e < 2e-6
    if mode == 'mind_shaped':
        want = O.model_forward(sd, cfg, batch)
        assert rel_err(got.numpy(), want.numpy()) < 1e-3
        monkeypatch.setattr(newsEncoders, 'DEDUP', True)
        assert torch.equal(run(model, batch, False), got)                      # ordered compaction: bitwise reproducible


def test_training_gradients_with_and_without_dedup(monkeypatch):
    """Training step's forward + backward (dropout off) with the all-padding sequences encoded once -- the representative
    collects the gradients of all its slots -- against the same step with every slot encoded: loss and every parameter
    gradient (fixed-order reductions on both sides; the difference is fp32 summation order)."""
    from lime_cikm25_amd.training import negative_log_softmax
    cfg = make_config(vocabulary_size=4000, max_history_num=20, max_title_length=32, max_abstract_length=64, batch_size=16)
    model, sd = gpu_model(cfg, seed=33)
    model.eval()
    model.training = True
    batch = [v.cuda() for v in synth.make_batch(cfg, 16, 5, seed=34).values()]
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(newsEncoders, 'DEDUP', flag)
        model.zero_grad(set_to_none=True)
        loss = negative_log_softmax(model(*batch))
        loss.backward()
        grads[flag] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    (l1, g1), (l0, g0) = grads[True], grads[False]
    assert abs(l1 - l0) < 1e-6 * max(1.0, abs(l0)) and set(g1) == set(g0) and len(g1) > 40
    worst, worst_name = 0.0, ''
    # a parameter's gradient is judged against its own largest entry -- unless that is rounding noise: the key bias of the
    # candidate-aware attention has gradient exactly zero (a constant added to every key shifts all scores of a query alike, softmax
    # cancels it; observed 1e-11 against 1e-4 .. 1 elsewhere), and the two paths round differently wherever their GEMM kernels differ
    gmax = max(float(b.abs().max()) for b in g0.values())
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        scale = max(float(b.abs().max()), 1e-6 * gmax)
        d = float((a - b).abs().max()) / scale
        if d > worst:
            worst, worst_name = d, k
    # (the two paths put their GEMMs, weight gradients and attention on different kernels -- fp32 MFMA below 4096 rows, the split
    # product above -- and a weight gradient is a sum over 10^4 .. 10^5 tokens: a few 1e-5 of its largest entry)
    print('dedup vs dense gradients: worst max-normalised difference %.2e (%s)' % (worst, worst_name))
    assert worst < 1e-4


def test_bf16_model_with_and_without_dedup(monkeypatch):
    """The bf16 token encoders (BASELINE config 3 arithmetic) on the compacted batch against the same model with every slot
    encoded: the live rows, the padding rows and the layer behind them come from the same kernels with the same per-row
    arithmetic."""
    cfg = make_config(vocabulary_size=50000, compute_dtype='bf16')
    model, sd = gpu_model(cfg, seed=29)
    batch = synth.make_batch(cfg, 32, 5, seed=30)
    monkeypatch.setattr(newsEncoders, 'DEDUP', True)
    got = run(model, batch, False)
    model._graphs.clear()
    monkeypatch.setattr(newsEncoders, 'DEDUP', False)
    dense = run(model, batch, False)
    model._graphs.clear()
    e = rel_err(got.numpy(), dense.numpy())
    print('bf16 dedup vs dense %.2e' % e)
    assert torch.isfinite(got).all() and e < 1e-5


def test_mhsa_model_with_and_without_dedup(monkeypatch):
    """LIME-MHSA-CROWN (title only, BASELINE configs[1] read literally) at full size: the padding news' title is encoded once.
    This encoder masks its padding tokens, so a sequence repeats the representative only when ids AND mask are the padding
    news' -- an all-zero title under a different mask (rows 3 and 7 below) must still be encoded on its own."""
    cfg = make_config(vocabulary_size=50000, content_encoder='MHSA')
    model, sd = gpu_model(cfg, seed=37)
    batch = synth.make_batch(cfg, 32, 5, seed=38)
    for r in (3, 7):                                       # all-zero ids with an unusual mask
        batch['user_title_text'][r, 0] = 0
        batch['user_title_mask'][r, 0] = True
    batch['user_title_text'][3, 1] = 0
    batch['user_title_mask'][3, 1] = False
    batch['user_title_mask'][3, 1, 0] = False
    batch['user_title_mask'][3, 1, 5] = True
    monkeypatch.setattr(newsEncoders, 'DEDUP', True)
    got = run(model, batch, False)
    model._graphs.clear()
    monkeypatch.setattr(newsEncoders, 'DEDUP', False)
    dense = run(model, batch, False)
    model._graphs.clear()
    e = rel_err(got.numpy(), dense.numpy())
    want = O.model_forward(sd, cfg, batch)
    print('MHSA dedup vs dense %.2e, vs oracle %.2e' % (e, rel_err(got.numpy(), want.numpy())))
    assert torch.isfinite(got).all() and e < 2e-6 and rel_err(got.numpy(), want.numpy()) < 1e-3


@pytest.mark.parametrize('n,T', [(300, 32), (65, 20), (7, 130)])
def test_mhsa_compaction_helpers_against_torch(n, T):
    """lime_mhsa_live_ids / lime_mhsa_compact_mask against the tensor formulation they replace: the -1 sentinel of all-zero sequences
    under a mask that is not the padding news' mask, the clamped compact ids and the key mask in compact order."""
    g = torch.Generator().manual_seed(n + T)
    ids = torch.randint(1, 50, (n, T), generator=g, dtype=torch.int32)
    mask = torch.rand(n, T, generator=g) < 0.7
    kind = torch.randint(0, 4, (n,), generator=g)
    e0 = torch.arange(T) == 0
    for s in range(n):
        if kind[s] == 0:                    # the padding news: zero ids, first position set
            ids[s] = 0
            mask[s] = e0
        elif kind[s] == 1:                  # zero ids under another mask: must stay live
            ids[s] = 0
            mask[s, 0] = False if T > 1 else True
            mask[s, -1] = True
    ids_d, mask_d = ids.cuda(), mask.cuda()
    odd = (ids == 0).all(dim=1) & ~(mask == e0).all(dim=1)
    want_eff = ids.clone()
    want_eff[:, 0] = torch.where(odd, torch.full_like(ids[:, 0], -1), ids[:, 0])
    eff = ops.mhsa_live_ids(ids_d, mask_d)
    assert torch.equal(eff.cpu(), want_eff)
    c = ops.compact_sequences(eff)
    want_ids_c = c.ids_c.clone().clamp(min=0)
    src = c.seq_src.long().cpu()
    want_mask = torch.where((src < 0).unsqueeze(1), e0.unsqueeze(0), mask[src.clamp(min=0)])
    mask_c = ops.mhsa_compact_mask(c, mask_d)
    n_c = int(c.counts[0])
    assert torch.equal(c.ids_c.cpu()[:n_c * T], want_ids_c.cpu()[:n_c * T]) and int(c.ids_c.min()) >= 0
    assert torch.equal(mask_c.cpu().bool(), want_mask)


def _with_history_fill(cfg, batch, keep):
    """A copy of `batch` in which impression b keeps the first keep(b) of its history slots live and pads the rest with the all-zero
    <PAD> news (mask bit 0 set, corpus.py:476-477) -- or, where keep(b) exceeds its live slots, fills them with fresh texts."""
    rng = np.random.default_rng(7)
    b2 = {k: v.clone() for k, v in batch.items()}
    B, H = b2['user_history_mask'].shape
    for b in range(B):
        n = keep(b)
        for key, S in (('user_title_text', cfg.max_title_length), ('user_content_text', cfg.max_abstract_length)):
            t = b2[key][b]
            t[n:] = 0
            for h in range(n):
                if not bool((t[h] != 0).any()):
                    ln = int(rng.integers(3, S + 1))
                    t[h, :ln] = torch.from_numpy(rng.integers(1, cfg.vocabulary_size, size=ln).astype(np.int32))
        m = b2['user_title_mask'][b]
        m[n:] = False
        m[n:, 0] = True
        m[:n] = b2['user_title_text'][b][:n] != 0
        b2['user_history_mask'][b, :n] = True
        b2['user_history_mask'][b, n:] = False
    return b2


@pytest.mark.parametrize('kind', ['fp32', 'bf16', 'mhsa'])
def test_one_captured_graph_follows_the_live_counts(kind, monkeypatch):
    """The design claim behind the compacted path: ONE captured HIP graph stays valid when a batch has more or fewer live sequences /
    live tokens than the batch it was captured on, because every row count lives in device memory (m_dev, n_seq_dev, counts[]) and
    the buffers have their full-batch size.  Captured on batch A; replayed on B (every history slot live: many more rows than at
    capture), C (almost nothing but padding: rows beyond the counts still hold A's and B's data) and A again; every replay against
    the same batch through the eager, dense path (use_graph off, DEDUP off)."""
    over = dict(vocabulary_size=50000)
    if kind == 'bf16':
        over['compute_dtype'] = 'bf16'
    if kind == 'mhsa':
        over['content_encoder'] = 'MHSA'
    cfg = make_config(**over)
    model, sd = gpu_model(cfg, seed=61)
    A = synth.make_batch(cfg, 32, 5, seed=62)
    H = cfg.max_history_num
    batches = {'A': A, 'B': _with_history_fill(cfg, A, lambda b: H), 'C': _with_history_fill(cfg, A, lambda b: 1 if b % 8 == 0 else 0),
               'D': _with_history_fill(cfg, synth.make_batch(cfg, 32, 5, seed=63), lambda b: (7 * b) % (H + 1))}
    monkeypatch.setattr(newsEncoders, 'DEDUP', True)
    model.use_graph = True
    model._graphs.clear()
    got = {}
    for name in ('A', 'B', 'C', 'D', 'A'):                  # the first call captures; everything behind it replays
        got.setdefault(name, []).append(run(model, batches[name], False))
    assert len(model._graphs) == 1, 'the replays must have gone through ONE captured graph'
    assert torch.equal(got['A'][0], got['A'][1]), 'batch A replayed after B, C and D differs from its capture run'
    monkeypatch.setattr(newsEncoders, 'DEDUP', False)
    model.use_graph = False
    tol = 1e-5 if kind != 'bf16' else 2e-5                  # same kernels per row on both sides (bf16: dedup vs dense, see the test above)
    for name in ('A', 'B', 'C', 'D'):
        want = run(model, batches[name], False)
        e = rel_err(got[name][0].numpy(), want.numpy())
        print('%s batch %s: graph replay (compacted) vs eager dense %.2e' % (kind, name, e))
        assert torch.isfinite(got[name][0]).all() and e < tol, (kind, name, e)
    model.use_graph = True


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_two_encoder_layers(dtype, monkeypatch):
    """num_layers = 2 (config.py:70 allows 1 or 2; newsEncoders.py:244-247) at a size that takes the big kernels: the compacted path
    (layer 0 over the live tokens, layer 1 over the compact rows through the identity row map) against the dense path and the oracle."""
    over = dict(vocabulary_size=20000, num_layers=2, batch_size=8)
    if dtype == 'bf16':
        over['compute_dtype'] = 'bf16'
    cfg = make_config(**over)
    model, sd = gpu_model(cfg, seed=71)
    assert len(model.news_encoder.base_news_encoder.body_transformer.layers) == 2
    batch = synth.make_batch(cfg, 8, 5, seed=72)
    monkeypatch.setattr(newsEncoders, 'DEDUP', True)
    got = run(model, batch, False)
    model._graphs.clear()
    monkeypatch.setattr(newsEncoders, 'DEDUP', False)
    dense = run(model, batch, False)
    model._graphs.clear()
    want = O.model_forward(sd, cfg, batch)
    e_dense, e_oracle = rel_err(got.numpy(), dense.numpy()), rel_err(got.numpy(), want.numpy())
    print('two layers, %s: dedup vs dense %.2e, vs oracle %.2e' % (dtype, e_dense, e_oracle))
    assert torch.isfinite(got).all()
    if dtype == 'fp32':
        assert e_dense < 2e-6 and e_oracle < 1e-3
    else:
        scale = float(want.abs()[want != 0].mean())
        assert e_dense < 2e-5 and float((got - want).abs().max()) < 2e-2 * scale
