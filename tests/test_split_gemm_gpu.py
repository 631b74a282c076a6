"""lime_linear_f32's split-product kernel (csrc/gemm_sp_f32.hip: fp32 operands split in registers into three bf16 terms, six bf16
MFMAs per product block) on the MI355X: every epilogue the encoder layers use, against an fp64 torch-CPU statement of the same
operation -- and beside the fp32-MFMA kernel (gemm_pp_f32.hip) on the same inputs, whose error it must not exceed by more than
a rounding."""
import math

import pytest
import torch

from helpers import rel_err
from oracle import lime_oracle as O

pytestmark = pytest.mark.gpu
TIGHT = 2e-5


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available()
    from lime_cikm25_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True)
def _restore(ops):
    prev = ops.set_split_gemm(True, force=True)        # every shape below on the split kernel, however few tiles it makes
    yield
    ops.set_split_gemm(prev)


def last_kernel():
    from lime_cikm25_amd import _lib
    return _lib.load().lime_last_linear_kernel().decode()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def dev(t):
    return None if t is None else t.cuda()


def both(ops, fn):
    """fn() under the split kernel and under the fp32-MFMA kernel -> (split result, its kernel name, fp32 result)."""
    ops.set_split_gemm(True, force=True)
    a = fn().cpu()
    ka = last_kernel()
    ops.set_split_gemm(False)
    b = fn().cpu()
    kb = last_kernel()
    ops.set_split_gemm(True, force=True)
    assert ka.startswith('gemm_sp_kernel'), ka
    assert not kb.startswith('gemm_sp_kernel'), kb
    return a, ka, b


def check(got, want64, ref32=None, what=''):
    assert torch.isfinite(got).all(), what
    e = rel_err(got.double().numpy(), want64.numpy())
    assert e < TIGHT, '%s: rel err %.3e' % (what, e)
    if ref32 is not None:                         # not worse than the fp32-MFMA kernel by more than a factor of two (+ an ulp)
        e32 = rel_err(ref32.double().numpy(), want64.numpy())
        assert e < 2.0 * e32 + 2e-7, '%s: split %.3e vs fp32 MFMA %.3e' % (what, e, e32)
    return e


@pytest.mark.parametrize('M,N,K', [(30000, 512, 300), (25001, 300, 300), (9000, 960, 300), (40000, 256, 512), (33000, 320, 64),
                                   (30100, 900, 304), (26000, 576, 100), (70400, 512, 300), (24832, 640, 1000), (50000, 300, 512)])
@pytest.mark.parametrize('act', [None, 'relu'])
def test_split_plain(ops, M, N, K, act):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    want = a.double() @ w.double().t() + b.double()
    if act == 'relu':
        want = torch.relu(want)
    da, dw, db = dev(a), dev(w), dev(b)
    got, name, ref = both(ops, lambda: ops.linear(da, dw, db, act=act))
    check(got, want, ref, what='split linear %s' % ((M, N, K, act),))
    got2, _, ref2 = both(ops, lambda: ops.linear(da, dw, None, act=act))
    check(got2, torch.relu(a.double() @ w.double().t()) if act == 'relu' else a.double() @ w.double().t(), ref2, what='no bias')


def test_split_wide_dynamic_range(ops):
    """Operands spanning 2^-40 .. 2^40 and exact zeros: bf16 terms keep fp32's exponent range."""
    M, N, K = 30000, 320, 128
    g = torch.Generator().manual_seed(5)
    a = rnd(M, K, seed=1) * torch.exp2(torch.randint(-40, 41, (M, 1), generator=g).float())
    w = rnd(N, K, seed=2) * torch.exp2(torch.randint(-40, 41, (N, 1), generator=g).float())
    a[:, ::7] = 0.0
    want = a.double() @ w.double().t()
    da, dw = dev(a), dev(w)
    got, _, ref = both(ops, lambda: ops.linear(da, dw, None))
    # per-element relative error against sum |a w| (rows differ by 2^80 in scale)
    mag = a.double().abs() @ w.double().abs().t()
    assert torch.isfinite(got).all()
    e = float(((got.double() - want).abs() / mag.clamp_min(1e-300)).max())
    e32 = float(((ref.double() - want).abs() / mag.clamp_min(1e-300)).max())
    assert e < 1e-6 and e < 2 * e32 + 2e-7, (e, e32)


@pytest.mark.parametrize('M,N,K', [(25000, 300, 300), (30001, 300, 512), (26000, 320, 64), (28000, 288, 300), (40000, 260, 100)])
def test_split_residual_layernorm(ops, M, N, K):
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    g, be = rnd(N, seed=5) + 1.5, rnd(N, seed=6)
    da, dw, db, dr, dg, dbe = dev(a), dev(w), dev(b), dev(r), dev(g), dev(be)
    z = r.double() + a.double() @ w.double().t() + b.double()
    ln64 = O.layer_norm(z, g.double(), be.double())
    got, name, ref = both(ops, lambda: ops.linear(da, dw, db, res=dr, ln=(dg, dbe)))
    assert name.startswith('gemm_sp_kernel<10, true, false, 1, false, false'), name
    check(got, ln64, ref, what='split res + LN')
    got, _, ref = both(ops, lambda: ops.linear(da, dw, db, res=dr))
    check(got, z, ref, what='split res')
    rstd = torch.empty(M, device='cuda')
    got = ops.linear(da, dw, db, res=dr, ln=(dg, dbe), ln_rstd=rstd)
    assert last_kernel().startswith('gemm_sp_kernel<10, true, false, 1, false, true'), last_kernel()
    check(got.cpu(), ln64, what='split res + LN + rstd')
    var = z.var(dim=1, unbiased=False)
    assert rel_err(rstd.cpu().double().numpy(), (1.0 / torch.sqrt(var + 1e-5)).numpy()) < TIGHT


@pytest.mark.parametrize('M,S,N', [(33024, 32, 960), (9000, 128, 960), (26000, 100, 300), (30000, 128, 512)])
def test_split_gather_and_periodic_residual(ops, M, S, N):
    """in_proj as the encoder issues it: A = table rows by id, the positional term as a periodic [S, N] residual."""
    V, E = 700, 300
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w, b = rnd(N, E, seed=9, scale=0.06), rnd(N, seed=10)
    x = table[ids.long()].double() + pe[torch.arange(M) % S].double()
    dt, dw, dids, db = dev(table), dev(w), dev(ids), dev(b)
    pew = ops.linear(dev(pe), dw, db)
    got, _, ref = both(ops, lambda: ops.linear(dt, dw, None, a_ids=dids, res=pew, res_mod=S))
    check(got, x @ w.double().t() + b.double(), ref, what='split gather A + periodic residual')
    got, _, ref = both(ops, lambda: ops.linear(dt, dw, db, a_ids=dids, act='relu'))
    check(got, torch.relu(table[ids.long()].double() @ w.double().t() + b.double()), ref, what='split gather A')


@pytest.mark.parametrize('n_live', [9000, 20000, 4097, 0, 33024])
def test_split_compacted_rows_and_device_count(ops, n_live):
    """c_ids + m_dev: the first *m_dev A rows are computed, result row r lands at out[c_ids[r]], the periodic residual is indexed by
    c_ids[r] % res_mod; rows of `out` that no live row maps to keep their contents."""
    M, S, N, V, E = 33024, 128, 960, 600, 300
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, V, (M,), generator=g, dtype=torch.int32)
    c_ids = torch.randperm(M + 500, generator=g)[:M].to(torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, N, seed=8)
    w = rnd(N, E, seed=9, scale=0.06)
    dt, dw, dids, dc, dpe = dev(table), dev(w), dev(ids), dev(c_ids), dev(pe)
    m_dev = torch.tensor([n_live], dtype=torch.int32, device='cuda')
    want = torch.full((M + 500, N), 7.0, dtype=torch.float64)
    live = slice(0, min(n_live, M))
    want[c_ids[live].long()] = table[ids[live].long()].double() @ w.double().t() + pe[(c_ids[live] % S).long()].double()

    def run():
        out = torch.full((M + 500, N), 7.0, device='cuda')
        ops.linear(dt, dw, None, a_ids=dids, res=dpe, res_mod=S, out=out, m_dev=m_dev, c_ids=dc)
        return out
    got, name, ref = both(ops, run)
    assert name.startswith('gemm_sp_kernel<10, false, false, 1, false, false, true>'), name
    check(got, want, ref, what='split c_ids / m_dev, %d live' % n_live)
    untouched = torch.ones(M + 500, dtype=torch.bool)
    untouched[c_ids[live].long()] = False
    assert (got[untouched] == 7.0).all()


@pytest.mark.parametrize('M,S', [(33024, 32), (25000, 128), (30000, 7)])
def test_split_gathered_residual_layernorm(ops, M, S):
    """out_proj: residual = table[ids] + pe[t] rebuilt in the accumulators, LayerNorm epilogue."""
    V, E, N = 500, 300, 300
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w, b = rnd(N, E, seed=9, scale=0.06), rnd(N, seed=10)
    x = table[ids.long()].double() + pe[torch.arange(M) % S].double()
    attn = rnd(M, E, seed=11)
    g, be = rnd(N, seed=12) + 1.5, rnd(N, seed=13)
    da, dw, db, dt, dids, dpe, dg, dbe = dev(attn), dev(w), dev(b), dev(table), dev(ids), dev(pe), dev(g), dev(be)
    from lime_cikm25_amd import _lib
    lib = _lib.load()
    run = lambda: ops.linear(da, dw, db, res=dt, res_ids=dids, res_pe=dpe, res_period=S, ln=(dg, dbe))
    lib.lime_set_split_gemm(5)
    run()
    assert last_kernel().startswith('gemm_pp_kernel'), 'without bit 1 this instantiation stays on the fp32 kernel: ' + last_kernel()
    ref = run().cpu()
    lib.lime_set_split_gemm(7)                           # route it to the split kernel (off by default: slower there)
    got = run().cpu()
    name = last_kernel()
    assert name.startswith('gemm_sp_kernel<10, true, false, 2'), name
    check(got, O.layer_norm(x + attn.double() @ w.double().t() + b.double(), g.double(), be.double()), ref, what='split gather residual + LN')
    m_dev = torch.tensor([M - 3000], dtype=torch.int32, device='cuda')
    out = torch.full((M, N), 7.0, device='cuda')
    ops.linear(da, dw, db, res=dt, res_ids=dids, res_pe=dpe, res_period=S, ln=(dg, dbe), out=out, m_dev=m_dev)
    assert last_kernel().startswith('gemm_sp_kernel'), last_kernel()
    lib.lime_set_split_gemm(5)
    assert torch.equal(out[:M - 3000].cpu(), got[:M - 3000]) and (out[M - 3000:] == 7.0).all()


@pytest.mark.parametrize('M,K', [(25600, 512), (33024, 300), (70400, 512)])
def test_split_pool32_epilogue(ops, M, K):
    N = 300
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    g, be = rnd(N, seed=5) + 1.5, rnd(N, seed=6)
    da, dw, db, dr, dg, dbe = dev(a), dev(w), dev(b), dev(r), dev(g), dev(be)
    full = ops.linear(da, dw, db, res=dr, ln=(dg, dbe))
    assert last_kernel().startswith('gemm_sp_kernel'), last_kernel()
    big = torch.full((M // 32, N + 52), 7.0, device='cuda')
    got = ops.linear(da, dw, db, res=dr, ln=(dg, dbe), pool32=True, out=big[:, :N])
    assert last_kernel().startswith('gemm_sp_kernel<10, true, false, 1, true'), last_kernel()
    assert rel_err(got.cpu().numpy(), full.cpu().view(M // 32, 32, N).mean(dim=1).numpy()) < TIGHT
    assert (big[:, N:] == 7).all()
    want = O.layer_norm(r.double() + a.double() @ w.double().t() + b.double(), g.double(), be.double()).view(M // 32, 32, N).mean(dim=1)
    check(got.cpu(), want, what='split pool32 vs fp64')
    m_dev = torch.tensor([M - 6400], dtype=torch.int32, device='cuda')
    out = torch.full((M // 32, N), 7.0, device='cuda')
    ops.linear(da, dw, db, res=dr, ln=(dg, dbe), pool32=True, out=out, m_dev=m_dev)
    assert torch.equal(out[:(M - 6400) // 32], got[:(M - 6400) // 32].contiguous()) and (out[(M - 6400) // 32:] == 7.0).all()


def test_split_strided_views_and_untouched_padding(ops):
    M, N, K = 26000, 300, 300
    big_a, big_w, big_c = rnd(M, K + 40, seed=4), rnd(N, K + 8, seed=5), torch.full((M, N + 100), 7.0)
    a, w = big_a[:, 8:8 + K], big_w[:, 4:4 + K]
    ca, cw, cc = dev(big_a), dev(big_w), dev(big_c)
    ops.linear(ca[:, 8:8 + K], cw[:, 4:4 + K], None, out=cc[:, 60:60 + N])
    assert last_kernel().startswith('gemm_sp_kernel'), last_kernel()
    out = cc.cpu()
    check(out[:, 60:60 + N], a.double() @ w.double().t(), what='split strided')
    assert (out[:, :60] == 7).all() and (out[:, 60 + N:] == 7).all()


def test_split_is_deterministic(ops):
    M, N, K = 40000, 300, 300
    a, w, b, r = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2, scale=0.05)), dev(rnd(N, seed=3)), dev(rnd(M, N, seed=4))
    g, be = dev(rnd(N, seed=5) + 1.5), dev(rnd(N, seed=6))
    first = ops.linear(a, w, b, res=r, ln=(g, be)).clone()
    assert last_kernel().startswith('gemm_sp_kernel'), last_kernel()
    for _ in range(5):
        assert torch.equal(ops.linear(a, w, b, res=r, ln=(g, be)), first)


def test_small_problems_stay_on_the_fp32_kernels(ops):
    a, w = dev(rnd(3000, 300, seed=1)), dev(rnd(300, 300, seed=2))
    ops.linear(a, w, None)
    assert not last_kernel().startswith('gemm_sp_kernel'), last_kernel()


@pytest.mark.parametrize('M,N,K,act', [(14080, 400, 400, 'tanh'), (42240, 400, 400, 'tanh'), (13000, 900, 1800, 'sigmoid'), (20000, 320, 96, 'tanh')])
def test_split_runtime_activation(ops, M, N, K, act):
    """tanh / sigmoid in the epilogue (Attention.affine1, layers.py:288; the gates) for the large batches of BASELINE configs[2] / [4]."""
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    f = torch.tanh if act == 'tanh' else torch.sigmoid
    da, dw, db = dev(a), dev(w), dev(b)
    got, name, ref = both(ops, lambda: ops.linear(da, dw, db, act=act))
    check(got, f(a.double() @ w.double().t() + b.double()), ref, what='split + %s' % act)


@pytest.mark.parametrize('M,N,K', [(14080, 400, 900), (16300, 400, 400), (30000, 300, 512)])
def test_split_gathered_residual_without_layernorm(ops, M, N, K):
    """LIME.project as encode_flat issues it: content half of the GEMM + the freshness half gathered from a [buckets^2, N] table by the
    bucket pair (newsEncoders.py:151-153)."""
    T = 100
    a, w, table = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(T, N, seed=3)
    ids = torch.randint(0, T, (M,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
    da, dw, dt, dids = dev(a), dev(w), dev(table), dev(ids)
    got, name, ref = both(ops, lambda: ops.linear(da, dw, None, res=dt, res_ids=dids))
    assert ', 2, ' in name, name
    check(got, a.double() @ w.double().t() + table[ids.long()].double(), ref, what='split + gathered residual')


@pytest.mark.parametrize('M,H,N,K', [(16000, 50, 400, 400), (81500, 50, 400, 400), (12800, 25, 320, 64)])
def test_split_broadcast_residual(ops, M, H, N, K):
    """SAGEConv's lin_r with lin_l(mean) as a residual row shared by the H history rows of a user row (userEncoders.py:153)."""
    a, w, l = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(M // H, N, seed=3)
    da, dw, dl = dev(a), dev(w), dev(l)
    got, name, ref = both(ops, lambda: ops.linear(da, dw, None, res=dl, res_div=H))
    check(got, a.double() @ w.double().t() + l.double().repeat_interleave(H, dim=0), ref, what='split + broadcast residual')


def test_badly_filling_launches_stay_on_the_other_kernels(ops):
    ops.set_split_gemm(True)                             # the default rules
    a, w = dev(rnd(5280, 400, seed=1)), dev(rnd(400, 400, seed=2))
    ops.linear(a, w, None, act='tanh')
    assert last_kernel().startswith('gemm_mid'), last_kernel()
    a = dev(rnd(14080, 400, seed=3))                     # 110 tiles of 256 x 256 on 256 CUs
    ops.linear(a, w, None)
    assert last_kernel().startswith('gemm_pp'), last_kernel()
    a = dev(rnd(28160, 400, seed=4))                     # 220 tiles: the split kernel
    ops.linear(a, w, None)
    assert last_kernel().startswith('gemm_sp'), last_kernel()


# ---------------------------------------------------------------------------------------------------
# token attention on the split product (csrc/token_attn_sp_f32.hip)
# ---------------------------------------------------------------------------------------------------
def _attn_ref64(qkv, n_seq, S, h, hd, scale):
    E = h * hd
    q, k, v = [t.double().view(n_seq, S, h, hd).transpose(1, 2) for t in qkv.split(E, dim=1)]
    a = (q * scale) @ k.transpose(-2, -1)
    return (torch.softmax(a, dim=-1) @ v).transpose(1, 2).reshape(n_seq * S, E)


@pytest.mark.parametrize('S', [32, 64, 128, 256, 512])
@pytest.mark.parametrize('h,hd,n_seq', [(10, 30, 37), (7, 20, 5), (3, 32, 1), (10, 30, 700)])
def test_token_attention_split_product(ops, S, h, hd, n_seq):
    """The encoder layers' attention (heads padded to 32 columns, no mask) on the bf16 matrix cores: against fp64, and no further
    from it than the fp32-MFMA kernel plus a rounding.  n_seq x h is not a multiple of the pairs per group (partial last group);
    700 sequences make every persistent workgroup walk several groups (the register prefetch).  S = 256 / 512: the key-block
    kernel with its running maximum."""
    if S > 128 and n_seq > 100:
        n_seq = 150                                      # (enough tasks for several per workgroup; 700 x 512 tokens would be 4 GB)
    E, W = h * hd, h * 32
    qkv = rnd(n_seq * S, 3 * E, seed=S + h, scale=2.0)
    padded = ops.pad_heads(dev(qkv.t().contiguous()), 3 * h, hd, 32).t().contiguous()
    scale = 1.0 / math.sqrt(hd)
    run = lambda: ops.token_attention(padded[:, :W], padded[:, W:2 * W], padded[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32)
    ops.set_split_gemm(True)
    a = run().cpu()
    ops.set_split_gemm(False)
    b = run().cpu()
    ops.set_split_gemm(True, force=True)
    want = _attn_ref64(qkv, n_seq, S, h, hd, scale)
    ea, eb = rel_err(a.double().numpy(), want.numpy()), rel_err(b.double().numpy(), want.numpy())
    assert torch.isfinite(a).all()
    assert ea <= TIGHT and ea <= 2.0 * eb + 2e-7, (ea, eb)
    assert not torch.equal(a, b)                        # two different kernels did run


def test_token_attention_split_product_wide_scores(ops):
    """Peaked softmaxes (|score| up to ~60): the three-term split keeps q . k to fp32 accuracy where one bf16 term would not."""
    n_seq, S, h, hd = 9, 128, 10, 30
    E, W = h * hd, h * 32
    qkv = rnd(n_seq * S, 3 * E, seed=5, scale=6.0)
    padded = ops.pad_heads(dev(qkv.t().contiguous()), 3 * h, hd, 32).t().contiguous()
    scale = 1.0 / math.sqrt(hd)
    got = ops.token_attention(padded[:, :W], padded[:, W:2 * W], padded[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32).cpu()
    want = _attn_ref64(qkv, n_seq, S, h, hd, scale)
    assert rel_err(got.double().numpy(), want.numpy()) <= TIGHT


@pytest.mark.parametrize('S', [32, 128, 512])
def test_token_attention_rows_split_product(ops, S):
    """The row-map variant (compacted batches): bit-identical to the dense call on the materialised rows, device-side sequence count."""
    n_seq, h, hd = 23, 10, 30
    W = h * 32
    g = torch.Generator().manual_seed(S)
    n_rows = 400 if S <= 128 else 3000
    qkv = (torch.rand(n_rows, 3 * W, generator=g) * 4 - 2)
    qkv.view(n_rows, 3 * h, 32)[:, :, hd:] = 0
    row_map = torch.randint(0, n_rows, (n_seq * S,), generator=g, dtype=torch.int32)
    d, rm = dev(qkv), dev(row_map)
    full = d[rm.long()]
    scale = 1.0 / math.sqrt(hd)
    want = ops.token_attention(full[:, :W], full[:, W:2 * W], full[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32)
    n_dev = torch.tensor([n_seq - 3], dtype=torch.int32, device='cuda')
    got = ops.token_attention_rows(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], rm, n_dev, n_seq, S, h, hd, scale)
    torch.cuda.synchronize()
    live = (n_seq - 3) * S
    assert torch.equal(got[:live], want[:live])


@pytest.mark.parametrize('M,N,K,scale', [(5000, 512, 300, 1.0), (4096, 512, 300, 1.25), (300, 512, 300, 1.25), (5000, 300, 512, 2.0)])
def test_relu_grad_epilogue(ops, M, N, K, scale):
    """LIME_ACT_RELU_GRAD (dH = (h > 0) ? (dY W) * scale : 0): fused into the split kernel's epilogue for the 256-column tiles
    from 4096 rows on, GEMM + relu_bwd_kernel otherwise -- the same result either way, and equal to the two-step form."""
    dy, w, h = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(M, N, seed=3)
    h = torch.where(h > 0.3, h, torch.zeros_like(h))                     # a ReLU output: exact zeros where it did not pass
    want = torch.where(h > 0, (dy.double() @ w.double().t()) * scale, torch.zeros(M, N, dtype=torch.float64))
    got = ops.linear(dev(dy), dev(w), None, act='relu_grad', res=dev(h), act_scale=scale).cpu()
    k = last_kernel()
    assert (k == 'gemm_sp_kernel<8, false, false, 3, false, false, false>') == (M >= 4096 and N == 512), k
    assert rel_err(got.double().numpy(), want.numpy()) <= TIGHT
    assert torch.equal(got == 0, (h <= 0) | (got == 0))                  # gated entries are exact zeros
    two = ops.relu_bwd_(ops.linear(dev(dy), dev(w), None), dev(h), scale).cpu()
    assert rel_err(got.double().numpy(), two.double().numpy()) <= 2e-6


@pytest.mark.parametrize('M,N,K', [(5000, 512, 300), (300, 512, 300), (4500, 300, 64)])
def test_relu_dropout_epilogue(ops, M, N, K):
    """dropout_p of lime_linear_args: nn.Dropout behind linear1's ReLU (newsEncoders.py:244-247, training mode) in the GEMM's epilogue --
    the same values as the GEMM followed by lime_dropout_f32 with the same (p, seed, site), fused (split kernel, M >= 4096) or not."""
    a, w, b = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2, scale=0.2)), dev(rnd(N, seed=3))
    p, seed, site = 0.2, 987654321, 3
    got = ops.linear(a, w, b, act='relu', dropout=(p, seed, site))
    fused = last_kernel().startswith('gemm_sp_kernel')
    assert fused == (M >= 4096), last_kernel()
    want = ops.dropout(ops.linear(a, w, b, act='relu'), p, seed, site)
    assert torch.equal(got, want)
    frac = float((got == 0).float().mean())
    assert 0.5 < frac < 0.75                              # about half from the ReLU, a fifth of the rest from the mask
