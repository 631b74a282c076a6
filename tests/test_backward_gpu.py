"""Backward / optimizer kernels of the training step (include/lime_hip.h, "Training step") on the MI355X, each against
torch's own fp32 / fp64 CPU autograd of the same op on seeded inputs.  Tolerance: 1e-3 relative (the north star's), with the
tighter bound exact-fp32 kernels are expected to reach asserted alongside."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu

TIGHT = 5e-5


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from lime_cikm25_amd import ops as _ops
    from lime_cikm25_amd import _lib
    _lib.load()
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, want, tol=TIGHT, what=''):
    got = got.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what
    e = rel_err(got.numpy(), want.detach().numpy())
    assert e <= tol, '%s: rel err %.3e > %.1e' % (what, e, tol)
    return e


@pytest.mark.parametrize('split', [True, False], ids=['split_product', 'fp32_mfma'])
@pytest.mark.parametrize('M,N,K', [(1000, 900, 300), (5000, 300, 512), (333, 50, 100), (70, 400, 1800), (4099, 512, 300),
                                   (20000, 960, 300), (4127, 300, 300), (8200, 1024, 640), (6000, 64, 320), (4096, 300, 304)])
def test_wgrad(ops, M, N, K, split):
    """From 4096 rows on the default kernel is the split-product one (csrc/wgrad_sp_f32.hip; (5000, 300, 512) runs transposed,
    (4127, 300, 300) has a 44-row tile tail and a ragged last chunk, (8200, 1024, 640) several tiles both ways, (4096, 300, 304)
    the ones column in its own 16-column tile); `split=False`: the fp32-MFMA kernels for every shape."""
    prev = ops.set_split_gemm(split)
    try:
        _wgrad_case(ops, M, N, K)
    finally:
        ops.set_split_gemm(prev)


def test_wgrad_small_magnitudes(ops):
    """Gradient-sized operands (1e-7 .. 1e-3, mixed signs) through the split product: bf16 has fp32's exponent range."""
    M, N, K = 6000, 300, 300
    g = torch.Generator().manual_seed(5)
    dy = rnd(M, N, seed=1) * torch.pow(10.0, -3 - 4 * torch.rand(M, 1, generator=g))
    x = rnd(M, K, seed=2)
    want = (dy.double().t() @ x.double()).float()
    got, got_b = ops.linear_wgrad(dy.cuda(), x.cuda(), want_bias=True)
    close(got, want, what='wgrad small magnitudes')
    close(got_b, dy.double().sum(0).float(), what='bias gradient small magnitudes')


def _wgrad_case(ops, M, N, K):
    dy, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
    want = (dy.double().t() @ x.double()).float()
    got, got_b = ops.linear_wgrad(dy.cuda(), x.cuda(), want_bias=True)
    close(got, want, what='wgrad %s' % ((M, N, K),))
    close(got_b, dy.double().sum(0).float(), what='bias gradient %s' % ((M, N, K),))
    # accumulate into an existing gradient, strided operands
    wide_dy, wide_x = rnd(M, N + 8, seed=3).cuda(), rnd(M, K + 4, seed=4).cuda()
    base = rnd(N, K, seed=5)
    out = base.clone().cuda()
    base_b = rnd(N, seed=6)
    out_b = base_b.clone().cuda()
    ops.linear_wgrad(wide_dy[:, 4:N + 4], wide_x[:, :K], out=out, accumulate=True, bias_out=out_b)
    want = base + (wide_dy[:, 4:N + 4].cpu().double().t() @ wide_x[:, :K].cpu().double()).float()
    close(out, want, what='wgrad accumulate')
    close(out_b, base_b + wide_dy[:, 4:N + 4].cpu().double().sum(0).float(), what='bias accumulate')


@pytest.mark.parametrize('M,N', [(1, 7), (1000, 300), (70000, 960)])
def test_colsum(ops, M, N):
    x = rnd(M, N, seed=6)
    close(ops.colsum(x.cuda()), x.double().sum(0).float(), what='colsum')


@pytest.mark.parametrize('M,E,div', [(96, 300, 1), (5000, 300, 1), (4096 * 2, 300, 32), (70016, 300, 128), (300, 400, 1), (64, 50, 16)])
def test_layernorm_bwd_and_rstd(ops, M, E, div):
    K = 64
    a, w, b = rnd(M, K, seed=1), rnd(E, K, seed=2, scale=0.3), rnd(E, seed=3)
    res = rnd(M, E, seed=4)
    gamma, beta = rnd(E, seed=5) * 0.5 + 1.0, rnd(E, seed=6)
    dy = rnd((M + div - 1) // div, E, seed=7)
    z = (a @ w.t() + b + res).double().requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    y = F.layer_norm(z, (E,), gd, bd, 1e-5)
    dy_full = dy.double().repeat_interleave(div, dim=0)[:M] / div
    y.backward(dy_full)
    rstd_want = (1.0 / torch.sqrt(z.detach().var(dim=1, unbiased=False) + 1e-5)).float()

    if E <= 320:          # the fused LayerNorm epilogue of lime_linear_f32 hands out rstd
        rstd = torch.empty(M, dtype=torch.float32, device='cuda')
        yg = ops.linear(a.cuda(), w.cuda(), b.cuda(), res=res.cuda(), ln=(gamma.cuda(), beta.cuda()), ln_rstd=rstd)
        close(yg, y.detach().float(), what='fused LN forward')
        close(rstd, rstd_want, what='ln_rstd')
    else:
        yg, rstd = y.detach().float().cuda(), rstd_want.cuda()
    dz, dg, db, dzs = ops.layernorm_bwd(dy.cuda(), yg, gamma.cuda(), beta.cuda(), rstd, dy_div=div, dy_scale=1.0 / div)
    close(dz, z.grad.float(), tol=2e-4, what='dz')
    close(dg, gd.grad.float(), tol=2e-4, what='dgamma')
    close(db, bd.grad.float(), tol=2e-4, what='dbeta')
    close(dzs, z.grad.sum(0).float(), tol=2e-4, what='dzsum')


def test_relu_bwd(ops):
    h = torch.relu(rnd(777, 512, seed=1))
    dh = rnd(777, 512, seed=2)
    got = ops.relu_bwd_(dh.clone().cuda(), h.cuda())
    assert torch.equal(got.cpu(), dh * (h > 0))


@pytest.mark.parametrize('n_seq,S,nh,hd,hs', [(3, 16, 10, 30, 32), (5, 32, 10, 30, 32), (2, 50, 4, 20, 20), (3, 64, 10, 30, 32),
                                              (2, 128, 10, 30, 32), (1, 100, 2, 32, 32), (37, 32, 10, 30, 32),
                                              (2, 512, 10, 30, 32), (3, 200, 3, 30, 32), (1, 129, 2, 20, 20), (2, 256, 4, 32, 32),
                                              (30, 128, 10, 30, 32), (3, 96, 10, 30, 32), (2, 68, 4, 30, 32), (2, 70, 4, 30, 32)])
@pytest.mark.parametrize('split', [True, False], ids=['split_product', 'fp32_mfma'])
def test_token_attention_bwd(ops, n_seq, S, nh, hd, hs, split):
    """split=True (the default setting): the forward for S = 32 / 64 / 128, the whole one-pass backward for 64 < S <= 128 (S % 4 == 0, heads
    32 columns apart: token_attn_bwd_sp_f32.hip) and, for S > 128, the row statistics and the Q K^T / dO V^T products of the blocked
    backward run as split products on the bf16 matrix cores; split=False: every product on the fp32 MFMA."""
    prev = ops.set_split_gemm(split)
    try:
        _attention_bwd_case(ops, n_seq, S, nh, hd, hs)
    finally:
        ops.set_split_gemm(prev)


def _attention_bwd_case(ops, n_seq, S, nh, hd, hs):
    tok = n_seq * S
    scale = 1.0 / math.sqrt(hd)
    W = nh * hs
    qkv = torch.zeros(tok, 3 * W)
    vals = rnd(tok, 3, nh, hd, seed=1)
    qkv.view(tok, 3, nh, hs)[..., :hd] = vals
    dout = rnd(tok, nh * hd, seed=2)
    x = vals.double().requires_grad_()
    q, k, v = (x[:, i].reshape(n_seq, S, nh, hd).permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(tok, nh * hd)
    o.backward(dout.double())
    want = torch.zeros(tok, 3, nh, hs)
    want[..., :hd] = x.grad.float()
    g = qkv.cuda()
    # the forward kernel on the same operands, for completeness of the pair
    out = ops.token_attention(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], n_seq, S, nh, hd, scale, head_stride=hs)
    close(out, o.detach().float(), what='attention forward')
    dqkv = ops.token_attention_bwd(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], dout.cuda(), n_seq, S, nh, hd, scale, head_stride=hs,
                                   out=out)
    close(dqkv, want.view(tok, 3 * W), tol=2e-4, what='dqkv')
    # the pair that hands the softmax statistics from the forward to the backward (what the training path uses for S > 128)
    lse = torch.empty(tok * nh, device='cuda')
    out2 = ops.token_attention(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], n_seq, S, nh, hd, scale, head_stride=hs, lse=lse)
    close(out2, o.detach().float(), what='attention forward (lse variant)')
    want_lse = torch.logsumexp((q @ k.transpose(-1, -2) * scale).detach(), dim=-1) / math.log(2.0)           # [n_seq, nh, S], log2 domain
    close(lse.view(n_seq, S, nh).permute(0, 2, 1), want_lse.float(), tol=1e-5, what='lse')
    dqkv2 = ops.token_attention_bwd(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], dout.cuda(), n_seq, S, nh, hd, scale, head_stride=hs,
                                    out=out2, lse=lse)
    close(dqkv2, want.view(tok, 3 * W), tol=2e-4, what='dqkv from the forward statistics')
    if hs > hd:
        assert (dqkv.view(tok, 3, nh, hs)[..., hd:] == 0).all(), 'pad columns must be exact zeros'


@pytest.mark.parametrize('rows,first', [(20000, 0), (60000, 0), (60000, 7), (9000, 0)])
def test_embed_bwd(ops, rows, first):
    """40 % of the positions carry one id (the padding word in the model; `first` = 7: some other smallest id): above 8192 of them the
    sorted back end sums that run in its own pass (256 slices + an ordered sum), below it in the chunk passes."""
    V, D = 500, 300
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(first + 1, V, (rows,), generator=g, dtype=torch.int32)
    ids[torch.rand(rows, generator=g) < 0.4] = first              # the padding word dominates
    dx = rnd(rows, D, seed=4)
    want = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.long(), dx.double()).float()
    got = ops.embed_bwd(ids.cuda(), dx.cuda(), torch.zeros(V, D, device='cuda'), hot_id=0)
    close(got, want, tol=2e-4, what='embedding gradient')
    got = ops.embed_bwd(ids.cuda(), dx.cuda(), torch.zeros(V, D, device='cuda'), hot_id=-1)
    close(got, want, tol=2e-4, what='embedding gradient, no hot row')


def test_embed_bwd_into_a_table_that_holds_a_gradient(ops):
    """accumulate=True: every back end ADDS to what dtable holds (the sorted back end alone would store over the touched rows)."""
    V, D, rows = 700, 300, 9000
    g = torch.Generator().manual_seed(8)
    ids1 = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    ids2 = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    dx1, dx2 = rnd(rows, D, seed=9), rnd(rows, D, seed=10)
    want = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids1.long(), dx1.double()).index_add_(0, ids2.long(), dx2.double()).float()
    for det in (True, False):
        ops.DETERMINISTIC_EMBED_BWD = det
        try:
            t = ops.embed_bwd(ids1.cuda(), dx1.cuda(), torch.zeros(V, D, device='cuda'), hot_id=0)
            t = ops.embed_bwd(ids2.cuda(), dx2.cuda(), t, hot_id=0, accumulate=True)
        finally:
            ops.DETERMINISTIC_EMBED_BWD = True
        close(t, want, tol=2e-4, what='two calls into one table (deterministic back end %s)' % det)


def test_embed_bwd_small_table(ops):
    T, D, rows = 10, 500, 1760
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 3, (rows,), generator=g, dtype=torch.int32)          # realistic inputs reach buckets {0, 1, 2} only
    dx = rnd(rows, D, seed=6)
    base = rnd(T, D, seed=7)
    want = base.double().index_add_(0, ids.long(), dx.double()).float()
    got = ops.embed_bwd(ids.cuda(), dx.cuda(), base.clone().cuda(), hot_id=-1)
    close(got, want, tol=2e-4, what='small-table gradient')
    again = ops.embed_bwd(ids.cuda(), dx.cuda(), base.clone().cuda(), hot_id=-1)
    assert torch.equal(got, again), 'the small-table path has a fixed summation order'


def test_nll_softmax(ops):
    logits = rnd(37, 5, seed=1, scale=4.0)
    x = logits.double().requires_grad_()
    loss = (-torch.log_softmax(x, dim=1).select(dim=1, index=0)).mean()
    loss.backward()
    got_loss, got_d = ops.nll_softmax(logits.cuda())
    close(got_loss, loss.detach().float().reshape(1), what='loss')
    close(got_d, x.grad.float(), what='dlogits')


def test_clip_and_adam_follow_torch(ops):
    n = 100003
    p0, grads = rnd(n, seed=1), [rnd(n, seed=10 + i, scale=0.05 * (i + 1)) for i in range(4)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.0)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step, g in enumerate(grads, 1):
        ref.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_([ref], 4.0)
        opt.step()
        gg = g.cuda()
        coef = ops.grad_clip_coef(gg, 4.0)
        close(coef[:1], norm.reshape(1), what='grad norm')
        ops.adam_step_(p, gg, m, v, step, 1e-3, grad_scale=coef[1:])
        close(p, ref.detach(), tol=1e-5, what='parameters after step %d' % step)


# ---- backward of the fused tail kernels (csrc/tail_backward_f32.hip) against torch fp64 autograd of the same stage -----------
@pytest.mark.parametrize('M,k,D,A', [(7, 3, 400, 400), (300, 3, 400, 400), (5, 2, 64, 48)])
def test_intent_fuse_bwd(ops, M, k, D, A):
    iv, hid = rnd(2 * M * k, D, seed=1), torch.tanh(rnd(2 * M * k, A, seed=2))
    a2t, a2b = rnd(A, seed=3, scale=0.2), rnd(A, seed=4, scale=0.2)
    dcontent = rnd(M, 2 * D, seed=5)
    x, h, wt, wb = (t.double().requires_grad_() for t in (iv, hid, a2t, a2b))

    def pool(xx, hh, w):
        alpha = torch.softmax((hh.view(M, k, A) * w).sum(-1), dim=1)
        return (alpha.unsqueeze(-1) * xx.view(M, k, D)).sum(1)
    t, b = pool(x[:M * k], h[:M * k], wt), pool(x[M * k:], h[M * k:], wb)
    sim = (F.cosine_similarity(t, b, dim=1) + 1) / 2
    content = torch.cat([t, sim.unsqueeze(1) * b], dim=1)
    content.backward(dcontent.double())
    out = torch.empty(M, 2 * D, device='cuda')
    ops.intent_fuse(iv.cuda(), hid.cuda(), a2t.cuda(), a2b.cuda(), out, M, k, D, A)
    close(out, content.detach().float(), what='intent_fuse forward')
    d_int, d_hid, da_t, da_b = ops.intent_fuse_bwd(iv.cuda(), hid.cuda(), a2t.cuda(), a2b.cuda(), dcontent.cuda(), M, k, D, A)
    close(d_int, x.grad.float(), tol=2e-4, what='d intents')
    close(d_hid, h.grad.float(), tol=2e-4, what='d hidden')
    close(da_t, wt.grad.float(), tol=2e-4, what='d affine2 title')
    close(da_b, wb.grad.float(), tol=2e-4, what='d affine2 body')


@pytest.mark.parametrize('rows,D', [(13, 400), (1600, 400), (9, 70)])
def test_gate_ln_bwd(ops, rows, D):
    y, x = rnd(rows, D, seed=1), rnd(rows, D, seed=2)
    s = torch.softmax(rnd(rows, seed=3), dim=0) * 5
    bias, gamma, beta = rnd(D, seed=4, scale=0.3), rnd(D, seed=5, scale=0.3) + 1, rnd(D, seed=6, scale=0.3)
    dout = rnd(rows, D, seed=7)
    yd, xd, sd, bd, gd, ed = (t.double().requires_grad_() for t in (y, x, s, bias, gamma, beta))
    g = torch.sigmoid(sd.unsqueeze(1) * yd + bd)
    wc = sd.unsqueeze(1) * xd
    out = F.layer_norm(g * wc + (1 - g) * xd, (D,), gd, ed, 1e-5)
    out.backward(dout.double())
    c = lambda t: t.cuda()
    got = ops.gate_ln_bwd(c(y), c(x), c(s), c(bias), c(gamma), c(beta), 1e-5, c(dout))
    for name, a, b in zip(('dy', 'dx', 'dscale', 'dbias', 'dgamma', 'dbeta'), got, (yd, xd, sd, bd, gd, ed)):
        close(a.view(b.shape), b.grad.float(), tol=2e-4, what=name)


@pytest.mark.parametrize('B,N,H,A,D,penalty', [(3, 5, 50, 400, 400, True), (32, 5, 50, 400, 400, True), (2, 1, 7, 64, 48, False)])
def test_interest_match_bwd(ops, B, N, H, A, D, penalty):
    kp, qp = rnd(B * H, A, seed=1, scale=0.3), rnd(B * N, A, seed=2, scale=0.3)
    g, cand = rnd(B * H, D, seed=3), rnd(B * N, D, seed=4)
    remaining = rnd(B, N, seed=5, scale=8.0)
    dlogits = rnd(B, N, seed=6)
    alpha, beta, scale = 0.3, 0.3, 1.0 / math.sqrt(A)
    kd, qd, gd, cd = (t.double().requires_grad_() for t in (kp, qp, g, cand))
    a = torch.einsum('bha,bna->bnh', kd.view(B, H, A), qd.view(B, N, A)) * scale
    u = torch.softmax(a, dim=-1) @ gd.view(B, H, D)
    base = (u * cd.view(B, N, D)).sum(-1)
    r = remaining.double()
    if penalty:
        w = torch.sigmoid(alpha * r)
        w = torch.where(r >= 0, w, beta * w)
    else:
        w = torch.sigmoid(alpha * r.abs())
    (base * w).backward(dlogits.double())
    c = lambda t: t.cuda()
    _, logits = ops.interest_match(c(kp).view(-1), c(qp).view(-1), c(g).view(-1), c(cand).view(-1), c(remaining), B, N, H, A, D, scale,
                                   alpha, beta, True, penalty, want_user=False)
    close(logits, (base * w).detach().float(), what='interest_match forward')
    got = ops.interest_match_bwd(c(kp).view(-1), c(qp).view(-1), c(g).view(-1), c(cand).view(-1), c(remaining), c(dlogits), B, N, H, A, D,
                                 scale, alpha, beta, True, penalty)
    for name, x, ref in zip(('dkp', 'dqp', 'dg', 'dcand'), got, (kd, qd, gd, cd)):
        close(x, ref.grad.float(), tol=2e-4, what=name)


@pytest.mark.parametrize('B,N,H,nh,hd,p', [(4, 5, 50, 10, 40, 0.0), (4, 5, 50, 10, 40, 0.2), (3, 1, 7, 2, 8, 0.5), (2, 3, 70, 10, 40, 0.2)])
def test_cand_attn_weights_train_and_bwd(ops, B, N, H, nh, hd, p):
    """layers.py:66-81 with explicit dropout masks read back from the kernels' generator, against torch fp64 autograd."""
    D = nh * hd
    seed = 4242
    qp, kp = rnd(B * N, D, seed=1, scale=2.0), rnd(B * H, D, seed=2, scale=2.0)
    g = torch.Generator().manual_seed(9)
    mask = torch.rand(B, H, generator=g) < 0.7
    mask[0] = False                                                  # an impression with an empty history
    dagg = rnd(B, H, seed=3)
    m = ops.dropout(torch.ones(B * nh * N, H, device='cuda'), p, seed, 0).cpu().double().view(B, nh, N, H)
    qd, kd = qp.double().requires_grad_(), kp.double().requires_grad_()
    Q = qd.view(B, N, nh, hd).transpose(1, 2)
    K = kd.view(B, H, nh, hd).transpose(1, 2)
    sc = (Q @ K.transpose(-2, -1)) / (D ** 0.5)
    sc = sc.masked_fill(mask.view(B, 1, 1, H) == 0, -1e9)
    a = torch.softmax(sc, dim=-1) * m
    qw = torch.softmax(torch.norm(Q.transpose(1, 2).reshape(B, N, -1), dim=-1), dim=1)
    agg = torch.softmax((a.sum(dim=1) * qw.unsqueeze(-1)).sum(dim=1), dim=-1)
    agg.backward(dagg.double())
    got = ops.cand_attn_weights_train(qp.cuda().view(-1), kp.cuda().view(-1), mask.cuda(), B, N, H, D, nh, p, seed, 0)
    close(got, agg.detach().float(), what='agg')
    if p == 0.0:
        assert rel_err(ops.cand_attn_weights(qp.cuda().view(-1), kp.cuda().view(-1), mask.cuda(), B, N, H, D, nh).cpu().numpy(),
                       got.cpu().numpy()) < 1e-5                        # the scoring kernel computes the same thing
    dqp, dkp = ops.cand_attn_weights_bwd(qp.cuda().view(-1), kp.cuda().view(-1), mask.cuda(), dagg.cuda(), B, N, H, D, nh, p, seed, 0)
    close(dqp, qd.grad.float(), tol=2e-4, what='dqp')
    close(dkp, kd.grad.float(), tol=2e-4, what='dkp')


@pytest.mark.parametrize('n_seq,S,nh,hd', [(6, 32, 10, 20), (3, 16, 10, 20), (2, 100, 4, 32)])
def test_masked_token_attention_bwd(ops, n_seq, S, nh, hd):
    """layers.MultiHeadAttention's attention core (key mask filled with -1e9, layers.py:227-237) forward and backward."""
    tok, W = n_seq * S, nh * hd
    scale = 1.0 / math.sqrt(hd)
    vals = rnd(tok, 3, nh, hd, seed=1)
    dout = rnd(tok, W, seed=2)
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(1, S + 1, (n_seq,), generator=g)
    lens[0] = 0                                                      # a sequence with every key masked
    mask = torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)
    x = vals.double().requires_grad_()
    q, k, v = (x[:, i].reshape(n_seq, S, nh, hd).permute(0, 2, 1, 3) for i in range(3))
    a = (q @ k.transpose(-1, -2) * scale).masked_fill(mask.view(n_seq, 1, 1, S) == 0, -1e9)
    o = (torch.softmax(a, dim=-1) @ v).permute(0, 2, 1, 3).reshape(tok, W)
    o.backward(dout.double())
    qkv = vals.reshape(tok, 3 * W).cuda()
    out = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, nh, hd, scale, key_mask=mask.cuda())
    close(out, o.detach().float(), what='masked attention forward')
    dqkv = ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout.cuda(), n_seq, S, nh, hd, scale, key_mask=mask.cuda())
    close(dqkv, x.grad.float().reshape(tok, 3 * W), tol=2e-4, what='masked dqkv')


@pytest.mark.parametrize('n_seq,S,A,D', [(37, 32, 200, 400), (5, 100, 64, 50), (1, 1, 200, 400)])
def test_additive_pool_bwd(ops, n_seq, S, A, D):
    """layers.Attention over a sequence's tokens (layers.py:285-300), masked: forward and all three gradients against torch fp64."""
    hidden, a2, x = rnd(n_seq * S, A, seed=1), rnd(A, seed=2), rnd(n_seq * S, D, seed=3)
    g = torch.Generator().manual_seed(4)
    mask = (torch.rand(n_seq, S, generator=g) > 0.3).to(torch.uint8)
    mask[:, 0] = 1
    dout = rnd(n_seq, D, seed=5)
    h64, a64, x64 = hidden.double().requires_grad_(), a2.double().requires_grad_(), x.double().requires_grad_()
    score = (h64 @ a64).view(n_seq, S).masked_fill(mask == 0, -1e9)
    rep = (torch.softmax(score, dim=1).unsqueeze(-1) * x64.view(n_seq, S, D)).sum(dim=1)
    rep.backward(dout.double())
    got = ops.additive_pool(hidden.cuda(), a2.cuda(), x.cuda(), n_seq, S, mask=mask.cuda())
    close(got, rep.detach().float(), what='additive pool')
    dh, da2, dx = ops.additive_pool_bwd(hidden.cuda(), a2.cuda(), x.cuda(), dout.cuda(), n_seq, S, mask=mask.cuda())
    close(dh, h64.grad.float(), tol=2e-4, what='d hidden')
    close(da2, a64.grad.float(), tol=2e-4, what='d affine2')
    close(dx, x64.grad.float(), tol=2e-4, what='d x')
    assert (dh.cpu().view(n_seq, S, A)[mask == 0] == 0).all()          # a masked token's score is a constant


def test_fill_pad_rows(ops):
    """Rows whose id is the padding word take src[r % S]; the others are left alone."""
    S, cols, n_seq = 32, 960, 41
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 5, (n_seq * S,), generator=g, dtype=torch.int32)
    src, dst0 = rnd(S, cols, seed=1), rnd(n_seq * S, cols, seed=2)
    got = ops.fill_pad_rows(ids.cuda(), src.cuda(), dst0.clone().cuda(), S).cpu()
    want = torch.where((ids == 0).unsqueeze(1), src.repeat(n_seq, 1), dst0)
    assert torch.equal(got, want)


def test_training_forward_over_live_tokens_equals_the_dense_one(ops):
    """_TokenEncoder with the live-token lists (in_proj over the non-padding tokens + copied padding rows) against the same node without
    them: pooled output and every gradient within fp32 rounding of each other."""
    from lime_cikm25_amd import training as T
    M, S, E, nh, Fd, V = 400, 32, 300, 10, 512, 400
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1, V, (M, S), generator=g, dtype=torch.int32)
    ids[torch.rand(M, S, generator=g) < 0.6] = 0
    names = ['table', 'in_w', 'in_b', 'out_w', 'out_b', 'l1_w', 'l1_b', 'l2_w', 'l2_b', 'n1_w', 'n1_b', 'n2_w', 'n2_b']
    shapes = [(V, E), (3 * E, E), (3 * E,), (E, E), (E,), (Fd, E), (Fd,), (E, Fd), (E,), (E,), (E,), (E,), (E,)]
    vals = {n: rnd(*sh, seed=20 + i, scale=0.5 if n == 'table' else (1.0 / math.sqrt(sh[-1]) if len(sh) == 2 else 0.1)) + (1.0 if n in ('n1_w', 'n2_w') else 0.0)
            for i, (n, sh) in enumerate(zip(names, shapes))}
    pe, G = rnd(S, E, seed=40).cuda(), rnd(M, E, seed=41).cuda()
    flat = ids.reshape(-1)
    rows = torch.nonzero(flat != 0).reshape(-1).to(torch.int32)
    assert rows.numel() >= 4096                              # (the node itself falls back below 4096 live tokens)
    res = []
    for live in (None, (flat[rows.long()].contiguous().cuda(), rows.cuda())):
        dev = {n: v.clone().cuda().requires_grad_(True) for n, v in vals.items()}
        extra = () if live is None else (live,)
        pooled = T._TokenEncoder.apply(ids.cuda(), nh, 1e-5, 1e-5, 0.0, 0, dev['table'], pe, *[dev[n] for n in names[1:]], *extra)
        (pooled * G).sum().backward()
        res.append((pooled.detach().cpu(), {n: dev[n].grad.cpu() for n in names}))
    close(res[1][0], res[0][0], tol=1e-5, what='pooled')
    for n in names:
        scale = float(res[0][1][n].abs().max())
        assert float((res[1][1][n] - res[0][1][n]).abs().max()) <= 2e-5 * max(scale, 1e-6), n


@pytest.mark.parametrize('n_seq,S,nh', [(3, 68, 4), (2, 96, 10), (30, 128, 10), (2, 200, 3), (1, 512, 4)])
def test_token_attention_bwd_dropout_split_equals_fp32(ops, n_seq, S, nh):
    """Attention-probability dropout in the backward: the split-product kernels (one-pass for 64 < S <= 128 -- phase A hashes four
    consecutive keys per lane, phase B shares four hashes over a quad by DPP -- and the blocked kernel for S > 128) regenerate the
    same mask as the fp32-MFMA kernels: the two builds agree to fp32 rounding on the same (p, seed, site)."""
    hd, hs = 30, 32
    tok, W = n_seq * S, nh * hs
    qkv = torch.zeros(tok, 3 * W)
    qkv.view(tok, 3, nh, hs)[..., :hd] = rnd(tok, 3, nh, hd, seed=5)
    g = qkv.cuda()
    dout = rnd(tok, nh * hd, seed=6).cuda()
    scale = 1.0 / math.sqrt(hd)
    drop = (0.2, 424242, 2)
    res = {}
    for split in (True, False):
        prev = ops.set_split_gemm(split)
        try:
            out = ops.token_attention_dropout(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], n_seq, S, nh, hd, scale, *drop, head_stride=hs)
            res[split] = (out, ops.token_attention_bwd(g[:, :W], g[:, W:2 * W], g[:, 2 * W:], dout, n_seq, S, nh, hd, scale, head_stride=hs,
                                                        out=out, dropout=drop))
        finally:
            ops.set_split_gemm(prev)
    for a, b, what in ((res[True][0], res[False][0], 'forward'), (res[True][1], res[False][1], 'dqkv')):
        err = float((a - b).abs().max()) / float(b.abs().max())
        assert err < 2e-5, '%s: split and fp32 kernels differ by %.2e' % (what, err)
    dropped = float((res[False][1] != 0).float().mean())
    assert dropped > 0.5                                                          # a real gradient, not zeros
