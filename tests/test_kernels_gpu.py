"""Per-kernel parity on the MI355X: every C-ABI entry point against the CPU oracle's functions (or
a plain fp32 torch-CPU statement where the oracle has no finer-grained function), on seeded inputs.
Integer results (buckets) must be bit-exact; fp32 results within 1e-3 relative (observed ~1e-6)."""
import math

import numpy as np
import pytest
import torch

from helpers import rel_err
from oracle import lime_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-3          # the north star's tolerance; the assertions below also print what was observed
TIGHT = 2e-5        # what exact-fp32 kernels are expected to reach against an fp32 CPU evaluation


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from lime_cikm25_amd import ops as _ops
    from lime_cikm25_amd import _lib
    _lib.load()
    return _ops


@pytest.fixture(autouse=True)
def _fp32_mfma_kernels(ops):
    """This module pins the fp32-MFMA GEMM kernels (gemm_pp / gemm_mid / gemm_f32) by name; the split-product kernel that takes the
    big problems by default has its own module (test_split_gemm_gpu.py)."""
    prev = ops.set_split_gemm(False, force=True)       # force: also the problems the fill rules would hand to the 64-row-tile kernel
    yield
    ops.set_split_gemm(prev)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def dev(t):
    return None if t is None else t.cuda()


def check(got, want, tol=TIGHT, what=''):
    got = got.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what
    e = rel_err(got.numpy(), want.numpy())
    assert e < tol, '%s: rel err %.3e' % (what, e)
    return e


# ---------------------------------------------------------------------------------------------------
# lime_linear_f32
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [
    (1, 1, 4), (7, 5, 3), (33, 50, 100), (64, 64, 32), (100, 400, 50), (130, 300, 300), (257, 900, 300), (96, 400, 350),
    (200, 512, 300), (150, 300, 512), (50, 900, 1000), (77, 400, 1800), (4096, 128, 64), (4100, 900, 300), (5000, 70, 31),
])
@pytest.mark.parametrize('act', [None, 'relu', 'tanh', 'sigmoid'])
def test_linear_plain(ops, M, N, K, act):
    if act in ('tanh', 'sigmoid') and M > 300:
        pytest.skip('activation variants covered on the small shapes')
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    want = a @ w.t() + b
    want = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid, None: lambda x: x}[act](want)
    got = ops.linear(dev(a), dev(w), dev(b), act=act)
    check(got, want, what='linear %s' % ((M, N, K, act),))


def test_linear_strided_views_and_no_bias(ops):
    M, N, K = 300, 200, 128
    big_a, big_w, big_c = rnd(M, K + 40, seed=4), rnd(N, K + 8, seed=5), torch.zeros(M, N + 100)
    a, w = big_a[:, 8:8 + K], big_w[:, 4:4 + K]
    want = a @ w.t()
    ca, cw, cc = dev(big_a), dev(big_w), dev(big_c)
    ops.linear(ca[:, 8:8 + K], cw[:, 4:4 + K], None, out=cc[:, 60:60 + N])
    out = cc.cpu()
    check(out[:, 60:60 + N], want, what='strided')
    assert (out[:, :60] == 0).all() and (out[:, 60 + N:] == 0).all()


@pytest.mark.parametrize('M,S', [(40, 8), (64, 32), (4200, 128)])
def test_linear_gather_operand_and_residual(ops, M, S):
    V, E, N = 500, 300, 300
    rows = M
    ids = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w, b = rnd(N, E, seed=9, scale=0.06), rnd(N, seed=10)
    x = table[ids.long()] + pe[torch.arange(rows) % S]
    got = ops.linear(dev(table), dev(w), dev(b), a_ids=dev(ids), a_pe=dev(pe), a_period=S)
    check(got, x @ w.t() + b, what='gather A')
    attn = rnd(rows, E, seed=11)
    g, be = rnd(N, seed=12) + 1.5, rnd(N, seed=13)
    want = O.layer_norm(x + attn @ w.t() + b, g, be)
    got = ops.linear(dev(attn), dev(w), dev(b), res=dev(table), res_ids=dev(ids), res_pe=dev(pe), res_period=S,
                     ln=(dev(g), dev(be)))
    check(got, want, what='gather residual + LN')


@pytest.mark.parametrize('M,N,K', [(70, 300, 512), (129, 200, 400), (5000, 300, 300), (10, 37, 20), (300, 256, 64), (64, 320, 8),
                                   (9000, 300, 512), (257, 33, 6), (100, 300, 31)])
def test_linear_residual_layernorm(ops, M, N, K):
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    g, be = rnd(N, seed=5) + 1.5, rnd(N, seed=6)
    want = O.layer_norm(r + a @ w.t() + b, g, be)
    got = ops.linear(dev(a), dev(w), dev(b), res=dev(r), ln=(dev(g), dev(be)))
    check(got, want, what='res + LN')
    want2 = O.layer_norm(torch.relu(a @ w.t() + b), g, be)
    check(ops.linear(dev(a), dev(w), dev(b), act='relu', ln=(dev(g), dev(be))), want2, what='relu + LN')


def test_linear_broadcast_residual(ops):
    B, H, D = 7, 13, 400
    a, w, l = rnd(B * H, D, seed=1), rnd(D, D, seed=2, scale=0.05), rnd(B, D, seed=3)
    want = a @ w.t() + l.repeat_interleave(H, dim=0)
    check(ops.linear(dev(a), dev(w), None, res=dev(l), res_div=H), want, what='res_div')


def test_linear_gated_residual_layernorm(ops):
    """layers.py:84-89 as one epilogue."""
    R, D = 130, 400
    x, w, b = rnd(R, D, seed=1), rnd(D, D, seed=2, scale=0.05), rnd(D, seed=3)
    s = torch.rand(R, generator=torch.Generator().manual_seed(4))
    g, be = rnd(D, seed=5) + 1.5, rnd(D, seed=6)
    wc = s[:, None] * x
    gate = torch.sigmoid(wc @ w.t() + b)
    want = O.layer_norm(gate * wc + (1 - gate) * x, g, be)
    y = ops.linear(dev(x), dev(w), None)
    got = ops.gate_ln(y, dev(x), dev(s), dev(b), dev(g), dev(be)).view(R, D)
    check(got, want, what='gate')


# ---- the two-workgroups-per-CU LDS-DMA kernel (gemm_pp_f32.hip): M >= 4096, 16-byte friendly operands -----------------
def last_kernel():
    from lime_cikm25_amd import _lib
    return _lib.load().lime_last_linear_kernel().decode()


@pytest.mark.parametrize('M,N,K', [(4096, 512, 300), (4100, 300, 300), (5003, 960, 300), (4097, 256, 512), (6000, 320, 32),
                                   (4300, 900, 304), (4200, 576, 44), (70000, 512, 300), (4128, 640, 1000)])
@pytest.mark.parametrize('act', [None, 'relu'])
def test_linear_pp_plain(ops, M, N, K, act):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    want = a @ w.t() + b
    if act == 'relu':
        want = torch.relu(want)
    got = ops.linear(dev(a), dev(w), dev(b), act=act)
    assert last_kernel().startswith('gemm_pp_kernel'), last_kernel()
    check(got, want, what='pp linear %s' % ((M, N, K, act),))
    got2 = ops.linear(dev(a), dev(w), None, act=act)             # no bias
    check(got2, torch.relu(a @ w.t()) if act == 'relu' else a @ w.t(), what='pp linear, no bias')


@pytest.mark.parametrize('M,N,K,width', [(5000, 400, 400, 208), (4500, 1200, 352, 304), (4096, 160, 400, 208), (6001, 400, 900, 208),
                                         (4200, 860, 64, 304), (4100, 560, 300, 304), (70400, 400, 400, 208)])
def test_linear_pp_trimmed_slabs(ops, M, N, K, width):
    """N whose last 256 / 320-column block would be too narrow runs on 19-tile (304) / 13-tile (208) slabs of the big-M kernel:
    plain, ReLU and dense-residual epilogues, the columns behind N untouched."""
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    trim = {304: ', 1>', 208: ', 3>'}[width]
    for act in (None, 'relu'):
        want = a @ w.t() + b
        got = ops.linear(dev(a), dev(w), dev(b), act=act)
        assert last_kernel().startswith('gemm_pp_kernel') and last_kernel().endswith(trim), last_kernel()
        check(got, torch.relu(want) if act == 'relu' else want, what='pp trimmed %s' % ((M, N, K, act),))
    got = ops.linear(dev(a), dev(w), None, res=dev(r))
    assert last_kernel().startswith('gemm_pp_kernel') and last_kernel().endswith(trim), last_kernel()
    check(got, r + a @ w.t(), what='pp trimmed + residual')
    big = torch.full((M, N + 24), 7.0, device='cuda')
    ops.linear(dev(a), dev(w), dev(b), out=big[:, :N])
    assert (big[:, N:] == 7.0).all()
    check(big[:, :N], a @ w.t() + b, what='pp trimmed, strided out')


@pytest.mark.parametrize('M,N,K', [(4096, 300, 300), (5001, 300, 512), (4500, 320, 64), (4100, 288, 300), (9000, 260, 100)])
def test_linear_pp_residual_layernorm(ops, M, N, K):
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    g, be = rnd(N, seed=5) + 1.5, rnd(N, seed=6)
    got = ops.linear(dev(a), dev(w), dev(b), res=dev(r), ln=(dev(g), dev(be)))
    assert last_kernel().startswith('gemm_pp_kernel<10, true, false, 1, false'), last_kernel()
    check(got, O.layer_norm(r + a @ w.t() + b, g, be), what='pp res + LN')
    got = ops.linear(dev(a), dev(w), dev(b), ln=(dev(g), dev(be)))
    assert last_kernel().startswith('gemm_pp_kernel<10, true, false, 0, false'), last_kernel()
    check(got, O.layer_norm(a @ w.t() + b, g, be), what='pp LN')
    got = ops.linear(dev(a), dev(w), dev(b), res=dev(r))
    assert last_kernel().startswith('gemm_pp_kernel'), last_kernel()
    check(got, r + a @ w.t() + b, what='pp res')


@pytest.mark.parametrize('M,S,N', [(4224, 32, 960), (4200, 128, 960), (4099, 100, 300), (8192, 128, 512)])
def test_linear_pp_gather_and_periodic_residual(ops, M, S, N):
    """in_proj as the encoder issues it: A = table rows by id, the positional term as a periodic [S, N] residual."""
    V, E = 700, 300
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w, b = rnd(N, E, seed=9, scale=0.06), rnd(N, seed=10)
    x = table[ids.long()] + pe[torch.arange(M) % S]
    pew = ops.linear(dev(pe), dev(w), dev(b))
    got = ops.linear(dev(table), dev(w), None, a_ids=dev(ids), res=pew, res_mod=S)
    assert last_kernel().startswith('gemm_pp_kernel'), last_kernel()
    check(got, x @ w.t() + b, what='pp gather A + periodic residual')
    got = ops.linear(dev(table), dev(w), dev(b), a_ids=dev(ids), act='relu')
    check(got, torch.relu(table[ids.long()] @ w.t() + b), what='pp gather A')


@pytest.mark.parametrize('M,S', [(4224, 32), (4200, 128), (5000, 7)])
def test_linear_pp_gathered_residual_layernorm(ops, M, S):
    """out_proj: residual = table[ids] + pe[t] rebuilt in the accumulators, LayerNorm epilogue."""
    V, E, N = 500, 300, 300
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w, b = rnd(N, E, seed=9, scale=0.06), rnd(N, seed=10)
    x = table[ids.long()] + pe[torch.arange(M) % S]
    attn = rnd(M, E, seed=11)
    g, be = rnd(N, seed=12) + 1.5, rnd(N, seed=13)
    got = ops.linear(dev(attn), dev(w), dev(b), res=dev(table), res_ids=dev(ids), res_pe=dev(pe), res_period=S,
                     ln=(dev(g), dev(be)))
    assert last_kernel().startswith('gemm_pp_kernel<10, true, false, 2, false'), last_kernel()
    check(got, O.layer_norm(x + attn @ w.t() + b, g, be), what='pp gather residual + LN')


def test_linear_pp_strided_views_and_untouched_padding(ops):
    M, N, K = 4500, 300, 300
    big_a, big_w, big_c = rnd(M, K + 40, seed=4), rnd(N, K + 8, seed=5), torch.full((M, N + 100), 7.0)
    a, w = big_a[:, 8:8 + K], big_w[:, 4:4 + K]
    ca, cw, cc = dev(big_a), dev(big_w), dev(big_c)
    ops.linear(ca[:, 8:8 + K], cw[:, 4:4 + K], None, out=cc[:, 60:60 + N])
    assert last_kernel().startswith('gemm_pp_kernel'), last_kernel()
    out = cc.cpu()
    check(out[:, 60:60 + N], a @ w.t(), what='pp strided')
    assert (out[:, :60] == 7).all() and (out[:, 60 + N:] == 7).all()


@pytest.mark.parametrize('M,S', [(300, 16), (4100, 32)])
def test_linear_periodic_residual_small_and_misaligned(ops, M, S):
    """res_mod on the general kernel (small M, and K not a multiple of 4)."""
    N, K = 130, 50
    a, w, t = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(S, N, seed=3)
    got = ops.linear(dev(a), dev(w), None, res=dev(t), res_mod=S)
    check(got, a @ w.t() + t[torch.arange(M) % S], what='res_mod')
    assert last_kernel().startswith('gemm_f32_kernel'), last_kernel()


def test_linear_pp_is_deterministic(ops):
    M, N, K = 20000, 300, 300
    a, w, b, r = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2, scale=0.05)), dev(rnd(N, seed=3)), dev(rnd(M, N, seed=4))
    g, be = dev(rnd(N, seed=5) + 1.5), dev(rnd(N, seed=6))
    first = ops.linear(a, w, b, res=r, ln=(g, be)).clone()
    for _ in range(5):
        assert torch.equal(ops.linear(a, w, b, res=r, ln=(g, be)), first)


@pytest.mark.parametrize('M,K', [(4096, 512), (9024, 300), (70400, 512)])
def test_linear_pool32_epilogue(ops, M, K):
    """pool32: the LayerNorm output averaged over 32-row blocks in the epilogue == mean_pool of the full result."""
    N = 300
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
    g, be = rnd(N, seed=5) + 1.5, rnd(N, seed=6)
    full = ops.linear(dev(a), dev(w), dev(b), res=dev(r), ln=(dev(g), dev(be)))
    big = torch.full((M // 32, N + 52), 7.0, device='cuda')
    got = ops.linear(dev(a), dev(w), dev(b), res=dev(r), ln=(dev(g), dev(be)), pool32=True, out=big[:, :N])
    assert last_kernel().startswith('gemm_pp_kernel<10, true, false, 1, false, true'), last_kernel()
    check(got, full.cpu().view(M // 32, 32, N).mean(dim=1), what='pool32')
    assert (big[:, N:] == 7).all()
    check(got, O.layer_norm(r + a @ w.t() + b, g, be).view(M // 32, 32, N).mean(dim=1), what='pool32 vs oracle')
    from lime_cikm25_amd._lib import LimeHipError
    with pytest.raises(LimeHipError):                      # needs the big-M kernel
        ops.linear(dev(a[:64]), dev(w), dev(b), res=dev(r[:64]), ln=(dev(g), dev(be)), pool32=True)
    with pytest.raises(ValueError):
        ops.linear(dev(a[:M - 4]), dev(w), dev(b), res=dev(r[:M - 4]), ln=(dev(g), dev(be)), pool32=True)      # M % 32 != 0


def test_linear_rejects_bad_shapes(ops):
    from lime_cikm25_amd._lib import LimeHipError
    a, w = dev(rnd(8, 16)), dev(rnd(4, 16))
    with pytest.raises(ValueError):
        ops.linear(a, dev(rnd(4, 12)))
    with pytest.raises(LimeHipError):
        ops.linear(dev(rnd(8, 500)), dev(rnd(500, 500)), ln=(dev(rnd(500)), dev(rnd(500))))      # LN needs N <= 320


# ---------------------------------------------------------------------------------------------------
# token attention
# ---------------------------------------------------------------------------------------------------
def attn_ref(qkv, n_seq, S, h, hd, scale, mask=None):
    E = h * hd
    q, k, v = [t.view(n_seq, S, h, hd).transpose(1, 2) for t in qkv.split(E, dim=1)]
    a = (q * scale) @ k.transpose(-2, -1)
    if mask is not None:
        a = a.masked_fill(mask.view(n_seq, 1, 1, S) == 0, -1e9)
    return (torch.softmax(a, dim=-1) @ v).transpose(1, 2).reshape(n_seq * S, E)


@pytest.mark.parametrize('S', [1, 8, 16, 31, 32, 33, 64, 96, 100, 128, 160, 256, 512])
@pytest.mark.parametrize('h,hd', [(10, 30), (10, 20), (3, 32)])
def test_token_attention(ops, S, h, hd):
    n_seq = 5 if S <= 128 else 2
    E = h * hd
    qkv = rnd(n_seq * S, 3 * E, seed=S, scale=2.0)
    scale = 1.0 / math.sqrt(hd)
    d = dev(qkv)
    got = ops.token_attention(d[:, :E], d[:, E:2 * E], d[:, 2 * E:], n_seq, S, h, hd, scale)
    check(got, attn_ref(qkv, n_seq, S, h, hd, scale), what='attn S=%d' % S)


@pytest.mark.parametrize('S', [1, 31, 32, 64, 96, 100, 128, 160, 256])
@pytest.mark.parametrize('h,hd', [(10, 30), (7, 20), (3, 32)])
def test_token_attention_padded_heads(ops, S, h, hd):
    """The model's layout: every head of Q/K/V padded to 32 columns by padding in_proj's rows (pad_heads)."""
    n_seq, hs = (37 if S <= 128 else 3), 32
    E, W = h * hd, h * hs
    qkv = rnd(n_seq * S, 3 * E, seed=S + 7, scale=2.0)
    padded = ops.pad_heads(dev(qkv.t().contiguous()), 3 * h, hd, hs).t().contiguous()      # rows of a [3E, tok] matrix
    assert padded.shape == (n_seq * S, 3 * W)
    ref_pad = torch.zeros(n_seq * S, 3 * h, hs)
    ref_pad[:, :, :hd] = qkv.view(n_seq * S, 3 * h, hd)
    assert torch.equal(padded.cpu(), ref_pad.view(n_seq * S, 3 * W))
    scale = 1.0 / math.sqrt(hd)
    got = ops.token_attention(padded[:, :W], padded[:, W:2 * W], padded[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=hs)
    check(got, attn_ref(qkv, n_seq, S, h, hd, scale), what='padded attn S=%d' % S)


def test_token_attention_padded_heads_garbage_pad(ops):
    """Pad columns hold zeros in the model, but the general (non-FAST) path must not read them at all."""
    n_seq, S, h, hd, hs = 4, 40, 10, 30, 32
    E, W = h * hd, h * hs
    qkv = rnd(n_seq * S, 3 * E, seed=3, scale=2.0)
    pad = torch.full((n_seq * S, 3 * h, hs), float('nan'))
    pad[:, :, :hd] = qkv.view(n_seq * S, 3 * h, hd)
    d = dev(pad.view(n_seq * S, 3 * W))
    scale = 1.0 / math.sqrt(hd)
    got = ops.token_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=hs)
    check(got, attn_ref(qkv, n_seq, S, h, hd, scale), what='padded attn, NaN pads')


def test_pad_heads_bias(ops):
    b = rnd(90, seed=1)
    got = ops.pad_heads(dev(b), 9, 10, 16).cpu()
    ref = torch.zeros(9, 16)
    ref[:, :10] = b.view(9, 10)
    assert torch.equal(got, ref.view(-1))


@pytest.mark.parametrize('S', [16, 32, 48])
def test_token_attention_key_mask(ops, S):
    n_seq, h, hd = 6, 10, 20
    E = h * hd
    qkv = rnd(n_seq * S, 3 * E, seed=S + 1, scale=2.0)
    lens = torch.tensor([1, S, S // 2, 3, S - 1, 0])          # a fully masked row softmaxes to uniform, as in the reference
    mask = torch.arange(S)[None, :] < lens[:, None]
    d = dev(qkv)
    got = ops.token_attention(d[:, :E], d[:, E:2 * E], d[:, 2 * E:], n_seq, S, h, hd, 1 / math.sqrt(hd), key_mask=dev(mask))
    check(got, attn_ref(qkv, n_seq, S, h, hd, 1 / math.sqrt(hd), mask), what='masked attn')


def test_encoder_layer_matches_oracle(ops):
    """The five-launch encoder layer against oracle.encoder_layer (newsEncoders.py:244-247,311-321)."""
    from lime_cikm25_amd import newsEncoders
    M, S, E, V = 9, 32, 300, 700
    tr = torch.nn.TransformerEncoder(torch.nn.TransformerEncoderLayer(E, 10, 512, 0.0, batch_first=True), 1)
    for i, (name, p) in enumerate(tr.named_parameters()):
        p.data = rnd(*p.shape, seed=100 + i, scale=0.08 if p.dim() == 2 else 0.3)
        if 'norm' in name and name.endswith('weight'):
            p.data += 1.2
    ids = torch.randint(0, V, (M, S), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    table, pe = rnd(V, E, seed=2, scale=0.6), O.positional_encoding(S, E)
    sd = {'l.' + k: v for k, v in tr.layers[0].state_dict().items()}
    want = O.encoder_layer(table[ids.long()] + pe, sd, 'l.', 10).reshape(M * S, E)
    got = newsEncoders.encode_tokens(dev(ids), dev(table), dev(pe), tr.cuda(), 10)
    check(got, want, what='encoder layer')
    pooled = ops.mean_pool(got, M, S)
    check(pooled, want.view(M, S, E).mean(dim=1), what='mean pool')


# ---------------------------------------------------------------------------------------------------
# small kernels
# ---------------------------------------------------------------------------------------------------
def test_embed_pe(ops):
    V, E, S, rows = 900, 300, 32, 32 * 41
    ids = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    table, pe = rnd(V, E, seed=2), rnd(S, E, seed=3)
    want = table[ids.long()] + pe[torch.arange(rows) % S]
    got = ops.embed_pe(dev(ids), dev(table), dev(pe), S).cpu()
    assert torch.equal(got, want)                                       # one add: bit-exact
    assert torch.equal(ops.embed_pe(dev(ids), dev(table)).cpu(), table[ids.long()])
    t2 = rnd(V, 50, seed=4)                                             # dim % 4 != 0 -> scalar path
    assert torch.equal(ops.embed_pe(dev(ids), dev(t2)).cpu(), t2[ids.long()])


def test_bucketize_bit_exact(ops):
    cuts = np.array(O.BUCKET_THRESHOLD_BITS, dtype=np.uint32)
    edge = np.concatenate([(cuts - 2), (cuts - 1), cuts, (cuts + 1)]).view(np.float32)
    rng = np.random.default_rng(0)
    x = np.concatenate([edge, np.exp(rng.uniform(-3, np.log(3e38), 200000)).astype(np.float32),
                        np.array([0.0, -1.0, 1.0, 0.99999, np.inf, np.nan, 3.4e38, 86400.0], dtype=np.float32)])
    xt = torch.from_numpy(x)
    got = ops.bucketize(dev(xt)).cpu()
    assert got.dtype == torch.int32
    assert torch.equal(got.long(), O.bucketize(xt))
    assert got[len(edge) + 200000 + 4].item() == 9 and got[len(edge) + 200000 + 5].item() == 0      # +inf, NaN


def test_topic_rep(ops):
    rows, C, SC, dc = 77, 18, 270, 50
    g = torch.Generator().manual_seed(1)
    cat = torch.randint(0, C, (rows,), generator=g, dtype=torch.int32)
    sub = torch.randint(0, SC, (rows,), generator=g, dtype=torch.int32)
    ct, st, w, b = rnd(C, dc, seed=2), rnd(SC, dc, seed=3), rnd(dc, 2 * dc, seed=4, scale=0.2), rnd(dc, seed=5)
    e = torch.cat([ct[cat.long()], st[sub.long()]], dim=1)
    emb = torch.zeros(rows, 130).cuda()
    got = ops.topic_rep(dev(cat), dev(sub), dev(ct), dev(st), dev(w), dev(b), emb_out=emb[:, 20:120])
    check(got, e @ w.t() + b, what='topic rep')
    assert torch.equal(emb.cpu()[:, 20:120], e)


def test_intent_fuse(ops):
    M, k, D, A = 37, 3, 400, 400
    intents, hidden = torch.relu(rnd(2 * M * k, D, seed=1)), torch.tanh(rnd(2 * M * k, A, seed=2, scale=2))
    a2t, a2b = rnd(A, seed=3, scale=0.3), rnd(A, seed=4, scale=0.3)
    iv, hv = intents.view(2, M, k, D), hidden.view(2, M, k, A)
    pooled = []
    for tb, a2 in ((0, a2t), (1, a2b)):
        alpha = torch.softmax(hv[tb] @ a2, dim=1)
        pooled.append((alpha.unsqueeze(-1) * iv[tb]).sum(dim=1))
    s = (torch.nn.functional.cosine_similarity(pooled[0], pooled[1], dim=1) + 1) / 2
    want = torch.cat([pooled[0], s[:, None] * pooled[1]], dim=1)
    content = torch.zeros(M, 900).cuda()
    ops.intent_fuse(dev(intents), dev(hidden), dev(a2t), dev(a2b), content, M, k, D, A)
    check(content.cpu()[:, :800], want, what='intent fuse')
    assert (content.cpu()[:, 800:] == 0).all()


def test_additive_pool(ops):
    n_seq, S, A, D = 11, 32, 400, 200
    hidden, x, a2 = torch.tanh(rnd(n_seq * S, A, seed=1, scale=2)), rnd(n_seq * S, D, seed=2), rnd(A, seed=3, scale=0.3)
    lens = torch.tensor([1, 32, 5, 0, 17, 32, 31, 2, 9, 12, 30])
    mask = torch.arange(S)[None, :] < lens[:, None]
    a = (hidden @ a2).view(n_seq, S).masked_fill(mask == 0, -1e9)
    want = torch.bmm(torch.softmax(a, dim=1).unsqueeze(1), x.view(n_seq, S, D)).squeeze(1)
    check(ops.additive_pool(dev(hidden), dev(a2), dev(x), n_seq, S, mask=dev(mask)), want, what='additive pool')


@pytest.mark.parametrize('B,N,H', [(5, 5, 50), (3, 1, 10), (4, 2, 70), (2, 20, 130)])
def test_cand_attn_weights(ops, B, N, H):
    D, heads = 400, 10
    qp, kp = rnd(B * N, D, seed=1, scale=3), rnd(B * H, D, seed=2, scale=3)
    lens = torch.randint(0, H + 1, (B,), generator=torch.Generator().manual_seed(3))
    lens[0] = 0
    mask = torch.arange(H)[None, :] < lens[:, None]
    Q = qp.view(B, N, heads, D // heads).transpose(1, 2)
    K = kp.view(B, H, heads, D // heads).transpose(1, 2)
    s = (Q @ K.transpose(-2, -1) / D ** 0.5).masked_fill(mask.view(B, 1, 1, H) == 0, -1e9)
    a = torch.softmax(s, dim=-1)
    qw = torch.softmax(torch.norm(qp.view(B, N, D), dim=-1), dim=1)
    want = torch.softmax((a.sum(dim=1) * qw.unsqueeze(-1)).sum(dim=1), dim=-1)
    for by_head in (True, False):                     # (row, head)-parallel two-launch form, and one workgroup per row
        ops.CAND_ATTN_BY_HEAD = by_head
        try:
            check(ops.cand_attn_weights(dev(qp), dev(kp), dev(mask), B, N, H, D, heads), want, what='cand attn weights (by head %s)' % by_head)
        finally:
            ops.CAND_ATTN_BY_HEAD = True


@pytest.mark.parametrize('B,H,n_user,n_src', [(4, 10, 4, 4), (6, 4, 6, 6), (3, 6, 8, 3), (5, 5, 5, 10)])
def test_sage_mean(ops, B, H, n_user, n_src):
    D = 400
    hist, un = rnd(B * H, D, seed=1), rnd(n_user, D, seed=2)
    X = torch.cat([hist.view(B, H, D), un.unsqueeze(0).expand(B, -1, -1)], dim=1)
    check(ops.sage_mean(dev(hist), dev(un), B, H, n_src, D), X[:, :n_src].mean(dim=1), what='sage mean')


def test_sage_mean_rejects_too_many_sources(ops):
    from lime_cikm25_amd._lib import LimeHipError
    with pytest.raises(LimeHipError):
        ops.sage_mean(dev(rnd(2 * 3, 8)), dev(rnd(2, 8)), 2, 3, 6, 8)


@pytest.mark.parametrize('B,N,H', [(6, 5, 50), (4, 1, 10), (3, 7, 33)])
@pytest.mark.parametrize('penalty', [True, False])
def test_interest_match(ops, B, N, H, penalty):
    A = D = 400
    kp, qp, g, cand = rnd(B * H, A, seed=1), rnd(B * N, A, seed=2), rnd(B * H, D, seed=3), rnd(B * N, D, seed=4)
    rem = torch.tensor([0.0, -0.0, 0.5, -0.5, 3.0, -3.0, 40.0, -40.0, 1e5, -1e5, 7.0, -2.0]).repeat(10)[:B * N].view(B, N)
    a = torch.einsum('bha,bna->bnh', kp.view(B, H, A), qp.view(B, N, A)) / math.sqrt(A)
    user = torch.softmax(a, dim=-1) @ g.view(B, H, D)
    cfg = type('C', (), dict(use_remaining_lifetime_weighting=True, use_expired_penalty=penalty, sigmoid_scaling_alpha=0.3,
                             penalty_scaling_beta=0.3))
    want = O.remaining_lifetime_weighting(cfg, user, cand.view(B, N, D), rem)
    u, l = ops.interest_match(dev(kp), dev(qp), dev(g), dev(cand), dev(rem), B, N, H, A, D, 1 / math.sqrt(A), 0.3, 0.3, True,
                              penalty)
    check(u, user, what='user rep')
    check(l, want, what='logits')
    z = want == 0
    assert torch.equal(torch.signbit(l.cpu()[z]), torch.signbit(want[z]))          # exact +-0
    l2 = ops.lifetime_score(dev(user), dev(cand.view(B, N, D)), dev(rem), 0.3, 0.3, True, penalty)
    check(l2, want, what='lifetime score')
    l3 = ops.lifetime_score(dev(user), dev(cand.view(B, N, D)), None, 0.3, 0.3, False, penalty)
    check(l3, (user * cand.view(B, N, D)).sum(-1), what='plain dot')


def test_row_scale_and_gather_rows(ops):
    x, s = rnd(50, 400, seed=1), rnd(50, seed=2)
    assert torch.equal(ops.row_scale(dev(x), dev(s)).cpu(), x * s[:, None])
    table = rnd(10, 500, seed=3)
    idx = torch.randint(0, 10, (64,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
    out = torch.zeros(64, 1000).cuda()
    ops.gather_rows(dev(idx), dev(table), out[:, 500:])
    assert torch.equal(out.cpu()[:, 500:], table[idx.long()])


def test_multi_copy(ops):
    g = torch.Generator().manual_seed(5)
    shapes = [((7,), torch.int32), ((3, 5), torch.float32), ((1000, 33), torch.uint8), ((0,), torch.float32),
              ((32, 50, 128), torch.int32), ((11,), torch.bool), ((4099,), torch.int64)] * 6           # 42 pairs: two launches
    srcs = []
    for shp, dt in shapes:
        t = torch.randint(0, 100, shp, generator=g)
        srcs.append(dev(t.to(dt)))
    big = torch.zeros(sum(s.numel() * s.element_size() + 7 for s in srcs) + 64, dtype=torch.uint8, device='cuda')
    dsts, off = [], 1                                       # deliberately misaligned destinations
    for s in srcs:
        nb = s.numel() * s.element_size()
        off = (off + s.element_size() - 1) // s.element_size() * s.element_size()
        dsts.append(big[off:off + nb].view(s.dtype).view(s.shape))
        off += nb + 3
    ops.multi_copy(list(zip(dsts, srcs)))
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    with pytest.raises(ValueError):
        ops.multi_copy([(dsts[0], srcs[1])])


# ---------------------------------------------------------------------------------------------------
# bf16 matrix-core path (BASELINE config 3).  Reference: the same operands rounded to bf16, fp32 arithmetic, result
# rounded to bf16 -- so the only differences are accumulation order and double rounding: tolerance 2 bf16 ulps (1.6e-2
# relative to the row scale, observed ~4e-3).
# ---------------------------------------------------------------------------------------------------
BF_TOL = 1.6e-2


def bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize('M,N,K', [(4096, 512, 304), (5003, 960, 304), (300, 256, 64), (4100, 304, 512), (4097, 640, 1000), (33, 320, 304)])
@pytest.mark.parametrize('act', [None, 'relu'])
def test_linear_bf16_plain(ops, M, N, K, act):
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=1 / math.sqrt(K))), rnd(N, seed=3)
    want = a.float() @ w.float().t() + b
    if act == 'relu':
        want = torch.relu(want)
    got = ops.linear_bf16(dev(a), dev(w), dev(b), act=act)
    assert got.dtype == torch.bfloat16 and last_kernel().split(', ')[4].startswith('true'), last_kernel()
    check(got.float(), bf(want).float(), tol=BF_TOL, what='bf16 linear %s' % ((M, N, K, act),))


def test_to_bf16_pads_and_rounds(ops):
    x = rnd(37, 300, seed=4)
    got = ops.to_bf16(dev(x), rows_out=40, cols_out=304).cpu()
    want = torch.zeros(40, 304, dtype=torch.bfloat16)
    want[:37, :300] = bf(x)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    v = ops.to_bf16(dev(rnd(300, seed=5)), cols_out=304).cpu()
    assert v.shape == (304,) and torch.equal(v[:300].view(torch.int16), bf(rnd(300, seed=5)).view(torch.int16)) and (v[300:] == 0).all()


@pytest.mark.parametrize('M,S', [(4224, 32), (4200, 128)])
def test_linear_bf16_encoder_layer_gemms(ops, M, S):
    """The four GEMMs of an encoder layer as the bf16 path issues them (K = N = 300 padded to 304 with zeros)."""
    V, E, EP, F, W = 600, 300, 304, 512, 960
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    table, pe = rnd(V, E, seed=7), rnd(S, E, seed=8)
    w_in, b_in = rnd(W, E, seed=9, scale=0.06), rnd(W, seed=10)
    w_o, b_o = rnd(E, E, seed=11, scale=0.06), rnd(E, seed=12)
    w1, b1 = rnd(F, E, seed=13, scale=0.06), rnd(F, seed=14)
    w2, b2 = rnd(E, F, seed=15, scale=0.05), rnd(E, seed=16)
    g1, be1, g2, be2 = rnd(E, seed=17) + 1.5, rnd(E, seed=18), rnd(E, seed=19) + 1.5, rnd(E, seed=20)
    pad_v = lambda v: dev(torch.cat([v, torch.zeros(EP - E)]))
    tb = ops.to_bf16(dev(table), cols_out=EP)
    # in_proj: gathered bf16 rows, fp32 periodic residual (PE W^T + b)
    pew = (pe @ w_in.t() + b_in)
    qkv = ops.linear_bf16(tb, ops.to_bf16(dev(w_in), cols_out=EP), None, a_ids=dev(ids), res=dev(pew), res_kind=1, res_mod=S)
    x_bf = bf(table)[ids.long()].float()
    want_qkv = x_bf @ bf(w_in).float().t() + pew[torch.arange(M) % S]
    check(qkv.float(), bf(want_qkv).float(), tol=BF_TOL, what='bf16 in_proj')
    # out_proj: residual = bf16 table rows + fp32 pe, LayerNorm over the 300 real columns, N padded to 304
    attn = bf(rnd(M, E, seed=21))
    attn_p = torch.zeros(M, EP, dtype=torch.bfloat16)
    attn_p[:, :E] = attn
    x1 = ops.linear_bf16(dev(attn_p), ops.to_bf16(dev(w_o), rows_out=EP, cols_out=EP), pad_v(b_o), res=tb, res_kind=2,
                         res_ids=dev(ids), res_pe=dev(torch.cat([pe, torch.zeros(S, EP - E)], dim=1)), res_period=S,
                         ln=(pad_v(g1), pad_v(be1)), ln_count=E)
    want_x1 = O.layer_norm(x_bf + pe[torch.arange(M) % S] + attn.float() @ bf(w_o).float().t() + b_o, g1, be1)
    assert (x1[:, E:] == 0).all()
    check(x1[:, :E].float(), bf(want_x1).float(), tol=BF_TOL, what='bf16 out_proj + LN')
    # linear1 (ReLU) and linear2 (bf16 residual + LN)
    h = ops.linear_bf16(x1, ops.to_bf16(dev(w1), cols_out=EP), dev(b1), act='relu')
    x1f = x1[:, :E].float().cpu()
    want_h = torch.relu(x1f @ bf(w1).float().t() + b1)
    check(h.float(), bf(want_h).float(), tol=BF_TOL, what='bf16 linear1')
    x2 = ops.linear_bf16(h, ops.to_bf16(dev(w2), rows_out=EP), pad_v(b2), res=x1, res_kind=3, ln=(pad_v(g2), pad_v(be2)), ln_count=E)
    want_x2 = O.layer_norm(x1f + h.float().cpu() @ bf(w2).float().t() + b2, g2, be2)
    assert (x2[:, E:] == 0).all()
    check(x2[:, :E].float(), bf(want_x2).float(), tol=BF_TOL, what='bf16 linear2 + LN')
    mb = M // 32 * 32                                            # pool32: fp32 means over 32-row blocks straight from the epilogue
    blocks = ops.linear_bf16(h[:mb], ops.to_bf16(dev(w2), rows_out=EP), pad_v(b2), res=x1[:mb], res_kind=3, ln=(pad_v(g2), pad_v(be2)),
                             ln_count=E, pool32=True)
    assert blocks.dtype == torch.float32 and blocks.shape == (mb // 32, EP) and (blocks[:, E:] == 0).all()
    check(blocks[:, :E], want_x2[:mb].view(mb // 32, 32, E).mean(dim=1), tol=BF_TOL, what='bf16 linear2 + LN + pool32')
    n_seq = M // S
    pooled = ops.mean_pool_bf16(x2[:n_seq * S], n_seq, S, E)
    check(pooled, x2[:n_seq * S, :E].float().cpu().view(n_seq, S, E).mean(dim=1), what='bf16 mean pool')


@pytest.mark.parametrize('M,S,compact', [(4096, 128, False), (4200, 32, False), (36001, 128, True), (130, 128, True)])
def test_inproj_bf16(ops, M, S, compact):
    """The activation-stationary q / k / v projection against the same arithmetic in torch: gathered word rows, the fp32 periodic
    rows indexed by the OUTPUT row, rows scattered by c_ids (compacted batch) or in place; row counts that end inside a tile."""
    V, E, EP, N = 700, 300, 304, 960
    table = torch.zeros(V, EP, dtype=torch.bfloat16)
    table[:, :E] = bf(rnd(V, E, seed=70))
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(71), dtype=torch.int32)
    w = rnd(N, E, seed=72, scale=0.06)
    add = rnd(S, N, seed=73)
    cap = M + 77 if compact else M
    c_ids = torch.randperm(cap, generator=torch.Generator().manual_seed(74))[:M].to(torch.int32) if compact else None
    orow = c_ids.long() if compact else torch.arange(M)
    want = table[ids.long(), :E].float() @ bf(w).float().t() + add[orow % S]
    wp = ops.inproj_pack_bf16(dev(w), EP)
    # packed [pass][chunk][320 rows][32 k]; row 16 t + 4 kg + q of a pass is output column 32 (t >> 1) + 8 kg + 4 (t & 1) + q
    wv = wp.cpu().view(N // 320, 10, 10, 2, 4, 4, 32).permute(0, 2, 4, 3, 5, 1, 6).reshape(N, 320)      # [pass][t >> 1][kg][t & 1][q] x [chunk][k]
    assert torch.equal(wv[:, :E].view(torch.int16), bf(w).view(torch.int16)) and (wv[:, E:] == 0).all()
    out = torch.full((cap, N), 3.0, dtype=torch.bfloat16, device='cuda')
    ops.inproj_bf16(dev(table), wp, dev(add), N, out, a_ids=dev(ids), c_ids=dev(c_ids))
    got = out.cpu()
    check(got[orow].float(), bf(want).float(), tol=BF_TOL, what='inproj bf16 %s' % ((M, S, compact),))
    if compact:
        untouched = torch.ones(cap, dtype=torch.bool)
        untouched[orow] = False
        assert (got[untouched] == 3.0).all()
    # a device-side row count: only the first rows are produced
    m_dev = torch.tensor([M // 2 + 3], dtype=torch.int32, device='cuda')
    out2 = torch.full((cap, N), 3.0, dtype=torch.bfloat16, device='cuda')
    ops.inproj_bf16(dev(table), wp, dev(add), N, out2, a_ids=dev(ids), c_ids=dev(c_ids), m_dev=m_dev)
    g2 = out2.cpu()
    k = M // 2 + 3
    assert torch.equal(g2[orow[:k]].view(torch.int16), got[orow[:k]].view(torch.int16)) and (g2[orow[k:]] == 3.0).all()


def test_inproj_bf16_more_than_2gb_of_rows(ops):
    """Rows written in place are addressed from their tile's first row: an output beyond 2 GB (the dense config-3 pass) works."""
    V, E, EP, N, S = 500, 300, 304, 960, 128
    M = 1_200_000 // S * S                                       # x 1,920 bytes = 2.3 GB
    g = torch.Generator(device='cuda').manual_seed(5)
    table = torch.zeros(V, EP, dtype=torch.bfloat16, device='cuda')
    table[:, :E] = ((torch.rand(V, E, generator=g, device='cuda') * 2 - 1)).to(torch.bfloat16)
    ids = torch.randint(0, V, (M,), generator=g, device='cuda', dtype=torch.int32)
    w = (torch.rand(N, E, generator=g, device='cuda') * 2 - 1) * 0.06
    add = torch.rand(S, N, generator=g, device='cuda') * 2 - 1
    out = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
    ops.inproj_bf16(table, ops.inproj_pack_bf16(w, EP), add, N, out, a_ids=ids)
    for lo in (0, M // 2 - 77, M - 4096):
        rows = torch.arange(lo, lo + 4096, device='cuda')
        want = table[ids[rows].long(), :E].float() @ w.to(torch.bfloat16).float().t() + add[rows % S]
        check(out[rows].float(), want.to(torch.bfloat16).float().cpu(), tol=BF_TOL, what='inproj rows %d..' % lo)
    del out


def _ffn_case(M, F, seed):
    E, EP = 300, 304
    x = torch.zeros(M, EP, dtype=torch.bfloat16)
    x[:, :E] = bf(rnd(M, E, seed=seed, scale=1.5))
    w1, b1 = rnd(F, E, seed=seed + 1, scale=0.06), rnd(F, seed=seed + 2)
    w2, b2 = rnd(E, F, seed=seed + 3, scale=0.05), rnd(E, seed=seed + 4)
    g, be = rnd(E, seed=seed + 5) + 1.5, rnd(E, seed=seed + 6)
    xf = x[:, :E].float()
    h = bf(torch.relu(xf @ bf(w1).float().t() + bf(b1).float())).float()          # b1 rides in the GEMM: applied in bf16
    want = O.layer_norm(xf + h @ bf(w2).float().t() + b2, g, be)
    return x, (w1, b1, w2, b2, g, be), want


@pytest.mark.parametrize('M,F', [(128, 512), (4096, 512), (4128, 512), (1000, 256), (33, 128), (36000, 512)])
def test_encoder_ffn_bf16(ops, M, F):
    """linear1 + ReLU + linear2 + residual + LayerNorm in one launch against the same arithmetic in torch (bf16 operands, fp32
    accumulation), rows and pooled block means; row counts that end inside a tile / a wave, one tile per workgroup and several."""
    E = 300
    x, (w1, b1, w2, b2, g, be), want = _ffn_case(M, F, seed=40)
    w1p, w2p = ops.ffn_pack_bf16(dev(w1), dev(b1), dev(w2))
    # the packed layouts: w1p [F / 128][10 chunks][128 rows][32 k] (K = 320: W1, b1 at k = E, zeros), w2p [F / 32][304 rows][32]
    w1v = w1p.cpu().view(F // 128, 10, 128, 32).permute(0, 2, 1, 3).reshape(F, 320)
    assert torch.equal(w1v[:, :E].view(torch.int16), bf(w1).view(torch.int16)) and torch.equal(w1v[:, E].view(torch.int16), bf(b1).view(torch.int16))
    assert (w1v[:, E + 1:] == 0).all()
    w2v = w2p.cpu().view(F // 32, 304, 4, 2, 4).permute(1, 0, 3, 2, 4).reshape(304, F)          # [n][block][a][kg][r]
    assert torch.equal(w2v[:E].view(torch.int16), bf(w2).view(torch.int16)) and (w2v[E:] == 0).all()
    got = ops.encoder_ffn_bf16(dev(x), w1p, w2p, dev(b2), (dev(g), dev(be)), 1e-5, E)
    assert got.dtype == torch.bfloat16 and got.shape == (M, 304) and (got[:, E:] == 0).all()
    check(got[:, :E].float(), bf(want).float(), tol=BF_TOL, what='fused bf16 ffn rows %s' % ((M, F),))
    mb = M // 32 * 32
    if mb:
        blocks = ops.encoder_ffn_bf16(dev(x[:mb]), w1p, w2p, dev(b2), (dev(g), dev(be)), 1e-5, E, pool32=True)
        assert blocks.dtype == torch.float32 and blocks.shape == (mb // 32, 304) and (blocks[:, E:] == 0).all()
        check(blocks[:, :E], want[:mb].view(mb // 32, 32, E).mean(dim=1), tol=2e-3, what='fused bf16 ffn block means %s' % ((M, F),))


@pytest.mark.parametrize('M,S,kind', [(4096, 128, 2), (4128, 32, 2), (1000, 128, 3), (36000, 128, 2)])
def test_encoder_block_bf16(ops, M, S, kind):
    """out_proj + residual + norm1 + the feed-forward half in one launch against the same arithmetic in torch: the gathered
    (word rows + positional rows) and the dense residual, rows and pooled block means."""
    E, EP, F, V = 300, 304, 512, 700
    x, (w1, b1, w2, b2, g2, be2), _ = _ffn_case(8, F, seed=60)
    attn = torch.zeros(M, EP, dtype=torch.bfloat16)
    attn[:, :E] = bf(rnd(M, E, seed=61))
    w0, b0 = rnd(E, E, seed=62, scale=0.06), rnd(E, seed=63)
    g1, be1 = rnd(E, seed=64) + 1.5, rnd(E, seed=65)
    if kind == 2:
        table = torch.zeros(V, EP, dtype=torch.bfloat16)
        table[:, :E] = bf(rnd(V, E, seed=66))
        ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(67), dtype=torch.int32)
        pe = rnd(S, E, seed=68)
        res_f = table[ids.long(), :E].float() + pe[torch.arange(M) % S]
        kw = dict(res=dev(table), res_kind=2, res_ids=dev(ids), add_rows=dev(pe + b0))
    else:
        xr = torch.zeros(M, EP, dtype=torch.bfloat16)
        xr[:, :E] = bf(rnd(M, E, seed=69))
        res_f = xr[:, :E].float()
        kw = dict(res=dev(xr), res_kind=3, add_rows=dev(b0.view(1, E)))
    x1 = bf(O.layer_norm(res_f + attn[:, :E].float() @ bf(w0).float().t() + b0, g1, be1)).float()
    h = bf(torch.relu(x1 @ bf(w1).float().t() + bf(b1).float())).float()
    want = O.layer_norm(x1 + h @ bf(w2).float().t() + b2, g2, be2)
    w0p = ops.oproj_pack_bf16(dev(w0))
    w0v = w0p.cpu().view(10, 304, 32).permute(1, 0, 2).reshape(304, 320)
    assert torch.equal(w0v[:E, :E].view(torch.int16), bf(w0).view(torch.int16)) and (w0v[E:] == 0).all() and (w0v[:, E:] == 0).all()
    w1p, w2p = ops.ffn_pack_bf16(dev(w1), dev(b1), dev(w2))
    common = dict(w1p=w1p, w2p=w2p, b2=dev(b2), ln2=(dev(g2), dev(be2)), ln2_eps=1e-5, E=E)
    common.update(ln1=(dev(g1), dev(be1)), ln1_eps=1e-5)
    got = ops.encoder_block_bf16(dev(attn), w0p, **kw, **common)
    assert got.dtype == torch.bfloat16 and got.shape == (M, 304) and (got[:, E:] == 0).all()
    check(got[:, :E].float(), bf(want).float(), tol=2 * BF_TOL, what='fused bf16 block rows %s' % ((M, S, kind),))
    mb = M // 32 * 32
    kwb = {k: (v[:mb] if k in ('res_ids',) or (k == 'res' and kind == 3) else v) for k, v in kw.items()}
    blocks = ops.encoder_block_bf16(dev(attn[:mb]), w0p, pool32=True, **kwb, **common)
    assert blocks.dtype == torch.float32 and blocks.shape == (mb // 32, 304) and (blocks[:, E:] == 0).all()
    check(blocks[:, :E], want[:mb].view(mb // 32, 32, E).mean(dim=1), tol=4e-3, what='fused bf16 block means %s' % ((M, S, kind),))


def test_encoder_ffn_bf16_device_row_count_and_refusals(ops):
    E, F, M = 300, 512, 4096
    x, (w1, b1, w2, b2, g, be), want = _ffn_case(M, F, seed=50)
    w1p, w2p = ops.ffn_pack_bf16(dev(w1), dev(b1), dev(w2))
    out = torch.full((M // 32, 304), 7.0, device='cuda')
    m_dev = torch.tensor([1056], dtype=torch.int32, device='cuda')
    ops.encoder_ffn_bf16(dev(x), w1p, w2p, dev(b2), (dev(g), dev(be)), 1e-5, E, pool32=True, m_dev=m_dev, out=out)
    check(out[:33, :E], want[:1056].view(33, 32, E).mean(dim=1), tol=2e-3, what='fused ffn, device row count')
    assert (out[33:] == 7.0).all()
    from lime_cikm25_amd._lib import LimeHipError
    with pytest.raises(LimeHipError):
        ops.ffn_pack_bf16(dev(rnd(500, E)), dev(rnd(500)), dev(rnd(E, 500)))           # F % 128
    with pytest.raises(LimeHipError):
        ops.ffn_pack_bf16(dev(rnd(512, 256)), dev(rnd(512)), dev(rnd(256, 512)))       # E outside 289..303


@pytest.mark.parametrize('S', [32, 64, 128, 256])
def test_token_attention_bf16(ops, S):
    n_seq, h, hd = (23 if S <= 128 else 3), 10, 30
    E = h * hd
    qkv = bf(rnd(n_seq * S, 3 * E, seed=S + 7, scale=2.0)).float()
    pad = torch.zeros(n_seq * S, 3 * h, 32)
    pad[:, :, :hd] = qkv.view(n_seq * S, 3 * h, hd)
    d = dev(bf(pad.view(n_seq * S, 3 * h * 32)))
    W = h * 32
    scale = 1.0 / math.sqrt(hd)
    got = ops.token_attention_bf16(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], n_seq, S, h, hd, scale, out_cols=304)
    assert got.shape == (n_seq * S, 304) and (got[:, E:] == 0).all()
    check(got[:, :E].float(), bf(attn_ref(qkv, n_seq, S, h, hd, scale)).float(), tol=BF_TOL, what='bf16 attn S=%d' % S)
    from lime_cikm25_amd._lib import LimeHipError
    with pytest.raises(LimeHipError):
        ops.token_attention_bf16(d[:96 * 2, :W], d[:96 * 2, W:2 * W], d[:96 * 2, 2 * W:], 2, 96, h, hd, scale)


# ---------------------------------------------------------------------------------------------------
# gemm_mid_kernel (csrc/gemm_mid_f32.hip): randomized shapes / epilogues against an fp64 statement
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('seed', range(24))
def test_linear_mid_kernel_randomized(ops, seed):
    """Small / mid-M launches with 16-byte friendly operands take the LDS-DMA ring kernel: random M, N, K (tails in every
    direction), activation, bias, a gathered A operand, the three residual kinds, a device-side row count."""
    import ctypes
    from lime_cikm25_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([1, 3, 31, 64, 65, 127, 200, 700, 1760, 3000]))
    N = int(rng.choice([4, 60, 64, 68, 200, 400, 900])) 
    K = int(rng.choice([16, 20, 52, 300, 400, 1000, 1800]))
    act = [None, 'relu', 'tanh', 'sigmoid'][int(rng.integers(0, 4))]
    bias = rnd(N, seed=seed + 1) if rng.random() < 0.7 else None
    w = rnd(N, K, seed=seed + 2, scale=1 / math.sqrt(K))
    gather = rng.random() < 0.3
    if gather:
        table = rnd(500, K, seed=seed + 3)
        ids = torch.from_numpy(rng.integers(0, 500, size=M).astype(np.int32))
        a_full = table[ids.long()]
        a_arg, ids_arg = dev(table), dev(ids)
    else:
        a_full = rnd(M, K, seed=seed + 3)
        a_arg, ids_arg = dev(a_full), None
    kind = int(rng.integers(0, 4))                          # 0 none, 1 broadcast rows (res_div), 2 periodic (res_mod), 3 gathered (res_ids)
    kw, res_rows = {}, None
    if kind == 1:
        div = int(rng.choice([1, 2, 50]))
        res = rnd((M + div - 1) // div, N, seed=seed + 4)
        res_rows = res[torch.arange(M) // div]
        kw = dict(res=dev(res), res_div=div)
    elif kind == 2:
        mod = int(rng.choice([1, 7, 32]))
        res = rnd(mod, N, seed=seed + 4)
        res_rows = res[torch.arange(M) % mod]
        kw = dict(res=dev(res), res_mod=mod)
    elif kind == 3:
        res = rnd(100, N, seed=seed + 4)
        rid = torch.from_numpy(rng.integers(0, 100, size=M).astype(np.int32))
        res_rows = res[rid.long()]
        kw = dict(res=dev(res), res_ids=dev(rid))
    want = a_full.double() @ w.double().t()
    if bias is not None:
        want = want + bias.double()
    want = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid, None: lambda x: x}[act](want)
    if res_rows is not None:
        want = want + res_rows.double()
    m_live = int(rng.integers(0, M + 1)) if rng.random() < 0.4 else None
    out = torch.full((M, N), float('nan'), device='cuda')
    m_dev = torch.tensor([m_live], dtype=torch.int32, device='cuda') if m_live is not None else None
    got = ops.linear(a_arg, dev(w), dev(bias), act=act, a_ids=ids_arg, out=out, m_dev=m_dev, **kw)
    torch.cuda.synchronize()
    assert ctypes.string_at(_lib.load().lime_last_linear_kernel()).decode() == 'gemm_mid_kernel'
    rows = M if m_live is None else m_live
    if rows:
        check(got[:rows], want[:rows].float(), what='M=%d N=%d K=%d act=%s kind=%d gather=%s m_dev=%s' % (M, N, K, act, kind, gather, m_live))
    assert torch.isnan(got[rows:]).all()                     # rows beyond the device count are left untouched


@pytest.mark.parametrize('G,H,D,row_div,n_src,n_user', [(12, 50, 400, 1, 12, 64), (12, 50, 400, 4, 70, 64), (6, 10, 400, 3, 10, 8), (8, 7, 128, 2, 3, 4),
                                                       (5, 50, 512, 1, 114, 64), (9, 33, 300, 9, 40, 16)])
def test_gate_ln_sage_fused(ops, G, H, D, row_div, n_src, n_user):
    """The fused gated-residual LayerNorm + GraphSAGE aggregate against gate_ln on the per-candidate copies + sage_mean over
    cat[refined history, user nodes] (what the two separate launches compute), and against a plain torch statement."""
    Gh = G // row_div
    x, y = rnd(Gh * H, D, seed=1), rnd(Gh * H, D, seed=2)
    scale, bias = torch.rand(G * H, generator=torch.Generator().manual_seed(3)), rnd(D, seed=4)
    gamma, beta = rnd(D, seed=5) + 1.5, rnd(D, seed=6)
    un = rnd(n_user, D, seed=7)
    xr = x.view(Gh, H, D).repeat_interleave(row_div, dim=0).reshape(G * H, D)
    yr = y.view(Gh, H, D).repeat_interleave(row_div, dim=0).reshape(G * H, D)
    gt = torch.sigmoid(scale[:, None] * yr + bias)
    blend = gt * (scale[:, None] * xr) + (1 - gt) * xr
    want = O.layer_norm(blend, gamma, beta)
    nodes = torch.cat([want.view(G, H, D), un.unsqueeze(0).expand(G, -1, -1)], dim=1)
    want_mean = nodes[:, :n_src].mean(dim=1)
    const = dev(un[:n_src - H].sum(dim=0)) if n_src > H else None
    got, mean = ops.gate_ln_sage(dev(y), dev(x), dev(scale), dev(bias), dev(gamma), dev(beta), 1e-5, G, H, D, row_div, n_src, const)
    check(got, want, what='gate_ln_sage rows')
    check(mean, want_mean, what='gate_ln_sage mean')
    two = ops.gate_ln(dev(yr), dev(xr), dev(scale), dev(bias), dev(gamma), dev(beta))
    check(got, two.cpu().view(G * H, D), what='fused vs gate_ln')
    check(mean, ops.sage_mean(two.view(G * H, D), dev(un), G, H, n_src, D).cpu(), what='fused vs sage_mean')


def test_linear_group_equals_separate_launches(ops):
    """lime_linear_group_f32: independent small GEMMs in one launch -- bit for bit what the single launches of the same kernel give,
    for plain / tanh / gathered-A + periodic-residual / broadcast-residual problems of different shapes."""
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 300, (128,), generator=g, dtype=torch.int32).cuda()
    table, pew = dev(rnd(300, 300, seed=1)), dev(rnd(32, 960, seed=2))
    probs = [dict(a=dev(rnd(32, 300, seed=3)), w=dev(rnd(960, 300, seed=4, scale=0.05)), bias=dev(rnd(960, seed=5))),
             dict(a=dev(rnd(5280, 400, seed=6))[:3000], w=dev(rnd(400, 400, seed=7, scale=0.05)), bias=dev(rnd(400, seed=8)), act='tanh'),
             dict(a=table, w=dev(rnd(960, 300, seed=9, scale=0.05)), bias=None, a_ids=ids, res=pew, res_mod=32),
             dict(a=dev(rnd(1600, 400, seed=10)), w=dev(rnd(400, 400, seed=11, scale=0.05)), bias=None, res=dev(rnd(32, 400, seed=12)), res_div=50)]
    ops.GROUP_SMALL_GEMMS = False
    try:
        want = [o.clone() for o in ops.linear_group([dict(p) for p in probs])]
    finally:
        ops.GROUP_SMALL_GEMMS = True
    got = ops.linear_group([dict(p) for p in probs])
    assert last_kernel() == 'gemm_mid_group_kernel', last_kernel()
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and torch.equal(a, b), 'problem %d' % i
    # a problem the mid-M kernel does not take (LayerNorm) sends the whole group down the single-launch path
    lnp = dict(a=dev(rnd(200, 300, seed=15)), w=dev(rnd(300, 300, seed=16, scale=0.05)), bias=None, ln=(dev(rnd(300, seed=17) + 1.5), dev(rnd(300, seed=18))))
    outs = ops.linear_group([dict(probs[0]), lnp])
    check(outs[1], O.layer_norm(lnp['a'].cpu() @ lnp['w'].cpu().t(), lnp['ln'][0].cpu(), lnp['ln'][1].cpu()), what='fallback')


def test_misaligned_layernorm_epilogue_warns_once(ops):
    """A big LayerNorm GEMM on a view that is not 16-byte aligned still computes (general kernel) but says that it is the slow path."""
    from lime_cikm25_amd import ops as O
    M, N, K = 4200, 300, 64
    wide = dev(rnd(M, K + 4, seed=1))
    a = wide[:, 1:K + 1]                                   # 4-byte offset: not 16-byte aligned
    w, res = dev(rnd(N, K, seed=2, scale=0.2)), dev(rnd(M, N, seed=3))
    g, b = dev(rnd(N, seed=4) * 0.5 + 1.0), dev(rnd(N, seed=5))
    O._SLOW_LN_WARNED = False
    with pytest.warns(RuntimeWarning, match='slow general kernel'):
        got = ops.linear(a, w, None, res=res, ln=(g, b))
    want = torch.nn.functional.layer_norm(a.cpu().double() @ w.cpu().double().t() + res.cpu().double(), (N,), g.cpu().double(), b.cpu().double())
    check(got, want.float(), what='misaligned LayerNorm GEMM')
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        ops.linear(a, w, None, res=res, ln=(g, b))          # only once per process


def test_cand_attn_weights_shared_history(ops):
    """hist_div: one history (key projections, mask) for hist_div consecutive rows equals the call on the repeated history, bit for bit."""
    Bh, K, N, H, D, nh = 6, 5, 1, 50, 400, 10
    B = Bh * K
    qp, kp = dev(rnd(B * N, D, seed=1)), dev(rnd(Bh * H, D, seed=2))
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(Bh, H, generator=g) > 0.3)
    mask[:, 0] = True
    shared = ops.cand_attn_weights(qp.view(-1), kp.view(-1), dev(mask), B, N, H, D, nh, hist_div=K)
    full = ops.cand_attn_weights(qp.view(-1), kp.view(Bh, H * D).repeat_interleave(K, dim=0).reshape(-1), dev(mask.repeat_interleave(K, dim=0)),
                                 B, N, H, D, nh)
    assert torch.equal(shared, full)


@pytest.mark.parametrize('S', [2, 3, 4, 8, 16])
def test_mean_pool_counted(ops, S):
    """lime_mean_pool_count_f32: the first *n_seq_dev sequences bit-equal to lime_mean_pool_f32's rows, nothing written behind the count."""
    n_seq, E = 700, 300
    x = dev(rnd(n_seq * S, 304, seed=S))[:, :E]
    full = ops.mean_pool(x, n_seq, S)
    want = x.view(n_seq, S, E).double().mean(dim=1)
    assert rel_err(full.cpu().numpy(), want.cpu().numpy()) < 1e-6
    for live in (0, 1, 333, 700, 900):
        out = torch.full((n_seq, E), -7.0, device='cuda')
        ops.mean_pool(x, n_seq, S, out=out, n_seq_dev=torch.tensor([live], dtype=torch.int32, device='cuda'))
        k = min(live, n_seq)
        assert torch.equal(out[:k], full[:k])
        assert bool((out[k:] == -7.0).all())
    with pytest.raises(Exception):
        ops.mean_pool(dev(rnd(4 * 32, 300, seed=1)), 4, 32, n_seq_dev=torch.tensor([2], dtype=torch.int32, device='cuda'))


def test_linear_mid_tile_shapes_agree_bitwise(ops):
    """gemm_mid_kernel picks its tile shape by the problem's tile count (32 x 32 while those fit one round of workgroups, 32 x 64, 64 x 64);
    an output element sees the same k order in all three, so a row's result does not depend on how many rows the launch has."""
    import ctypes
    from lime_cikm25_amd import _lib
    N, K = 400, 400
    a = dev(rnd(9000, K, seed=1))
    w, b = dev(rnd(N, K, seed=2, scale=0.05)), dev(rnd(N, seed=3))
    outs = {}
    for M in (40, 1700, 3000, 9000):                 # 32 x 32 (26 and 689 tiles), 32 x 64 (658), 64 x 64 (987)
        outs[M] = ops.linear(a[:M], w, b, act='tanh')
        assert ctypes.string_at(_lib.load().lime_last_linear_kernel()).decode() == 'gemm_mid_kernel'
    want = torch.tanh(a.double() @ w.double().t() + b.double())
    check(outs[9000], want.float().cpu(), what='64 x 64 tiles')
    for M in (40, 1700, 3000):
        assert torch.equal(outs[M], outs[9000][:M]), 'M = %d differs from the rows of the M = 9000 launch' % M
