"""Training-mode dropout inside the token encoders on the MI355X.  torch's random stream cannot be matched (it differs between
torch's own CPU and GPU generators), so the arithmetic is pinned differently: the masks the kernels use are read back through
``ops.dropout`` on all-ones tensors (same seed / site / element index) and fed to a plain torch fp64 statement of the post-LN
encoder layer with explicit masks; forward and every gradient of the HIP layer must match that (1e-3, north star)."""
import math

import pytest
import torch
import torch.nn.functional as F

import golden_cases
from helpers import rel_err
from lime_cikm25_amd import Model, synth
from lime_cikm25_amd import training as T

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from lime_cikm25_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_mask_statistics_and_determinism(ops):
    ones = torch.ones(4096, 300, device='cuda')
    for p in (0.1, 0.2, 0.5):
        m = ops.dropout(ones, p, 1234, 3)
        kept = (m != 0).float().mean().item()
        assert abs(kept - (1 - p)) < 3e-3, (p, kept)
        assert torch.allclose(m[m != 0], torch.tensor(1.0 / (1 - p), device='cuda'))
        assert torch.equal(m, ops.dropout(ones, p, 1234, 3))                       # a pure function of (seed, site, index)
        assert not torch.equal(m, ops.dropout(ones, p, 1234, 4))                   # another site
        assert not torch.equal(m, ops.dropout(ones, p, 1235, 3))                   # another seed
        rows = (m != 0).float().mean(dim=1)
        assert rows.std().item() < 3 * math.sqrt(p * (1 - p) / 300)                # no structure along the rows
    assert torch.equal(ops.dropout(ones, 0.0, 1, 1), ones)
    x = rnd(100, 64, seed=1).cuda()
    y = x.clone()
    ops.dropout(y, 0.3, 7, 0, out=y)                                               # in place
    assert torch.equal(y, ops.dropout(x, 0.3, 7, 0))


@pytest.mark.parametrize('M,S', [(3, 16), (5, 32), (2, 128), (4, 50), (2, 200), (1, 512)])
def test_encoder_layer_with_dropout_matches_torch_on_the_same_masks(ops, M, S):
    E, nh, Fd, V, p, seed = 300, 10, 512, 400, 0.2, 987654321
    hd = E // nh
    tok = M * S
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (M, S), generator=g, dtype=torch.int32)
    names = ['table', 'in_w', 'in_b', 'out_w', 'out_b', 'l1_w', 'l1_b', 'l2_w', 'l2_b', 'n1_w', 'n1_b', 'n2_w', 'n2_b']
    shapes = [(V, E), (3 * E, E), (3 * E,), (E, E), (E,), (Fd, E), (Fd,), (E, Fd), (E,), (E,), (E,), (E,), (E,)]
    vals = {}
    for i, (n, sh) in enumerate(zip(names, shapes)):
        v = rnd(*sh, seed=20 + i, scale=0.5 if n == 'table' else (1.0 / math.sqrt(sh[-1]) if len(sh) == 2 else 0.1))
        if n in ('n1_w', 'n2_w'):
            v = v + 1.0
        vals[n] = v
    pe = rnd(S, E, seed=40)
    G = rnd(M, E, seed=41)

    # HIP layer
    dev = {n: v.clone().cuda().requires_grad_(True) for n, v in vals.items()}
    pooled = T._TokenEncoder.apply(ids.cuda(), nh, 1e-5, 1e-5, p, seed, dev['table'], pe.cuda(), *[dev[n] for n in names[1:]])
    (pooled * G.cuda()).sum().backward()

    # the kernels' masks, read back through the same generator
    ones = lambda r, c: torch.ones(r, c, device='cuda')
    m_emb = ops.dropout(ones(tok, E), p, seed, T._SITE_EMB).cpu().double()
    m_pe = ops.dropout(ones(tok, E), p, seed, T._SITE_PE).cpu().double()
    m_att = ops.dropout(ones(M * nh * S, S), p, seed, T._SITE_ATTN).cpu().double().view(M, nh, S, S)
    m_d1 = ops.dropout(ones(tok, E), p, seed, T._SITE_DROP1).cpu().double()
    m_ff = ops.dropout(ones(tok, Fd), p, seed, T._SITE_FF).cpu().double()
    m_d2 = ops.dropout(ones(tok, E), p, seed, T._SITE_DROP2).cpu().double()

    # torch fp64 statement of nn.TransformerEncoderLayer (post-LN, ReLU) + the two input dropouts + mean pooling
    ref = {n: v.double().requires_grad_(True) for n, v in vals.items()}
    x0 = m_pe * (m_emb * ref['table'][ids.long().reshape(-1)] + pe.double().repeat(M, 1))
    qkv = x0 @ ref['in_w'].t() + ref['in_b']
    q, k, v = (t.reshape(M, S, nh, hd).permute(0, 2, 1, 3) for t in qkv.split(E, dim=1))
    P = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1) * m_att
    ao = (P @ v).permute(0, 2, 1, 3).reshape(tok, E)
    x1 = F.layer_norm(x0 + m_d1 * (ao @ ref['out_w'].t() + ref['out_b']), (E,), ref['n1_w'], ref['n1_b'], 1e-5)
    h = m_ff * torch.relu(x1 @ ref['l1_w'].t() + ref['l1_b'])
    y = F.layer_norm(x1 + m_d2 * (h @ ref['l2_w'].t() + ref['l2_b']), (E,), ref['n2_w'], ref['n2_b'], 1e-5)
    want = y.view(M, S, E).mean(dim=1)
    (want * G.double()).sum().backward()

    assert rel_err(pooled.detach().cpu().numpy(), want.detach().numpy()) < TOL
    worst = ('', 0.0)
    for n in names:
        got, exp = dev[n].grad.cpu().double(), ref[n].grad
        floor = max(float(exp.norm()) / max(1.0, exp.numel()) ** 0.5, 1e-6)
        e = rel_err(got.numpy(), exp.numpy(), floor=floor)
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < TOL, '%s: %.3e' % (n, e)
    print('M=%d S=%d: worst gradient %s rel err %.2e' % (M, S, *worst))


def test_p_zero_is_the_fused_path(ops):
    """p = 0 through the dropout kernels (forced) equals the fused scoring-style forward."""
    M, S, E, nh = 4, 32, 300, 10
    ids = torch.randint(0, 100, (M, S), dtype=torch.int32).cuda()
    table, pe = rnd(100, E, seed=1).cuda(), rnd(S, E, seed=2).cuda()
    x_fused = ops.embed_pe(ids.reshape(-1), table, pe, S)
    x_drop = ops.embed_pe_dropout(ids.reshape(-1), table, pe, S, 0.0, 1, 0, 1)
    assert torch.equal(x_fused, x_drop)
    t, res = rnd(M * S, E, seed=3).cuda(), rnd(M * S, E, seed=4).cuda()
    gmm, bta = (rnd(E, seed=5) + 1).cuda(), rnd(E, seed=6).cuda()
    y, rstd = ops.dropout_add_layernorm(t, res, gmm, bta, 1e-5, 0.0, 1, 3)
    want = F.layer_norm((t + res).cpu().double(), (E,), gmm.cpu().double(), bta.cpu().double(), 1e-5)
    assert rel_err(y.cpu().numpy(), want.numpy()) < 2e-5


@pytest.mark.parametrize('case,probe', [('cfg1_crown', 'title_transformer.layers.0.linear1.weight'), ('cfg1_mhsa', 'multiheadAttention.W_V.weight')])
def test_model_trains_with_the_reference_dropout_rate(case, probe):
    """model.train() at dropout_rate = 0.2 (the reference's config.py:78), both content encoders: finite loss and gradients,
    repeatable under torch.manual_seed, different under another seed."""
    cfg, batch, c = golden_cases.build_case(case)
    cfg.dropout_rate = 0.2
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, golden_cases.WEIGHT_SEED)
    model = model.cuda().train()
    b = [v.cuda() for v in batch.values()]

    def run(seed):
        torch.manual_seed(seed)
        model.zero_grad()
        loss = T.negative_log_softmax(model(*b))
        loss.backward()
        return float(loss.detach()), dict(model.news_encoder.base_news_encoder.named_parameters())[probe].grad.clone()

    l1, g1 = run(11)
    l2, g2 = run(11)
    l3, g3 = run(12)
    assert math.isfinite(l1) and torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert l1 == l2 and torch.equal(g1, g2)
    assert l1 != l3 and not torch.equal(g1, g3)


def test_fused_dropout_passes_equal_the_separate_ones():
    """lime_layernorm_bwd_dropout_f32's second result and lime_dropout2_f32 are bit for bit what separate lime_dropout_f32 passes give."""
    from lime_cikm25_amd import ops
    g = torch.Generator().manual_seed(3)
    M, E, p, seed = 5000, 300, 0.2, 1234567
    dy = (torch.rand(M, E, generator=g) * 2 - 1).cuda()
    z = (torch.rand(M, E, generator=g) * 2 - 1).cuda()
    gamma, beta = (torch.rand(E, generator=g) + 0.5).cuda(), (torch.rand(E, generator=g) - 0.5).cuda()
    y = torch.nn.functional.layer_norm(z, (E,), gamma, beta, 1e-5)
    rstd = 1.0 / torch.sqrt(z.var(dim=1, unbiased=False) + 1e-5)
    dz, dg, db, dzs = ops.layernorm_bwd(dy, y, gamma, beta, rstd)
    dz2, dg2, db2, dzs2, dt = ops.layernorm_bwd(dy, y, gamma, beta, rstd, dropout=(p, seed, 5))
    assert torch.equal(dz, dz2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    assert torch.equal(dt, ops.dropout(dz, p, seed, 5))
    want = dt.double().sum(dim=0)                                                  # dzsum of the fused form: column sums of the DROPPED gradient
    assert float((dzs2.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    assert not torch.equal(dzs, dzs2)
    assert 0.15 < float((dt == 0).float().mean()) < 0.25
    x = (torch.rand(M, E, generator=g) * 2 - 1).cuda()
    assert torch.equal(ops.dropout2(x, p, seed, 1, 0), ops.dropout(ops.dropout(x, p, seed, 1), p, seed, 0))
