"""End-to-end parity on the MI355X: the drop-in Model (HIP kernels through the C ABI) against the
CPU oracle and against the golden vectors captured from the reference, on every golden case, plus
full-size (BASELINE.json configs[1]) checks: direct comparison with the oracle, row-permutation
equivariance, determinism, and ranking / AUC agreement."""
import json

import numpy as np
import pytest
import torch

import golden_cases
from helpers import load_golden, rel_err, synth_state_dict
from lime_cikm25_amd import Model, make_config, synth
from oracle import lime_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 fp32 relative tolerance"


def gpu_model(cfg, seed=golden_cases.WEIGHT_SEED):
    m = Model(cfg)
    m.initialize()
    synth.fill_state_dict(m, seed)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda(), sd


def run(model, batch, eval_shape):
    model.eval()
    if not eval_shape:
        model.training = True          # [B, K] inputs; children stay in eval mode (no dropout)
    with torch.no_grad():              # scoring (with grad mode on, training = True selects the differentiable path)
        out = model(*[v.cuda() for v in batch.values()])
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize('name', list(golden_cases.CASES))
def test_golden_case(name):
    cfg, batch, c = golden_cases.build_case(name)
    g = load_golden(name)
    model, sd = gpu_model(cfg)
    # the drop-in's state_dict is the reference's, key for key and shape for shape
    assert [[k, list(v.shape)] for k, v in sd.items()] == json.loads(str(g['state_dict_spec']))
    logits = run(model, batch, c['eval_shape'])
    want = O.model_forward(sd, cfg, batch, eval_shape=c['eval_shape'])
    assert logits.shape == want.shape == g['logits'].shape
    e_oracle, e_ref = rel_err(logits.numpy(), want.numpy()), rel_err(logits.numpy(), g['logits'])
    print('%s: vs oracle %.2e, vs reference golden %.2e' % (name, e_oracle, e_ref))
    assert e_oracle < TOL and e_ref < TOL
    z = g['logits'] == 0                                       # saturated lifetime weight: exact +-0
    assert np.all(logits.numpy()[z] == 0) and np.array_equal(np.signbit(logits.numpy()[z]), np.signbit(g['logits'][z]))


@pytest.mark.parametrize('name', ['cfg1_crown', 'cfg1_mhsa', 'bucket_edges', 'spill', 'empty_history'])
def test_stages_against_golden(name):
    """Sub-module forwards (the reference's own module API) against the per-stage goldens."""
    cfg, batch, c = golden_cases.build_case(name)
    g = load_golden(name)
    model, sd = gpu_model(cfg)
    model.eval()
    b = {k: v.cuda() for k, v in batch.items()}
    ne, ue = model.news_encoder, model.user_encoder
    cand = ne(b['news_title_text'], b['news_title_mask'], b['news_title_entity'], b['news_content_text'], b['news_content_mask'],
              b['news_content_entity'], b['news_category'], b['news_subCategory'], None, b['news_freshness'],
              b['news_user_topic_lifetime'])
    assert rel_err(cand.cpu().numpy(), g['news_representation']) < TOL
    content = ne.base_news_encoder(b['news_title_text'], b['news_title_mask'], b['news_title_entity'], b['news_content_text'],
                                   b['news_content_mask'], b['news_content_entity'], b['news_category'], b['news_subCategory'],
                                   None, b['news_freshness'], b['news_user_topic_lifetime'])
    assert rel_err(content.cpu().numpy(), g['cand_content']) < TOL
    fresh = ne.freshness_encoder(b['news_freshness'], b['news_user_topic_lifetime'])
    assert rel_err(fresh.cpu().numpy(), g['cand_freshness']) < TOL
    # integer buckets: bit-exact against the reference's own bucketize
    fe = ne.freshness_encoder
    assert np.array_equal(fe.bucketize(b['news_freshness']).cpu().numpy(), g['cand_f_bucket'])
    assert np.array_equal(fe.bucketize(b['news_user_topic_lifetime']).cpu().numpy(), g['cand_l_bucket'])
    assert np.array_equal(fe.bucketize(b['user_freshness']).cpu().numpy(), g['hist_f_bucket'])
    assert np.array_equal(fe.bucketize(b['user_user_topic_lifetime']).cpu().numpy(), g['hist_l_bucket'])
    user = ue(b['user_title_text'], b['user_title_mask'], b['user_title_entity'], b['user_content_text'], b['user_content_mask'],
              b['user_content_entity'], b['news_category'], b['news_subCategory'], b['user_category'], b['user_subCategory'],
              b['user_history_mask'], b['user_history_graph'], b['user_history_category_mask'],
              b['user_history_category_indices'], None, cand, b['user_freshness'], b['user_user_topic_lifetime'])
    assert rel_err(user.cpu().numpy(), g['user_representation']) < TOL
    logits = model.remaining_lifetime_weighting(user, cand, b['remaining_lifetime'])
    assert rel_err(logits.cpu().numpy(), g['logits']) < TOL
    # candidate-aware attention stage
    r = g['hist_news_out'].shape[0]
    hist = ne(b['user_title_text'], b['user_title_mask'], b['user_title_entity'], b['user_content_text'], b['user_content_mask'],
              b['user_content_entity'], b['user_category'], b['user_subCategory'], None, b['user_freshness'],
              b['user_user_topic_lifetime'])
    assert rel_err(hist.cpu().numpy()[:r], g['hist_news_out']) < TOL
    refined, agg = ue.candidate_aware_attn(hist, ue._topic(b['user_category'], b['user_subCategory']),
                                           ue._topic(b['news_category'], b['news_subCategory']), mask=b['user_history_mask'])
    assert rel_err(agg.cpu().numpy(), g['attn_weights_agg']) < TOL
    assert rel_err(refined.cpu().numpy()[:r], g['hist_refined']) < TOL
    gcn = ue.graph_sage.forward_closed_form(refined, ue.user_node_embedding, n_src=hist.shape[0])
    assert rel_err(gcn.cpu().numpy()[:r], g['gcn_feature']) < TOL


def test_reference_checkpoint_round_trip():
    """A state_dict in the reference's layout loads strictly and drives the same logits."""
    cfg, batch, c = golden_cases.build_case('cfg1_crown')
    g = load_golden('cfg1_crown')
    spec = json.loads(str(g['state_dict_spec']))
    sd = synth_state_dict([(k, s) for k, s in spec if not k.endswith('.pe')])
    model = Model(cfg)
    missing = model.load_state_dict(sd, strict=False)
    assert all(k.endswith('.pe') for k in missing.missing_keys) and not missing.unexpected_keys
    logits = run(model.cuda(), batch, False)
    assert rel_err(logits.numpy(), g['logits']) < TOL


@pytest.fixture(scope='module')
def full_size():
    """BASELINE.json configs[1]: batch 32, history 50, title 32 (+ body 128), K = 1+4, 300-d, fp32."""
    cfg = make_config(vocabulary_size=50000)
    model, sd = gpu_model(cfg, seed=21)
    batch = synth.make_batch(cfg, 32, 5, seed=22)
    logits = run(model, batch, False)
    return cfg, model, sd, batch, logits


def test_full_size_against_oracle(full_size):
    cfg, model, sd, batch, logits = full_size
    torch.set_num_threads(max(1, torch.get_num_threads()))
    want = O.model_forward(sd, cfg, batch)
    e = rel_err(logits.numpy(), want.numpy())
    print('config 2 (B=32,H=50,T=32,L=128,K=5): max rel err vs oracle %.2e' % e)
    assert e < TOL
    # ranking inside every impression agrees wherever the oracle's scores are separated by more than the tolerance
    for r in range(want.shape[0]):
        o = want[r].numpy()
        if np.min(np.abs(o[:, None] - o[None, :])[~np.eye(len(o), dtype=bool)]) > 1e-3 * (np.abs(o).max() + 1e-6):
            assert np.array_equal(np.argsort(-logits[r].numpy(), kind='stable'), np.argsort(-o, kind='stable'))


def test_full_size_deterministic(full_size):
    cfg, model, sd, batch, logits = full_size
    again = run(model, batch, False)
    assert torch.equal(again, logits)                 # no atomics anywhere: bitwise reproducible


def test_full_size_row_permutation_equivariance(full_size):
    """No op couples different impression rows except through the row count: permuting
    the rows permutes the logits bit for bit (every fma chain and reduction order is position independent)."""
    cfg, model, sd, batch, logits = full_size
    perm = torch.from_numpy(np.random.default_rng(0).permutation(32))
    pb = type(batch)((k, v[perm]) for k, v in batch.items())
    assert torch.equal(run(model, pb, False), logits[perm])


def test_full_size_shared_news_gets_one_embedding(full_size):
    """The same news as candidate 0 of row 0 and as history slot 0 of row 1 is encoded to the same vector."""
    cfg, model, sd, batch, _ = full_size
    b = type(batch)((k, v.clone()) for k, v in batch.items())
    for src, dst in (('news_title_text', 'user_title_text'), ('news_content_text', 'user_content_text'),
                     ('news_category', 'user_category'), ('news_subCategory', 'user_subCategory'),
                     ('news_freshness', 'user_freshness'), ('news_user_topic_lifetime', 'user_user_topic_lifetime')):
        b[dst][1, 0] = b[src][0, 0]
    model.eval()
    cb = {k: v.cuda() for k, v in b.items()}
    ne = model.news_encoder
    cand, hist = ne.encode_many([
        (cb['news_title_text'], cb['news_title_mask'], cb['news_content_text'], cb['news_category'], cb['news_subCategory'],
         cb['news_freshness'], cb['news_user_topic_lifetime']),
        (cb['user_title_text'], cb['user_title_mask'], cb['user_content_text'], cb['user_category'], cb['user_subCategory'],
         cb['user_freshness'], cb['user_user_topic_lifetime'])])
    assert torch.equal(cand[0, 0], hist[1, 0])


def test_auc_matches_oracle(full_size):
    """|dAUC| <= 0.001 between the HIP scores and the oracle's on identical synthetic data (north star)."""
    from sklearn.metrics import roc_auc_score
    cfg, model, sd, batch, logits = full_size
    want = O.model_forward(sd, cfg, batch).numpy()
    labels = np.zeros_like(want)
    labels[:, 0] = 1
    aucs = []
    for s in (logits.numpy(), want):
        # MIND-style: per impression, then averaged (evaluate.py:32-89 scores 1/rank; ties keep candidate order)
        a = []
        for r in range(s.shape[0]):
            order = np.argsort(-s[r], kind='stable')
            rank = np.empty_like(order)
            rank[order] = np.arange(1, len(order) + 1)
            a.append(roc_auc_score(labels[r], 1.0 / rank))
        aucs.append(float(np.mean(a)))
    print('AUC hip %.6f oracle %.6f' % tuple(aucs))
    assert abs(aucs[0] - aucs[1]) <= 1e-3


def test_bf16_path_against_oracle(full_size):
    """BASELINE config 3's arithmetic (bf16 MFMA operands / activations in the token encoders, fp32 accumulate, softmax,
    LayerNorm) on the config-2 batch, against the fp32 oracle.  bf16 keeps 8 mantissa bits (2^-9 = 2e-3 per rounding);
    through one encoder layer + pooling the logits land ~4e-3 (observed) from the fp32 ones: tolerance 1e-2 relative to the
    mean logit magnitude, and |dAUC| <= 0.01 (the north star's 0.001 is an fp32 statement).  The configs[2] batch size (256)
    is covered by tests/test_fullsize_gpu.py."""
    from sklearn.metrics import roc_auc_score
    cfg, model32, sd, batch, logits32 = full_size
    cfg16 = make_config(vocabulary_size=50000, compute_dtype='bf16')
    model = Model(cfg16)
    model.load_state_dict(sd)
    model = model.cuda()
    got = run(model, batch, False)
    want = O.model_forward(sd, cfg, batch)
    scale = float(want.abs().mean())
    err = float((got - want).abs().max()) / scale
    print('bf16 path: max |dlogit| / mean |logit| = %.3e (fp32 path: %.3e)' % (err, float((logits32 - want).abs().max()) / scale))
    assert torch.isfinite(got).all() and err < 1e-2
    assert torch.equal(run(model, batch, False), got)                       # still bitwise reproducible
    labels = np.zeros_like(want.numpy())
    labels[:, 0] = 1
    aucs = []
    for s_ in (got.numpy(), want.numpy()):
        aucs.append(float(np.mean([roc_auc_score(labels[r], s_[r]) for r in range(s_.shape[0])])))
    assert abs(aucs[0] - aucs[1]) <= 0.01


def test_score_impressions_equals_eval_forward_on_expanded_rows():
    """BASELINE config 5 layout: B impressions x K candidates scored with every history encoded once.  The result must be
    the reference's eval-mode function of the B * K (impression, candidate) rows (util.py:86-111): equal to this model's
    eval forward on the expanded rows, and within tolerance of the oracle's eval forward."""
    cfg = make_config(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=64, vocabulary_size=5000)
    model, sd = gpu_model(cfg, seed=41)
    B, K = 5, 6
    batch = synth.make_batch(cfg, B, K, seed=42)
    c = {k: v.cuda() for k, v in batch.items()}
    model.eval()
    got = model.score_impressions(c['user_category'], c['user_subCategory'], c['user_title_text'], c['user_title_mask'],
                                  c['user_content_text'], c['user_freshness'], c['user_user_topic_lifetime'], c['user_history_mask'],
                                  c['news_category'], c['news_subCategory'], c['news_title_text'], c['news_title_mask'],
                                  c['news_content_text'], c['news_freshness'], c['news_user_topic_lifetime'], c['remaining_lifetime'])
    assert got.shape == (B, K)
    # the reference's layout: one row per (impression, candidate), history repeated, candidate tensors without the N axis
    exp = type(batch)()
    for k, v in batch.items():
        if k.startswith('news_') or k == 'remaining_lifetime':
            exp[k] = v.reshape((B * K,) + tuple(v.shape[2:]))
        else:
            exp[k] = v.repeat_interleave(K, dim=0)
    model.use_graph = False
    ref_rows = run(model, exp, True)
    # (not bitwise: the two layouts cross the M >= 4096 threshold between the two GEMM kernels, whose k order differs)
    assert rel_err(got.cpu().reshape(-1).numpy(), ref_rows.reshape(-1).numpy()) < 2e-5
    want = O.model_forward(sd, cfg, exp, eval_shape=True).reshape(-1)
    assert rel_err(got.cpu().reshape(-1).numpy(), want.numpy()) < TOL
    # chunked passes give the same numbers
    again = model.score_impressions(c['user_category'], c['user_subCategory'], c['user_title_text'], c['user_title_mask'],
                                    c['user_content_text'], c['user_freshness'], c['user_user_topic_lifetime'], c['user_history_mask'],
                                    c['news_category'], c['news_subCategory'], c['news_title_text'], c['news_title_mask'],
                                    c['news_content_text'], c['news_freshness'], c['news_user_topic_lifetime'], c['remaining_lifetime'],
                                    rows_per_pass=2 * K)
    assert torch.equal(again, got)


def test_long_body_shape_against_oracle():
    """BASELINE config 4's sequence shape (Adressa: body length 512, title 32) in the forward: the S = 512 attention path
    (16 key tiles, K / V tiled inside one workgroup) and the chunked token passes, against the oracle."""
    from lime_cikm25_amd import newsEncoders
    cfg = make_config(max_history_num=5, max_abstract_length=512, batch_size=8, vocabulary_size=3000)
    model, sd = gpu_model(cfg, seed=51)
    batch = synth.make_batch(cfg, 3, 2, seed=52)
    want = O.model_forward(sd, cfg, batch)
    got = run(model, batch, False)
    assert rel_err(got.numpy(), want.numpy()) < TOL
    old = newsEncoders.MAX_TOKENS_PER_PASS
    try:                                                     # force several token passes per encoder (7 news per pass)
        newsEncoders.MAX_TOKENS_PER_PASS = 7 * 512
        model._graphs.clear()
        again = run(model, batch, False)
    finally:
        newsEncoders.MAX_TOKENS_PER_PASS = old
    assert rel_err(again.numpy(), want.numpy()) < TOL


def test_eval_harness_on_gpu(tmp_path):
    """compute_scores (util.py:77-129) with the HIP model on eval-shaped rows: the rank file equals the one built from
    the oracle's scores wherever those are separated beyond the tolerance, and the four metrics agree."""
    from lime_cikm25_amd import util as U
    cfg = make_config(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8, vocabulary_size=5000)
    model, sd = gpu_model(cfg, seed=31)
    rng = np.random.default_rng(5)
    sizes = rng.integers(2, 7, size=12)
    indices = np.repeat(np.arange(12), sizes).tolist()
    rows = len(indices)
    full = synth.make_batch(cfg, rows, 1, seed=32, eval_shape=True)
    batches, want = [], []
    for lo in range(0, rows, 16):
        b = [v[lo:lo + 16] for v in full.values()]
        want.append(O.model_forward(sd, cfg, b, eval_shape=True).squeeze(1))
        batches.append(b[:25])
    want = torch.cat(want).numpy()
    labels = [[1] + [0] * (n - 1) for n in sizes]
    truth = tmp_path / 'truth.txt'
    truth.write_text('\n'.join('%d %s' % (i + 1, str(l).replace(' ', '')) for i, l in enumerate(labels)))
    got = U.compute_scores(model, batches, indices, str(tmp_path / 'hip.txt'), str(truth))
    U.write_rank_file(str(tmp_path / 'cpu.txt'), U.rank_impressions(want, indices))
    from lime_cikm25_amd.evaluate import scoring
    with open(truth) as tf, open(tmp_path / 'cpu.txt') as rf:
        ref = scoring(tf, rf)
    assert np.allclose(got, ref, atol=1e-3)
    assert (tmp_path / 'hip.txt').read_text().count('\n') == 11


@pytest.mark.parametrize('nb', [7, 16])
def test_other_bucket_counts(nb):
    """config.num_buckets != 10 (config.py:59): the device compares against cut points derived from the rule's own fp32 evaluation
    -- bucket indices bit-exact with the rule on both sides of every cut, logits against the oracle."""
    from lime_cikm25_amd.newsEncoders import bucket_cut_points
    cfg = make_config(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8, vocabulary_size=5000, num_buckets=nb)
    model, sd = gpu_model(cfg, seed=43)
    cuts = bucket_cut_points(nb).numpy()
    bits = cuts.view(np.uint32)
    x = np.concatenate([(bits - 1).view(np.float32), cuts, (bits + 1).view(np.float32), np.array([0.0, 1.0, 86400.0, 3e38], np.float32)])
    fe = model.news_encoder.freshness_encoder
    got = fe.bucketize(torch.from_numpy(x).cuda()).cpu()
    assert torch.equal(got, O.bucketize(torch.from_numpy(x), nb))
    batch = synth.make_batch(cfg, 8, 2, seed=44)
    batch['news_freshness'].reshape(-1)[:len(cuts)] = torch.from_numpy(cuts)
    logits = run(model, batch, False)
    assert rel_err(logits.numpy(), O.model_forward(sd, cfg, batch).numpy()) < TOL
