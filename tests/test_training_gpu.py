"""The training step on the MI355X against gradients captured from the imported reference
(tests/golden/grad_*.npz, tools/make_grad_goldens.py): loss, every parameter gradient, which parameters get none (Q20), then
the native clip + Adam step against torch.optim.Adam / clip_grad_norm_ driven by the same gradients.  Tolerance 1e-3
relative (north star); observed errors are printed."""
import json
import os

import numpy as np
import pytest
import torch

import golden_cases
from helpers import load_golden, rel_err
from lime_cikm25_amd import Model, synth
from lime_cikm25_amd.training import TrainStep, negative_log_softmax

pytestmark = pytest.mark.gpu
TOL = 1e-3
LR = 1e-5
GRAD_CASES = ['cfg1_crown', 'cfg1_mhsa', 'spill', 'empty_history', 'full_len', 'long_body', 'fusion_gated', 'two_layers']


def train_model(name):
    cfg, batch, c = golden_cases.build_case(name)
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, golden_cases.WEIGHT_SEED)
    model = model.cuda()
    model.eval()
    model.training = True              # [B, K] shape, children in eval mode: what the goldens were captured with
    return cfg, model, [v.cuda() for v in batch.values()]


def unique_named_parameters(model):
    seen = set()
    for k, p in model.named_parameters():
        if id(p) not in seen:
            seen.add(id(p))
            yield k, p


def compare_grads(g, named):
    worst = ('', 0.0)
    for k in json.loads(str(g['with_grad'])):
        got = named[k].grad
        assert got is not None, '%s has no gradient' % k
        got = got.detach().cpu().double().reshape(-1)
        assert torch.isfinite(got).all(), k
        scale = float(g['norm:' + k]) / max(1.0, got.numel()) ** 0.5          # rms of the reference gradient
        if 'full:' + k in g.files_:
            want = g['full:' + k].reshape(-1)
            e = rel_err(got.numpy(), want, floor=max(scale, 1e-5))
        else:
            idx, want = g['idx:' + k], g['val:' + k]
            e = rel_err(got.numpy()[idx], want, floor=max(scale, 1e-5))
            e = max(e, abs(float(got.norm()) - float(g['norm:' + k])) / (float(g['norm:' + k]) + 1e-6))
        if e > worst[1]:
            worst = (k, e)
        assert e < TOL, '%s: gradient rel err %.3e' % (k, e)
    return worst


class _G(dict):
    files_ = ()


def golden(name):
    d = _G(load_golden('grad_' + name))
    d.files_ = set(d.keys())
    return d


@pytest.mark.parametrize('name', GRAD_CASES)
def test_gradients_match_the_reference(name):
    g = golden(name)
    cfg, model, batch = train_model(name)
    logits = model(*batch)
    assert logits.requires_grad
    assert rel_err(logits.detach().cpu().numpy(), g['logits']) < TOL
    loss = negative_log_softmax(logits)
    assert abs(float(loss.detach()) - float(g['loss'])) < TOL * max(1.0, abs(float(g['loss'])))
    loss.backward()
    named = dict(unique_named_parameters(model))
    for k in json.loads(str(g['without_grad'])):
        assert named[k].grad is None, '%s: the reference leaves this gradient at None (SURVEY Q20)' % k
    worst = compare_grads(g, named)
    print('%s: loss %.6f (reference %.6f), worst gradient %s rel err %.2e' % (name, float(loss.detach()), float(g['loss']), *worst))


def test_native_step_follows_torch_adam_and_clip():
    """TrainStep (flat buckets, HIP clip + Adam) against clip_grad_norm_ + torch.optim.Adam fed with the SAME gradients,
    three steps; then the loss must have moved the way the torch-driven copy's did."""
    name = 'cfg1_crown'
    g = golden(name)
    cfg, model, batch = train_model(name)
    _, twin, _ = train_model(name)
    step = TrainStep(model, lr=LR, gradient_clip_norm=4.0)
    assert sorted(step.names) == sorted(json.loads(str(g['with_grad'])))
    params = [p for k, p in unique_named_parameters(twin) if k in set(step.names)]
    opt = torch.optim.Adam(params, lr=LR)
    losses = []
    for it in range(3):
        loss = step.step(*batch)
        losses.append(float(loss))
        # the twin: our backward, torch's optimizer
        opt.zero_grad()
        tl = negative_log_softmax(twin(*batch))
        tl.backward()
        norm = torch.nn.utils.clip_grad_norm_(params, 4.0)
        opt.step()
        assert abs(float(tl.detach()) - losses[-1]) < 1e-4 * max(1.0, abs(losses[-1])), (it, float(tl.detach()), losses[-1])
        assert abs(float(norm) - float(step.last_norm)) < 1e-4 * float(norm)
    named, tnamed = dict(unique_named_parameters(model)), dict(unique_named_parameters(twin))
    for k in step.names:
        a, b = named[k].detach().cpu().double().reshape(-1), tnamed[k].detach().cpu().double().reshape(-1)
        d = (a - b).abs()
        # Adam divides by sqrt(v): an element whose gradient is at rounding-noise level (|g| ~ eps = 1e-8; the word-table
        # scatter-add's atomic order differs run to run) can move by up to lr per step in either direction, so the bound is
        # "the typical element agrees to 1e-4 of the parameter scale, none differs by more than the three steps could move it"
        # (the optimizer arithmetic itself is pinned element-wise on identical gradients in test_backward_gpu.py)
        scale = float(b.abs().mean()) + 1e-12
        assert float(d.max()) <= 3 * LR * 1.01, '%s: max |diff| %.3e' % (k, float(d.max()))
        q = float(torch.quantile(d[:1 << 20], 0.5))
        assert q <= 1e-4 * scale + 1e-7, '%s after 3 steps: median |diff| %.3e (scale %.3e)' % (k, q, scale)
    assert abs(losses[0] - float(g['loss'])) < TOL * max(1.0, abs(float(g['loss'])))
    assert losses[2] < losses[0], 'three Adam steps on one batch must lower its loss: %s' % losses
    # scoring still works on the updated (re-pointed) parameters, and agrees with the training forward
    with torch.no_grad():
        scored = model(*batch)
    again = model(*batch)
    assert rel_err(scored.cpu().numpy(), again.detach().cpu().numpy()) < 1e-4


def test_mixed_dropout_modes_are_refused():
    """One probability for the six dropout sites of a token encoder: a half-train / half-eval news encoder is refused."""
    cfg, batch, c = golden_cases.build_case('cfg1_crown')
    cfg.dropout_rate = 0.2
    model = Model(cfg)
    model.initialize()
    model = model.cuda().train()
    model.news_encoder.base_news_encoder.title_transformer.eval()
    with pytest.raises(NotImplementedError):
        model(*[v.cuda() for v in batch.values()])


def test_checkpoint_resume_continues_bit_for_bit(tmp_path):
    """save_checkpoint after two steps, load into a fresh model + TrainStep, take the third step on both: identical loss and
    parameters (Adam moments, step count and the re-pointed flat bucket all survive).  The file is the reference's
    {model_name: state_dict} (trainer.py:220) with one extra key."""
    from lime_cikm25_amd.training import load_checkpoint, save_checkpoint
    cfg, model, batch = train_model('spill')
    step = TrainStep(model, lr=LR)
    for _ in range(2):
        step.step(*batch)
    path = str(tmp_path / 'ckpt')
    save_checkpoint(path, model, step)
    payload = torch.load(path, weights_only=True)
    assert set(payload) == {model.model_name, 'optimizer'} and set(payload[model.model_name]) == set(model.state_dict())
    _, fresh, _ = train_model('spill')
    fresh_step = TrainStep(fresh, lr=123.0)                       # hyper-parameters come back from the file
    load_checkpoint(path, fresh, fresh_step)
    assert fresh_step.step_count == 2 and fresh_step.lr == LR
    # the scatter-add of the word-table gradient uses float atomics: compare everything else exactly, the table closely
    l_a, l_b = float(step.step(*batch)), float(fresh_step.step(*batch))
    assert abs(l_a - l_b) <= 1e-6 * max(1.0, abs(l_a))
    a, b = dict(unique_named_parameters(model)), dict(unique_named_parameters(fresh))
    for k in step.names:
        assert rel_err(b[k].detach().cpu().numpy(), a[k].detach().cpu().numpy()) < 1e-5, k


def test_detached_bucket_is_detected():
    cfg, model, batch = train_model('spill')
    step = TrainStep(model, lr=LR)
    step.step(*batch)
    model.zero_grad()                      # set_to_none: the .grad views are gone
    with pytest.raises(RuntimeError, match='flat bucket'):
        step.step(*batch)


def test_scoring_graph_captured_before_training_is_not_reused():
    """A scoring HIP graph captured before TrainStep moved the parameters into its bucket must not be replayed afterwards
    (it would read the old, now stale, parameter storage): scoring after training equals a fresh eager evaluation."""
    cfg, model, batch = train_model('cfg1_crown')
    with torch.no_grad():
        before = model(*batch).clone()                     # captures the scoring graph
    step = TrainStep(model, lr=1e-3)
    for _ in range(2):
        step.step(*batch)
    with torch.no_grad():
        after = model(*batch).clone()
        model.use_graph = False
        eager = model(*batch).clone()
    assert rel_err(after.cpu().numpy(), eager.cpu().numpy()) < 1e-5
    assert not torch.equal(after, before), 'two Adam steps at lr 1e-3 must change the scores'


def test_full_size_step_against_the_oracle_autograd():
    """BASELINE config 2b at full size (batch 32, history 50, title 32 + body 128, K = 1+4; 281,600 tokens): loss and gradients of
    the HIP training forward + backward against the CPU oracle's autograd (one forward + backward of the same batch)."""
    from lime_cikm25_amd import make_config
    from oracle import lime_oracle as O
    cfg = make_config()
    model = Model(cfg)
    model.initialize()
    synth.fill_state_dict(model, seed=1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    batch = synth.make_batch(cfg, 32, 5, seed=100)
    probes = ['news_encoder.base_news_encoder.body_transformer.layers.0.self_attn.in_proj_weight',
              'news_encoder.base_news_encoder.title_transformer.layers.0.linear2.weight',
              'news_encoder.base_news_encoder.body_transformer.layers.0.norm1.weight',
              'news_encoder.base_news_encoder.intent_layers.1.weight', 'news_encoder.project.weight',
              'user_encoder.graph_sage.convs.0.lin_r.weight', 'user_encoder.candidate_aware_attn.gate_proj.weight']
    for k in probes:
        sd[k].requires_grad_(True)
    for k in list(sd):
        if k.startswith('user_encoder.news_encoder.'):
            sd[k] = sd[k[len('user_encoder.'):]]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = (-torch.log_softmax(O.model_forward(sd, cfg, batch, grad=True), dim=1).select(dim=1, index=0)).mean()
    ref.backward()
    model = model.cuda()
    model.eval()
    model.training = True
    loss = negative_log_softmax(model(*[v.cuda() for v in batch.values()]))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < TOL * max(1.0, abs(float(ref.detach())))
    named = dict(model.named_parameters())
    for k in probes:
        want, got = sd[k].grad, named[k].grad.cpu()
        floor = max(float(want.norm()) / want.numel() ** 0.5, 1e-5)
        e = rel_err(got.numpy(), want.numpy(), floor=floor)
        assert e < TOL, '%s: %.3e' % (k, e)


def test_bf16_scoring_models_do_not_train_silently_in_fp32():
    cfg, batch, c = golden_cases.build_case('cfg1_crown')
    cfg.compute_dtype = 'bf16'
    model = Model(cfg)
    model.initialize()
    model = model.cuda()
    model.training = True
    with pytest.raises(NotImplementedError, match='compute_dtype'):
        model(*[v.cuda() for v in batch.values()])
