"""Isolated timings of the training-step kernels at the config-2b body shape (1760 news x 128 tokens), for rocprofv3
(--kernel-trace / --pmc) and for quick A/B runs:

    python tools/bench_train_kernels.py [--iters 5]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).to(dev)
    M, S, E, F, nh, hd, hs = 1760, 128, 300, 512, 10, 30, 32
    tok, W = M * S, nh * hs
    dqkv, x0, dz, hbuf, x1, ao = rnd(tok, 3 * W), rnd(tok, E), rnd(tok, E), rnd(tok, F).relu_(), rnd(tok, E), rnd(tok, E)
    gamma, beta, rstd = rnd(E) + 1, rnd(E), rnd(tok).abs() + 1
    qkv = rnd(tok, 3 * W)
    qkv.view(tok, 3 * nh, hs)[:, :, hd:] = 0
    dao, dpool = rnd(tok, E), rnd(M, E)
    ids = torch.randint(0, 50000, (tok,), generator=g, dtype=torch.int32).to(dev)
    ids[torch.rand(tok, generator=g).to(dev) < 0.45] = 0
    table_g = torch.zeros(50000, E, device=dev)
    cases = {
        'wgrad in_proj  dW[960,300] = dqkv^T x0': (lambda: ops.linear_wgrad(dqkv, x0, want_bias=True), 2.0 * tok * 960 * 300),
        'wgrad linear1  dW[512,300] = dh^T x1': (lambda: ops.linear_wgrad(hbuf, x1, want_bias=True), 2.0 * tok * 512 * 300),
        'wgrad linear2  dW[300,512] = dz^T h': (lambda: ops.linear_wgrad(dz, hbuf), 2.0 * tok * 300 * 512),
        'wgrad out_proj dW[300,300] = dz^T ao': (lambda: ops.linear_wgrad(dz, ao), 2.0 * tok * 300 * 300),
        'layernorm_bwd (mean pool folded in)': (lambda: ops.layernorm_bwd(dpool, x1, gamma, beta, rstd, dy_div=S, dy_scale=1.0 / S), 0.0),
        'layernorm_bwd': (lambda: ops.layernorm_bwd(dz, x1, gamma, beta, rstd), 0.0),
        'token_attention_bwd S=128': (lambda: ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dao, M, S, nh, hd,
                                                                      1.0 / math.sqrt(hd), head_stride=hs), 5 * 2.0 * M * nh * S * S * 32),
        'relu_bwd': (lambda: ops.relu_bwd_(hbuf.clone(), hbuf), 0.0),
        'embed_bwd (45 % padding word)': (lambda: ops.embed_bwd(ids, dz, table_g, hot_id=0), 0.0),
    }
    for name, (fn, flops) in cases.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print('%-42s %8.1f us%s' % (name, us, '   %6.1f TFLOP/s' % (flops / us / 1e6) if flops else ''), flush=True)


if __name__ == '__main__':
    main()
