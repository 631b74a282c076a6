"""Micro-benchmark of the encoder-layer kernels at the config-2 shapes (run on the GPU box, optionally
under ``rocprofv3 --kernel-trace --stats`` or ``--pmc ...``).

    python tools/bench_kernels.py [--iters 10] [--only qkv_body,ffn2_body,...]
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
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--news', type=int, default=1760)
    args = ap.parse_args()
    only = set(x for x in args.only.split(',') if x)
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    V, E, F = 50000, 300, 512

    def rnd(*s, scale=1.0):
        return ((torch.rand(*s, generator=g) * 2 - 1) * scale).to(dev)

    table = rnd(V, E, scale=0.6)
    w_in, b_in = rnd(3 * E, E, scale=0.06), rnd(3 * E, scale=0.1)
    w_o, b_o = rnd(E, E, scale=0.06), rnd(E, scale=0.1)
    w1, b1 = rnd(F, E, scale=0.06), rnd(F, scale=0.1)
    w2, b2 = rnd(E, F, scale=0.05), rnd(E, scale=0.1)
    ln = (rnd(E) + 1.5, rnd(E, scale=0.1))
    results = []
    for name, S in (('title', 32), ('body', 128)):
        M = args.news
        tok = M * S
        ids = torch.randint(0, V, (tok,), generator=g, dtype=torch.int32).to(dev)
        pe = rnd(S, E)
        W = 320                              # heads padded to 32 columns, as newsEncoders.encode_tokens lays them out
        w_in_p = ops.pad_heads(w_in, 30, 30, 32)
        b_in_p = ops.pad_heads(b_in, 30, 30, 32)
        qkv = torch.empty(tok, 3 * W, device=dev)
        pew = ops.linear(pe, w_in_p, b_in_p)                 # the positional term of in_proj, [S, 3W]
        attn = torch.empty(tok, E, device=dev)
        x1 = torch.empty(tok, E, device=dev)
        h = torch.empty(tok, F, device=dev)
        x2 = torch.empty(tok, E, device=dev)
        blocks = torch.empty(tok // 32, E, device=dev)
        cases = [
            ('qkv_' + name, 2.0 * tok * 3 * E * E,
             lambda: ops.linear(table, w_in_p, None, a_ids=ids, res=pew, res_mod=S, out=qkv)),
            ('attn_' + name, 4.0 * tok * S * E,
             lambda: ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, 10, 30, 1 / math.sqrt(30), out=attn,
                                         head_stride=32)),
            ('out_' + name, 2.0 * tok * E * E,
             lambda: ops.linear(attn, w_o, b_o, res=table, res_ids=ids, res_pe=pe, res_period=S, ln=ln, out=x1)),
            ('ffn1_' + name, 2.0 * tok * F * E, lambda: ops.linear(x1, w1, b1, act='relu', out=h)),
            # linear2 as the model issues it: residual + LayerNorm + token means over 32-row blocks in the epilogue (pool32),
            # then (S > 32) the mean over a sequence's S / 32 block rows
            ('ffn2_' + name, 2.0 * tok * F * E, lambda: ops.linear(h, w2, b2, res=x1, ln=ln, pool32=True, out=blocks)),
            ('pool_' + name, 0.0, lambda: ops.mean_pool(blocks, M, S // 32)),
            # the stand-alone word gather + positional add (the fused GEMMs do not use it): HBM-bound, 2 x 1200 B per token
            ('embed_' + name, 0.0, lambda: ops.embed_pe(ids, table, pe, S, out=x2)),
        ]
        for cname, flops, fn in cases:
            if only and cname not in only:
                fn()            # keep the data flowing for the later stages
                continue
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            results.append((cname, us, flops / us / 1e6 if flops else 0.0))
    hbm = {'embed': 2400.0}                                            # algorithmic bytes per token (read + write)
    tokens = {'title': args.news * 32, 'body': args.news * 128}
    for cname, us, tf in results:
        kind, shape = cname.split('_')
        extra = '  %6.2f TB/s algorithmic' % (hbm[kind] * tokens[shape] / us / 1e6) if kind in hbm else ''
        print('%-12s %9.1f us  %7.2f TFLOP/s%s' % (cname, us, tf, extra))
    print('total %.1f us' % sum(r[1] for r in results))


if __name__ == '__main__':
    main()
