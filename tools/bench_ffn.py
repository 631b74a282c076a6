"""Time lime_encoder_ffn_bf16 against the two lime_linear_bf16 launches it replaces (body-chunk shape of BASELINE config 3).

    python tools/bench_ffn.py [rows]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 360448
    E, EP, F = 300, 304, 512
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).cuda()
    x = torch.zeros(M, EP, dtype=torch.bfloat16, device='cuda')
    x[:, :E] = rnd(M, E).to(torch.bfloat16)
    w1, b1, w2, b2 = rnd(F, E), rnd(F), rnd(E, F), rnd(E)
    gam, bet = rnd(E) + 1, rnd(E)
    padv = lambda v: torch.cat([v, v.new_zeros(EP - E)])
    w1b, w2b = ops.to_bf16(w1, cols_out=EP), ops.to_bf16(w2, rows_out=EP)
    w1p, w2p = ops.ffn_pack_bf16(w1, b1, w2)
    ln = (gam, bet)
    lnp = (padv(gam), padv(bet))
    b2p = padv(b2)

    def two():
        h = ops.linear_bf16(x, w1b, b1, act='relu', k_alg=E)
        return ops.linear_bf16(h, w2b, b2p, res=x, res_kind=3, ln=lnp, ln_eps=1e-5, ln_count=E, pool32=True, n_alg=E)

    def fused():
        return ops.encoder_ffn_bf16(x, w1p, w2p, b2, ln, 1e-5, E, pool32=True)

    V, S = 60000, 128
    table = torch.zeros(V, EP, dtype=torch.bfloat16, device='cuda')
    table[:, :E] = rnd(V, E).to(torch.bfloat16)
    ids = torch.randint(0, V, (M,), generator=g, dtype=torch.int32).cuda()
    pe_p = torch.zeros(S, EP, device='cuda')
    pe_p[:, :E] = rnd(S, E)
    w0, b0, g1, be1 = rnd(E, E), rnd(E), rnd(E) + 1, rnd(E)
    w0b, w0p = ops.to_bf16(w0, rows_out=EP, cols_out=EP), ops.oproj_pack_bf16(w0)
    add = pe_p[:, :E] + b0

    def oproj():
        return ops.linear_bf16(x, w0b, padv(b0), res=table, res_kind=2, res_ids=ids, res_pe=pe_p, res_period=S, ln=(padv(g1), padv(be1)),
                               ln_eps=1e-5, ln_count=E, n_alg=E, k_alg=E)

    def block():
        return ops.encoder_block_bf16(x, w0p, add, (g1, be1), 1e-5, res=table, res_kind=2, res_ids=ids, w1p=w1p, w2p=w2p,
                                      b2=b2, ln2=ln, ln2_eps=1e-5, E=E, pool32=True)

    x1 = oproj()
    ref = ops.encoder_ffn_bf16(x1, w1p, w2p, b2, ln, 1e-5, E, pool32=True)
    print('max |out_proj + ffn - block| = %.3e' % (ref - block()).abs().max().item())
    print('out_proj alone %.1f us   block %.1f us' % (timed(oproj), timed(block)))
    # in_proj: 220k live tokens scattered into a 360k-row qkv buffer
    Ml, N = 220160, 960
    w_in = rnd(N, E)
    pew = rnd(S, N)
    tok = torch.randint(0, V, (Ml,), generator=g, dtype=torch.int32).cuda()
    rows = torch.sort(torch.randperm(M, generator=g)[:Ml]).values.to(torch.int32).cuda()
    qkv = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
    w_in_b, w_in_p = ops.to_bf16(w_in, cols_out=EP), ops.inproj_pack_bf16(w_in, EP)
    old_in = lambda: ops.linear_bf16(table, w_in_b, None, a_ids=tok, res=pew, res_kind=1, res_mod=S, out=qkv, c_ids=rows, n_alg=3 * E, k_alg=E)
    new_in = lambda: ops.inproj_bf16(table, w_in_p, pew, N, qkv, a_ids=tok, c_ids=rows)
    old_in()
    ref_q = qkv.clone()
    qkv.zero_()
    new_in()
    print('in_proj max |old - new| = %.3e' % (ref_q.float() - qkv.float()).abs().max().item())
    t_o, t_n = timed(old_in), timed(new_in)
    fl = 2.0 * Ml * 900 * E
    print('in_proj rows %d   gemm_pp %.1f us (%.0f TF)   activation-stationary %.1f us (%.0f TF = %.3f of 2.5 PF)' %
          (Ml, t_o, fl / t_o / 1e6, t_n, fl / t_n / 1e6, fl / t_n / 1e6 / 2500))
    a, b = two(), fused()
    print('max |two - fused| = %.3e (mean |two| %.3e)' % ((a - b).abs().max().item(), a.abs().mean().item()))
    t2, tf = timed(two), timed(fused)
    flops = 2.0 * M * (E * F * 2)
    print('rows %d   two launches %.1f us (%.0f TF)   fused %.1f us (%.0f TF = %.3f of 2.5 PF)' %
          (M, t2, flops / t2 / 1e6, tf, flops / tf / 1e6, flops / tf / 1e6 / 2500))


if __name__ == '__main__':
    main()
