"""Per-kernel time of the encoder-layer GEMMs, split-product kernel vs fp32-MFMA kernel, on the live-row counts of a config-2b batch."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lime_cikm25_amd import ops, _lib

def rnd(*s, scale=1.0):
    return (torch.rand(*s, device='cuda') * 2 - 1) * scale

def time_it(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

V, E, F, S = 50000, 300, 512, 128
table = rnd(V, E, scale=0.3)
Ms = [int(x) for x in (sys.argv[1:] or [62208, 16512, 225280])]
for M in Ms:
    ids = torch.randint(0, V, (M,), device='cuda', dtype=torch.int32)
    cids = torch.randperm(M, device='cuda').to(torch.int32)
    w_in, pew = rnd(960, E, scale=0.05), rnd(S, 960)
    w_o, b_o, pe = rnd(E, E, scale=0.05), rnd(E), rnd(S, E)
    w1, b1, w2, b2 = rnd(F, E, scale=0.05), rnd(F), rnd(E, F, scale=0.05), rnd(E)
    g, be = rnd(E) + 1.5, rnd(E)
    attn, x1, h = rnd(M, E), rnd(M, E), rnd(M, F)
    qkv = torch.empty(M, 960, device='cuda'); o300 = torch.empty(M, E, device='cuda'); o512 = torch.empty(M, F, device='cuda'); pool = torch.empty(M // 32, E, device='cuda')
    Mp = M // 32 * 32
    cases = [
        ('in_proj cid', lambda: ops.linear(table, w_in, None, a_ids=ids, res=pew, res_mod=S, out=qkv, c_ids=cids), 2 * M * 900 * 300),
        ('in_proj dense', lambda: ops.linear(table, w_in, None, a_ids=ids, res=pew, res_mod=S, out=qkv), 2 * M * 900 * 300),
        ('plain 960', lambda: ops.linear(attn, w_in, None, out=qkv), 2 * M * 900 * 300),
        ('out_proj', lambda: ops.linear(attn, w_o, b_o, res=table, res_ids=ids, res_pe=pe, res_period=S, ln=(g, be), out=o300), 2 * M * 300 * 300),
        ('linear1', lambda: ops.linear(x1, w1, b1, act='relu', out=o512), 2 * M * 512 * 300),
        ('linear2 pool', lambda: ops.linear(h[:Mp], w2, b2, res=x1[:Mp], ln=(g, be), pool32=True, out=pool[:Mp // 32]), 2 * Mp * 300 * 512),
        ('linear2', lambda: ops.linear(h, w2, b2, res=x1, ln=(g, be), out=o300), 2 * M * 300 * 512),
        ('plain 300', lambda: ops.linear(attn, w_o, None, out=o300), 2 * M * 300 * 300),
    ]
    print('M = %d' % M)
    for name, fn, fl in cases:
        row = []
        for split in (False, True):
            ops.set_split_gemm(split)
            t = time_it(fn)
            row.append((t, fl / t / 1e6, _lib.load().lime_last_linear_kernel().decode()[:40]))
        print('  %-14s fp32 %7.1f us %6.1f TF | split %7.1f us %6.1f TF  (%s)' % (name, row[0][0], row[0][1], row[1][0], row[1][1], row[1][2]))
