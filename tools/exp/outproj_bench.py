"""out_proj (gathered residual + positional rows + LayerNorm [+ rstd]) on the fp32-MFMA kernel against the split-product kernel
(lime_set_split_gemm(3)).   python tools/exp/outproj_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lime_cikm25_amd import ops, _lib
lib = _lib.load()
def rnd(*s, scale=1.0): return (torch.rand(*s, device='cuda') * 2 - 1) * scale
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
V, E, S = 50000, 300, 128
table, pe = rnd(V, E), rnd(S, E)
w, b, g, be = rnd(E, E, scale=.06), rnd(E), rnd(E) + 1.5, rnd(E)
for M in (29440, 117760, 225280):
    a = rnd(M, E)
    ids = torch.randint(0, V, (M,), device='cuda', dtype=torch.int32)
    ids[torch.rand(M, device='cuda') < 0.5] = 0
    out = torch.empty(M, E, device='cuda')
    rstd = torch.empty(M, device='cuda')
    res = {}
    for mode, name in ((1, 'fp32 kernel'), (3, 'split kernel')):
        lib.lime_set_split_gemm(mode)
        for with_rstd in (False, True):
            fn = lambda: ops.linear(a, w, b, res=table, res_ids=ids, res_pe=pe, res_period=S, ln=(g, be), out=out, ln_rstd=rstd if with_rstd else None)
            us = t(fn)
            res[(mode, with_rstd)] = (us, out.clone(), lib.lime_last_linear_kernel().decode()[:40])
    d = (res[(1, False)][1] - res[(3, False)][1]).abs().max().item()
    print('M %6d  fp32 %7.1f us (%5.1f TF) / rstd %7.1f | split %7.1f us (%5.1f TF) / rstd %7.1f   max |diff| %.1e   %s' % (
        M, res[(1, False)][0], 2.0 * M * E * E / res[(1, False)][0] * 1e-6, res[(1, True)][0], res[(3, False)][0],
        2.0 * M * E * E / res[(3, False)][0] * 1e-6, res[(3, True)][0], d, res[(3, False)][2]), flush=True)
lib.lime_set_split_gemm(1)
