import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lime_cikm25_amd import ops
def rnd(*s): return torch.rand(*s, device='cuda') * 2 - 1
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, N, H, D = 32, 5, 50, 400
qp, kp = rnd(B * N, D), rnd(B * H, D)
mask = torch.ones(B, H, dtype=torch.bool, device='cuda')
for by in (True, False):
    ops.CAND_ATTN_BY_HEAD = by
    print('cand_attn by_head=%s: %.1f us' % (by, t(lambda: ops.cand_attn_weights(qp, kp, mask, B, N, H, D, 10))))
x, y = rnd(B * H, D), rnd(B * H, D)
sc, bias, g, be, un = torch.rand(B * H, device='cuda'), rnd(D), rnd(D), rnd(D), rnd(64, D)
print('gate_ln_sage fused: %.1f us' % t(lambda: ops.gate_ln_sage(y, x, sc, bias, g, be, 1e-5, B, H, D, 1, 32, None)))
print('gate_ln + sage_mean: %.1f us' % t(lambda: ops.sage_mean(ops.gate_ln(y, x, sc, bias, g, be).view(B * H, D), un, B, H, 32, D)))
