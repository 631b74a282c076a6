"""Where the 256-row-tile split kernel starts to beat the 64-row-tile kernel: affine1-like (tanh) and plain GEMMs at several fills.
    python tools/exp/sp_fill.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lime_cikm25_amd import ops, _lib
def rnd(*s, scale=1.0): return (torch.rand(*s, device='cuda') * 2 - 1) * scale
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, act in ((42240, 200, 400, 'tanh'), (28160, 200, 400, 'tanh'), (56320, 200, 400, 'tanh'), (14080, 400, 400, None), (21000, 400, 400, None),
                     (25000, 400, 400, None), (14080, 400, 900, None), (33000, 200, 400, 'tanh'), (36000, 256, 400, 'tanh'), (20000, 512, 300, None)):
    a, w, b = rnd(M, K), rnd(N, K, scale=.05), rnd(N)
    fn = lambda: ops.linear(a, w, b, act=act)
    row = []
    for mode in ('off', 'rule', 'force'):
        if mode == 'off': ops.set_split_gemm(False)
        elif mode == 'rule': ops.set_split_gemm(True)
        else: ops.set_split_gemm(True, force=True)
        row.append((t(fn), _lib.load().lime_last_linear_kernel().decode()[:22]))
    tiles = (M + 255) // 256
    print('M %6d N %4d K %4d %-5s | off %7.1f us (%s) | rule %7.1f us (%s) | force %7.1f us (%s)' % (M, N, K, act, row[0][0], row[0][1], row[1][0], row[1][1], row[2][0], row[2][1]), flush=True)
ops.set_split_gemm(True)
