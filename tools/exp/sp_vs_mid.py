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
for M in (14080, 16300, 28160, 42240, 102400, 153600, 815000):
    a4, a9, a352 = rnd(M, 400), rnd(M, 900), rnd(M, 352)
    w44, w49, wint = rnd(400, 400, scale=.05), rnd(400, 900, scale=.05), rnd(1200, 352, scale=.05)
    b4 = rnd(400)
    table = rnd(100, 400); ids = torch.randint(0, 100, (M,), device='cuda', dtype=torch.int32)
    l = rnd(M // 50 + 1, 400)
    cases = [('affine1 tanh 400x400', lambda: ops.linear(a4, w44, b4, act='tanh'), 2 * M * 400 * 400),
             ('project K900 + gathered res', lambda: ops.linear(a9, w49, None, res=table, res_ids=ids), 2 * M * 400 * 900),
             ('lin_r + broadcast res', lambda: ops.linear(a4, w44, None, res=l, res_div=50), 2 * M * 400 * 400),
             ('K plain 400x400', lambda: ops.linear(a4, w44, None), 2 * M * 400 * 400),
             ('intents relu 1200x352', lambda: ops.linear(a352, wint, rnd(1200), act='relu'), 2 * M * 1200 * 352)]
    print('M = %d' % M)
    for name, fn, fl in cases:
        row = []
        for split in (False, True):
            ops.set_split_gemm(split)
            us = t(fn)
            row.append((us, fl / us / 1e6, _lib.load().lime_last_linear_kernel().decode()[:34]))
        print('  %-28s off %8.1f us %6.1f TF (%s) | on %8.1f us %6.1f TF (%s)' % (name, row[0][0], row[0][1], row[0][2], row[1][0], row[1][1], row[1][2]))
    if M >= 400000: break
