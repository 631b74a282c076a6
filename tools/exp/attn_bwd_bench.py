"""Token attention backward at S = 128 / 512.  S = 128: the one-pass kernels -- every product on the split product
(token_attn_bwd_sp_f32.hip) against the fp32-MFMA kernel, without and with probability dropout; S = 512: the blocked kernel without /
with the forward's statistics.   python tools/exp/attn_bwd_bench.py"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from lime_cikm25_amd import ops


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


h, hd = 10, 30
W = h * 32
for n_seq, S in ((1000, 128), (1760, 128), (3300, 128), (440, 512)):
    qkv = torch.randn(n_seq * S, 3 * W, device='cuda')
    qkv.view(-1, 3 * h, 32)[:, :, hd:] = 0
    dout = torch.randn(n_seq * S, h * hd, device='cuda')
    scale = 1.0 / math.sqrt(hd)
    lse = torch.empty(n_seq * S * h, device='cuda')
    out = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32, lse=lse)
    dqkv = torch.empty_like(qkv)
    run = lambda **kw: ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout, n_seq, S, h, hd, scale, head_stride=32, out=out, dqkv=dqkv, **kw)
    if S <= 128:
        res = {}
        for split in (True, False):
            prev = ops.set_split_gemm(split)
            a = timed(lambda: run())
            ref = dqkv.clone()
            b = timed(lambda: run(dropout=(0.2, 1234, 3)))
            res[split] = (a, b, ref, dqkv.clone())
            ops.set_split_gemm(prev)
        e0 = ((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max()).item()
        e1 = ((res[True][3] - res[False][3]).abs().max() / res[False][3].abs().max()).item()
        print('n_seq %5d S %4d   split %8.1f us (dropout %8.1f)   fp32 MFMA %8.1f us (dropout %8.1f)   rel diff %.1e / %.1e' %
              (n_seq, S, res[True][0], res[True][1], res[False][0], res[False][1], e0, e1), flush=True)
        continue
    a = timed(lambda: run())
    ref = dqkv.clone()
    b = timed(lambda: run(lse=lse))
    err = ((dqkv - ref).abs().max() / ref.abs().max()).item()
    print('n_seq %5d S %4d   plain %8.1f us   with lse %8.1f us   rel diff %.1e' % (n_seq, S, a, b, err), flush=True)
