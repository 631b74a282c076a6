"""Token attention backward at S = 128 / 512: the one-pass kernel (recomputes the softmax) against the blocked kernel fed with the
forward's statistics.   python tools/exp/attn_bwd_bench.py"""
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
for n_seq, S in ((1000, 128), (1760, 128), (440, 512)):
    qkv = torch.randn(n_seq * S, 3 * W, device='cuda')
    qkv.view(-1, 3 * h, 32)[:, :, hd:] = 0
    dout = torch.randn(n_seq * S, h * hd, device='cuda')
    scale = 1.0 / math.sqrt(hd)
    lse = torch.empty(n_seq * S * h, device='cuda')
    out = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32, lse=lse)
    dqkv = torch.empty_like(qkv)
    a = timed(lambda: ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout, n_seq, S, h, hd, scale, head_stride=32, out=out, dqkv=dqkv))
    ref = dqkv.clone()
    b = timed(lambda: ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout, n_seq, S, h, hd, scale, head_stride=32, out=out, dqkv=dqkv, lse=lse))
    err = ((dqkv - ref).abs().max() / ref.abs().max()).item()
    print('n_seq %5d S %4d   plain %8.1f us   with lse %8.1f us   rel diff %.1e' % (n_seq, S, a, b, err), flush=True)
