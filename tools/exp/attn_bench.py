"""Token attention forward (heads padded to 32 columns), fp32-MFMA kernel against the split-product kernel.
    python tools/exp/attn_bench.py     (GPU box; us per call, TFLOP/s on 4 S^2 32 per (sequence, head))"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from lime_cikm25_amd import ops


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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    h, hd = 10, 30
    W = h * 32
    for n_seq, S in ((516, 32), (486, 128), (1760, 128), (1760, 32), (440, 512), (1760, 512), (880, 256)):
        qkv = torch.randn(n_seq * S, 3 * W, device='cuda')
        qkv.view(-1, 3 * h, 32)[:, :, hd:] = 0
        out = torch.empty(n_seq * S, h * hd, device='cuda')
        scale = 1.0 / math.sqrt(hd)
        res = []
        for on in (False, True):
            ops.set_split_gemm(on)
            res.append(timed(lambda: ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, h, hd, scale, head_stride=32, out=out)))
        gf = 4.0 * S * S * 32 * n_seq * h
        print('n_seq %5d S %4d   fp32 %7.1f us %6.1f TF   split %7.1f us %6.1f TF' % (n_seq, S, res[0], gf / res[0] * 1e-6, res[1], gf / res[1] * 1e-6), flush=True)
    ops.set_split_gemm(True)


if __name__ == '__main__':
    main()
