"""Weight-gradient launches of one encoder layer (lime_linear_wgrad_f32), fp32-MFMA kernels against the split-product kernel.
    python tools/exp/wgrad_bench.py            (GPU box; us per call = kernel + partial-sum reduction, TFLOP/s on 2 M N K)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from lime_cikm25_amd import ops

SHAPES = [('in_proj', 960, 300), ('out_proj', 300, 300), ('linear1', 512, 300), ('linear2', 300, 512)]


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
    for M in (16512, 62208, 225280):
        for name, N, K in SHAPES:
            dy = torch.randn(M, N, device='cuda')
            x = torch.randn(M, K, device='cuda')
            out = torch.empty(N, K, device='cuda')
            db = torch.empty(N, device='cuda')
            res = []
            ref = None
            for on in (False, True):
                ops.set_split_gemm(on)
                t = timed(lambda: ops.linear_wgrad(dy, x, out=out, bias_out=db))
                res.append(t)
                if ref is None:
                    ref = out.clone()
                else:
                    err = ((out - ref).abs().max() / ref.abs().max()).item()
            gf = 2.0 * M * N * K
            print('M %6d %-8s N %4d K %4d   fp32 %7.1f us %6.1f TF   split %7.1f us %6.1f TF   rel diff %.1e' % (
                M, name, N, K, res[0], gf / res[0] * 1e-6, res[1], gf / res[1] * 1e-6, err), flush=True)
    ops.set_split_gemm(True)


if __name__ == '__main__':
    main()
