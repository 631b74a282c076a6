"""Diagnostic: where does token_attn_bwd_kernel spend its time?  Builds variants of backward_f32.hip with phases removed
(-DLIME_ATTN_BWD_ABLATE=mask: 1 no global staging, 2 no S / dP MFMAs, 4 no softmax, 8 no dV, 16 no dQ / dK, 32 no stores) and
times the body (S = 128) and title (S = 32) shapes.  64: the S / dP MFMAs run on register operands (no K / V fragment reads).

    python tools/attn_bwd_ablate.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

MASKS = [0, 2, 64, 4 | 8 | 16, 4 | 8 | 16 | 64, 2 | 4 | 8 | 16]


def build(mask):
    so = os.path.join(ROOT, 'tools', 'probes', 'liblime_attn_bwd_%d.so' % mask)
    src = os.path.join(ROOT, 'lime_cikm25_amd', 'csrc')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(src, 'backward_f32.hip')):
        subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
                        '-DLIME_ATTN_BWD_ABLATE=%d' % mask, '-o', so, os.path.join(src, 'backward_f32.hip'),
                        os.path.join(src, 'common.cpp')], check=True)
    return so


def main():
    dev = 'cuda'
    nh, hd, hs = 10, 30, 32
    W = nh * hs
    for mask in MASKS:
        lib = ctypes.CDLL(build(mask))
        f = lib.lime_token_attention_bwd_f32
        f.restype = ctypes.c_int32
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + \
                     [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_int32] * 5 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_int64,
                                                                                      ctypes.c_float, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        line = 'mask %2d' % mask
        for n_seq, S in ((1760, 128), (1760, 32)):
            tok = n_seq * S
            qkv = (torch.rand(tok, 3 * W, device=dev) - 0.5)
            dout = torch.rand(tok, nh * hd, device=dev) - 0.5
            dqkv = torch.empty_like(qkv)
            P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)

            def run():
                st = f(P(qkv), P(qkv, W), P(qkv, 2 * W), 3 * W, None, 0, P(dout), nh * hd, P(dqkv), P(dqkv, W), P(dqkv, 2 * W), 3 * W,
                       n_seq, S, nh, hd, hs, 1.0 / hd ** 0.5, None, 0, 0.0, 0, 0, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert st == 0
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            line += '   S=%3d %7.1f us' % (S, e0.elapsed_time(e1) * 100)
        print(line, flush=True)


if __name__ == '__main__':
    main()
