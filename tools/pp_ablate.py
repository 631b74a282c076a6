"""Diagnostic: what bounds gemm_pp_kernel?  Times the encoder GEMM shapes on three builds of the library:
full, -DLIME_PP_ABLATE=1 (operand DMA + epilogue only, no MFMAs) and -DLIME_PP_ABLATE=2 (MFMAs + epilogue, no DMA).

    python tools/pp_ablate.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import _lib, ops  # noqa: E402


def build(mode):
    so = os.path.join(ROOT, 'tools', 'probes', 'liblime_ablate%d.so' % mode)
    src = os.path.join(ROOT, 'lime_cikm25_amd', 'csrc')
    if not os.path.exists(so):
        subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-DLIME_PP_ABLATE=%d' % mode,
                        '-o', so, os.path.join(src, 'gemm_f32.hip'), os.path.join(src, 'gemm_pp_f32.hip'),
                        os.path.join(src, 'common.cpp')], check=True)
    return so


def main():
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).to(dev)
    tok, E, F, V, S = 225280, 300, 512, 50000, 128
    table, pe = rnd(V, E), rnd(S, E)
    ids = torch.randint(0, V, (tok,), generator=g, dtype=torch.int32).to(dev)
    x, h = rnd(tok, E), rnd(tok, F)
    ln = (rnd(E) + 1, rnd(E))
    pew, wq, w1, b1, w2, b2, wo, bo = rnd(S, 960), rnd(960, E), rnd(F, E), rnd(F), rnd(E, F), rnd(E), rnd(E, E), rnd(E)
    oq, o1, o2 = torch.empty(tok, 960, device=dev), torch.empty(tok, F, device=dev), torch.empty(tok, E, device=dev)
    cases = {
        'qkv_body': lambda: ops.linear(table, wq, None, a_ids=ids, res=pew, res_mod=S, out=oq),
        'ffn1_body': lambda: ops.linear(x, w1, b1, act='relu', out=o1),
        'ffn2_body': lambda: ops.linear(h, w2, b2, res=x, ln=ln, out=o2),
        'out_body': lambda: ops.linear(x, wo, bo, res=table, res_ids=ids, res_pe=pe, res_period=S, ln=ln, out=o2),
    }
    names = {0: 'full', 1: 'no MFMA', 2: 'no DMA'}
    for mode in (0, 1, 2):
        lib = ctypes.CDLL(build(mode))
        lib.lime_linear_f32.restype = ctypes.c_int32
        lib.lime_linear_f32.argtypes = [ctypes.POINTER(_lib.LinearArgs), ctypes.c_void_p]
        lib.lime_last_error_string.restype = ctypes.c_char_p
        lib.lime_last_linear_kernel.restype = ctypes.c_char_p

        class Shim:
            lime_linear_f32 = lib.lime_linear_f32
            lime_last_error_string = lib.lime_last_error_string
            lime_last_linear_kernel = lib.lime_last_linear_kernel
        _lib._lib = Shim
        line = []
        for name, fn in cases.items():
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            line.append('%s %.0f us' % (name, e0.elapsed_time(e1) * 50))
        print('%-8s %s' % (names[mode], '   '.join(line)))


if __name__ == '__main__':
    main()
