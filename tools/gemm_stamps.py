"""Diagnostic: where a persistent GEMM workgroup spends its cycles (s_memtime stamps, LIME_STAMPS build of gemm_f32.hip).

    python tools/gemm_stamps.py            # builds tools/probes/liblime_stamps.so if missing, runs the encoder GEMM shapes
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import _lib, ops  # noqa: E402

SO = os.path.join(ROOT, 'tools', 'probes', 'liblime_stamps.so')
SEG = ['acc_init', 'issue', 'mfma', 'commit', 'barrier', 'switch+tail', 'epilogue', 'next commit+barrier']
SEG_PP = ['acc_init', 'dma issue', 'reads+mfma', 'dma wait', 'barrier', 'switch', 'epilogue', '-']
SEG_SP = ['acc_init', 'chunk 0 compute', 'chunk compute', 'chunk 0 wait (DMA + stores)', 'chunk wait (DMA)', 'barrier', 'epilogue', '-']


def build():
    src = os.path.join(ROOT, 'lime_cikm25_amd', 'csrc')
    subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-DLIME_STAMPS', '-o', SO,
                    os.path.join(src, 'gemm_f32.hip'), os.path.join(src, 'gemm_pp_f32.hip'), os.path.join(src, 'gemm_sp_f32.hip'),
                    os.path.join(src, 'gemm_mid_f32.hip'), os.path.join(src, 'common.cpp')], check=True)


def main():
    if not os.path.exists(SO):
        build()
    _lib.LIB_PATH = SO
    lib = ctypes.CDLL(SO)
    lib.lime_linear_f32.restype = ctypes.c_int32
    lib.lime_linear_f32.argtypes = [ctypes.POINTER(_lib.LinearArgs), ctypes.c_void_p]
    lib.lime_last_error_string.restype = ctypes.c_char_p
    lib.lime_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.lime_debug_set_pp_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.lime_debug_set_sp_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.lime_last_linear_kernel.restype = ctypes.c_char_p

    class Shim:
        lime_linear_f32 = lib.lime_linear_f32
        lime_last_error_string = lib.lime_last_error_string
        lime_last_linear_kernel = lib.lime_last_linear_kernel
    _lib._lib = Shim
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).to(dev)
    tok, E, F, V, S = int(os.environ.get('STAMP_TOKENS', '225280')), 300, 512, 50000, 128
    table, pe = rnd(V, E), rnd(S, E)
    ids = torch.randint(0, V, (tok,), generator=g, dtype=torch.int32).to(dev)
    x, h = rnd(tok, E), rnd(tok, F)
    ln = (rnd(E) + 1, rnd(E))
    pew = rnd(S, 960)
    cases = {
        'qkv_body': lambda: ops.linear(table, rnd(960, E), None, a_ids=ids, res=pew, res_mod=S),
        'ffn1_body': lambda: ops.linear(x, rnd(F, E), rnd(F), act='relu'),
        'ffn2_body': lambda: ops.linear(h, rnd(E, F), rnd(E), res=x, ln=ln),
        'out_body': lambda: ops.linear(x, rnd(E, E), rnd(E), res=table, res_ids=ids, res_pe=pe, res_period=S, ln=ln),
    }
    for name, fn in cases.items():
        buf = torch.zeros(512 * 8 * 8 * 2, dtype=torch.int64, device=dev)
        lib.lime_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        lib.lime_debug_set_pp_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        lib.lime_debug_set_sp_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        fn()
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        kern = lib.lime_last_linear_kernel().decode()
        t = buf.view(-1, 8).double()
        t = t[t.sum(dim=1) > 0]
        share = t.sum(dim=0) / t.sum()
        seg = SEG_SP if kern.startswith('gemm_sp') else (SEG_PP if kern.startswith('gemm_pp') else SEG)
        print('%-10s %s  waves %d  wave total %.0f s_memtime ticks  ' % (name, kern, t.shape[0], t.sum(dim=1).mean().item()) +
              '  '.join('%s %.1f%%' % (s, 100 * v) for s, v in zip(seg, share.tolist())))
    lib.lime_debug_set_stamp_buffer(None)
    lib.lime_debug_set_pp_stamp_buffer(None)
    lib.lime_debug_set_sp_stamp_buffer(None)


if __name__ == '__main__':
    main()
