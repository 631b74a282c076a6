"""Diagnostic: where a workgroup of lime_encoder_ffn_bf16 spends its cycles (s_memtime stamps, LIME_STAMPS build of ffn_bf16.hip).

    python tools/ffn_stamps.py [rows]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import _lib, ops  # noqa: E402

SO = os.path.join(ROOT, 'tools', 'probes', 'liblime_ffn_stamps.so')
SEG = ['dma wait', 'barrier', 'out_proj reads+mfma', 'linear1 reads+mfma', 'linear2 reads+mfma', 'residual + tile issue', 'out_proj epilogue + relu/pack', 'epilogue']


def main():
    run('block stamps', ['-DLIME_STAMPS'], block=True)
    run('block plain', [], block=True)
    for name, flags in (('stamps', ['-DLIME_STAMPS']), ('plain', []), ('no compute', ['-DLIME_FFN_ABLATE=1']), ('no weight DMA', ['-DLIME_FFN_ABLATE=2']),
                        ('reads, no MFMA', ['-DLIME_FFN_ABLATE=3']), ('no LN epilogue', ['-DLIME_FFN_ABLATE=5'])):
        run(name, flags)


def run(name, flags, block=False):
    src = os.path.join(ROOT, 'lime_cikm25_amd', 'csrc')
    so = SO.replace('.so', '_%s.so' % name.replace(' ', '_'))
    subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', so,
                    os.path.join(src, 'ffn_bf16.hip'), os.path.join(src, 'common.cpp')] + flags, check=True)
    lib = ctypes.CDLL(so)
    stamps = '-DLIME_STAMPS' in flags
    lib.lime_encoder_ffn_bf16.restype = ctypes.c_int32
    lib.lime_encoder_ffn_bf16.argtypes = [ctypes.POINTER(_lib.FfnBf16Args), ctypes.c_void_p]
    lib.lime_encoder_block_bf16.restype = ctypes.c_int32
    lib.lime_encoder_block_bf16.argtypes = [ctypes.POINTER(_lib.EncoderBlockBf16Args), ctypes.c_void_p]
    if stamps:
        lib.lime_debug_set_ffn_stamp_buffer.argtypes = [ctypes.c_void_p]
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 360448
    E, EP, F = 300, 304, 512
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).cuda()
    x = torch.zeros(M, EP, dtype=torch.bfloat16, device='cuda')
    x[:, :E] = rnd(M, E).to(torch.bfloat16)
    w1, b1, w2, b2 = rnd(F, E), rnd(F), rnd(E, F), rnd(E)
    gam, bet = rnd(E) + 1, rnd(E)
    w1p, w2p = ops.ffn_pack_bf16(w1, b1, w2)
    out = torch.empty((M // 32, EP), device='cuda')
    a = _lib.FfnBf16Args()
    a.x, a.ldx, a.w1p, a.w2p = x.data_ptr(), EP, w1p.data_ptr(), w2p.data_ptr()
    a.b2, a.ln_gamma, a.ln_beta, a.ln_eps, a.pool32 = b2.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5, 1
    a.out, a.ldo, a.M, a.E, a.F = out.data_ptr(), EP, M, E, F
    call = lambda: lib.lime_encoder_ffn_bf16(ctypes.byref(a), st)
    if block:
        V, S = 60000, 128
        table = torch.zeros(V, EP, dtype=torch.bfloat16, device='cuda')
        table[:, :E] = rnd(V, E).to(torch.bfloat16)
        ids = torch.randint(0, V, (M,), generator=g, dtype=torch.int32).cuda()
        add = rnd(S, E)
        w0p = ops.oproj_pack_bf16(rnd(E, E))
        g1, be1 = rnd(E) + 1, rnd(E)
        b = _lib.EncoderBlockBf16Args()
        b.attn, b.lda, b.w0p, b.add_rows, b.ld_add, b.add_period = x.data_ptr(), EP, w0p.data_ptr(), add.data_ptr(), E, S
        b.res_kind, b.res, b.ldr, b.res_rows, b.res_ids = 2, table.data_ptr(), EP, V, ids.data_ptr()
        b.ln1_gamma, b.ln1_beta, b.ln1_eps, b.pool32 = g1.data_ptr(), be1.data_ptr(), 1e-5, 1
        b.w1p, b.w2p, b.b2, b.ln2_gamma, b.ln2_beta, b.ln2_eps = w1p.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5
        b.M, b.E, b.F, b.out, b.ldo = M, E, F, out.data_ptr(), EP
        call = lambda: lib.lime_encoder_block_bf16(ctypes.byref(b), st)
    buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device='cuda')
    if stamps:
        lib.lime_debug_set_ffn_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(4):
        assert call() == 0
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record()
    torch.cuda.synchronize()
    if not stamps:
        print('%-14s rows %d  %.1f us' % (name, M, e0.elapsed_time(e1) * 1e3 / 5))
        return
    t = buf.view(-1, 8).double()
    t = t[t.sum(dim=1) > 0]
    share = t.sum(dim=0) / t.sum()
    print('stamps         rows %d  %.1f us  waves %d  wave total %.0f s_memtime ticks' % (M, e0.elapsed_time(e1) * 1e3 / 5, t.shape[0], t.sum(dim=1).mean().item()))
    print('  '.join('%s %.1f%%' % (s, 100 * v) for s, v in zip(SEG, share.tolist())))


if __name__ == '__main__':
    main()
