"""Diagnostic: cycle shares of the attention kernel's loop segments (LIME_STAMPS build of token_attn_f32.hip)."""
import ctypes
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SO = os.path.join(ROOT, 'tools', 'probes', 'liblime_attn_stamps.so')
SEG = ['stash', 'barrier1', 'q+prefetch issue', 'QK', 'softmax', 'PV', 'out', 'barrier2']


def main():
    src = os.path.join(ROOT, 'lime_cikm25_amd', 'csrc')
    if not os.path.exists(SO):
        subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-DLIME_STAMPS', '-o', SO,
                        os.path.join(src, 'token_attn_f32.hip'), os.path.join(src, 'token_attn_bf16.hip'), os.path.join(src, 'common.cpp')], check=True)
    lib = ctypes.CDLL(SO)
    v = ctypes.c_void_p
    lib.lime_token_attention_f32.argtypes = [v, v, v, ctypes.c_int64, v, v, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, v]
    lib.lime_debug_set_attn_stamp_buffer.argtypes = [v]
    for S, n_seq in ((128, 1760), (32, 1760)):
        E, W = 300, 320                     # heads padded to 32 columns (the model's layout)
        qkv = (torch.rand(n_seq * S, 3 * W) * 2 - 1).cuda()
        qkv.view(-1, 30, 32)[:, :, 30:] = 0
        out = torch.empty(n_seq * S, E, device='cuda')
        buf = torch.zeros(768 * 4 * 8, dtype=torch.int64, device='cuda')
        lib.lime_debug_set_attn_stamp_buffer(v(buf.data_ptr()))
        for _ in range(2):
            buf.zero_()
            st = lib.lime_token_attention_f32(v(qkv.data_ptr()), v(qkv.data_ptr() + W * 4), v(qkv.data_ptr() + 2 * W * 4), 3 * W, None,
                                              v(out.data_ptr()), E, n_seq, S, 10, 30, 32, 1 / math.sqrt(30), None)
            assert st == 0
            torch.cuda.synchronize()
        t = buf.view(-1, 8).double()
        t = t[t.sum(dim=1) > 0]
        share = t.sum(dim=0) / t.sum()
        print('S=%d  waves %d  wave total %.0f cycles  ' % (S, t.shape[0], t.sum(dim=1).mean().item()) +
              '  '.join('%s %.1f%%' % (n, 100 * x) for n, x in zip(SEG, share.tolist())))


if __name__ == '__main__':
    main()
