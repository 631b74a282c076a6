import sys, math, torch
sys.path.insert(0, '.')
from lime_cikm25_amd import ops
nh, hd, hs = 10, 30, 32
W = nh * hs
for n_seq, S in ((1760, 128), (1760, 32)):
    tok = n_seq * S
    qkv = torch.rand(tok, 3 * W, device='cuda') - 0.5
    qkv.view(tok, 3 * nh, hs)[:, :, hd:] = 0
    q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
    sc = 1 / math.sqrt(hd)
    fns = {'tuned scoring kernel': lambda: ops.token_attention(q, k, v, n_seq, S, nh, hd, sc, head_stride=hs),
           'transposed, p=0': lambda: ops.token_attention_dropout(q, k, v, n_seq, S, nh, hd, sc, 0.0, 1, 2, head_stride=hs),
           'transposed, p=0.2': lambda: ops.token_attention_dropout(q, k, v, n_seq, S, nh, hd, sc, 0.2, 1, 2, head_stride=hs)}
    a, b = fns['tuned scoring kernel'](), fns['transposed, p=0']()
    print('S=%d max diff %.2e' % (S, float((a - b).abs().max())))
    for name, fn in fns.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print('  %-22s %7.1f us' % (name, e0.elapsed_time(e1) * 50))
