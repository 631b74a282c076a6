"""One shape of the attention backward, a few launches (for counter collection).  python tools/exp/attn_bwd_one.py [n_seq] [dropout|plain] [S]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from lime_cikm25_amd import ops
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
drop = (0.2, 1234, 3) if len(sys.argv) > 2 and sys.argv[2] == 'dropout' else None
h, hd = 10, 30
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
W = h * 32
qkv = torch.randn(n_seq * S, 3 * W, device='cuda')
qkv.view(-1, 3 * h, 32)[:, :, hd:] = 0
dout = torch.randn(n_seq * S, h * hd, device='cuda')
dqkv = torch.empty_like(qkv)
lse = torch.empty(n_seq * S * h, device='cuda') if S > 128 else None
out = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, h, hd, 1.0 / math.sqrt(hd), head_stride=32, lse=lse) if S > 128 else None
for _ in range(3):
    ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout, n_seq, S, h, hd, 1.0 / math.sqrt(hd), head_stride=32, dqkv=dqkv, dropout=drop,
                            out=out, lse=lse)
torch.cuda.synchronize()
