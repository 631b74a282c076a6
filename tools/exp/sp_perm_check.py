import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lime_cikm25_amd import ops, _lib
torch.manual_seed(0)
def rnd(*s, scale=1.0): return (torch.rand(*s, device='cuda') * 2 - 1) * scale
M, E, F, S, V = 40000, 300, 512, 128, 5000
a, r = rnd(M, E), rnd(M, E)
w300, b300, w960, w512, b512, w2 = rnd(E, E, scale=.05), rnd(E), rnd(960, E, scale=.05), rnd(F, E, scale=.05), rnd(F), rnd(E, F, scale=.05)
g, be = rnd(E) + 1.5, rnd(E)
h = rnd(M, F)
perm = torch.randperm(M, device='cuda')
blk = (torch.randperm(M // 32, device='cuda')[:, None] * 32 + torch.arange(32, device='cuda')[None, :]).reshape(-1)     # permutes 32-row blocks
table = rnd(V, E); ids = torch.randint(0, V, (M,), device='cuda', dtype=torch.int32); pew = rnd(S, 960)
for split in (True, False):
    ops.set_split_gemm(split)
    k = lambda: _lib.load().lime_last_linear_kernel().decode()[:48]
    y = ops.linear(a, w300, b300); yp = ops.linear(a[perm].contiguous(), w300, b300); print(split, 'plain300', torch.equal(y[perm], yp), k())
    y = ops.linear(a, w960, None); yp = ops.linear(a[perm].contiguous(), w960, None); print(split, 'plain960', torch.equal(y[perm], yp), k())
    y = ops.linear(a, w512, b512, act='relu'); yp = ops.linear(a[perm].contiguous(), w512, b512, act='relu'); print(split, 'relu512', torch.equal(y[perm], yp), k())
    y = ops.linear(h, w2, b300, res=r, ln=(g, be)); yp = ops.linear(h[perm].contiguous(), w2, b300, res=r[perm].contiguous(), ln=(g, be)); print(split, 'res+LN', torch.equal(y[perm], yp), k())
    y = ops.linear(h, w2, b300, res=r, ln=(g, be), pool32=True); yp = ops.linear(h[blk].contiguous(), w2, b300, res=r[blk].contiguous(), ln=(g, be), pool32=True)
    print(split, 'pool32', torch.equal(y[blk[::32] // 32], yp), k())
    y = ops.linear(table, w960, None, a_ids=ids, res=pew, res_mod=S)
    p2 = (torch.randperm(M // S, device='cuda')[:, None] * S + torch.arange(S, device='cuda')[None, :]).reshape(-1)
    yp = ops.linear(table, w960, None, a_ids=ids[p2].contiguous(), res=pew, res_mod=S); print(split, 'gather+periodic', torch.equal(y[p2], yp), k())
