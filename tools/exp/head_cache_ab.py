"""Experiment (not product code): how much of a config-2b step is the weight-only part of the head?  Memoises everything in the
forward that depends on parameters alone (padded in_proj weights, positional table through in_proj, the S padding rows, the
freshness table, the stacked intent weights) and times the plain loop with and without, interleaved in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from lime_cikm25_amd import newsEncoders as ne, ops

run = bench.Run('cfg2b', 0, 1)
sync = torch.cuda.synchronize

orig_prepare = ne.compact_prepare
memo = {}

def cached_prepare(ids, table, pe, transformer, nhead):
    M, S = ids.shape
    key = (M, S, id(transformer))
    E = table.shape[1]; hd = E // nhead; W = nhead * 32
    cap = (M + 1) * S
    cmp = ops.compact_sequences(ids)
    if key not in memo:
        sa = transformer.layers[0].self_attn
        w_in = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, 32)
        b_in = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, 32)
        pew = ops.linear(pe[:S], w_in, b_in)
        qkv = torch.empty((cap + S, 3 * W), dtype=torch.float32, device=ids.device)
        ops.linear(table, w_in, None, a_ids=ne._zero_ids(S, ids.device), res=pew, res_mod=S, out=qkv[cap:])
        memo[key] = (w_in, pew, qkv)
    w_in, pew, qkv = memo[key]
    return cmp, w_in, pew, qkv

orig_lime_flat = ne.LIME.encode_flat

def lime_flat_cached(self, title_text, title_mask, content_text, category, subCategory, freshness, lifetime):
    M = title_text.shape[0]
    cdim = self.base_news_encoder.news_embedding_dim
    fe = self.freshness_encoder
    E, nb = fe.freshness_embedding.embedding_dim, fe.num_buckets
    main = torch.cuda.current_stream()
    side = ne._side_stream(title_text.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        pair = torch.add(fe.buckets(lifetime), fe.buckets(freshness), alpha=nb)
        if 'table' not in memo:
            t_f = ops.linear(fe.freshness_embedding.weight, fe.dense.weight[:, :E], None)
            t_l = ops.linear(fe.lifetime_embedding.weight, fe.dense.weight[:, E:], fe.dense.bias)
            fresh = torch.tanh(t_f.unsqueeze(1) + t_l.unsqueeze(0)).view(nb * nb, -1)
            memo['table'] = ops.linear(fresh, self.project.weight[:, cdim:], self.project.bias)
        table = memo['table']
    content = torch.empty((M, cdim), dtype=torch.float32, device=title_text.device)
    self.base_news_encoder.encode_flat(title_text, title_mask, content_text, category, subCategory, content)
    main.wait_stream(side)
    return ops.linear(content, self.project.weight[:, :cdim], None, res=table, res_ids=pair)

orig_cat, orig_pad = torch.cat, torch.nn.functional.pad

def timed(n):
    sync(); t0 = time.perf_counter()
    for _ in range(n): run.step()
    sync(); return (time.perf_counter() - t0) / n * 1e3

res = {'base': [], 'cached': [], 'cached2': []}
for rnd in range(3):
    for mode in ('base', 'cached', 'cached2'):
        ne.compact_prepare = orig_prepare if mode == 'base' else cached_prepare
        ne.LIME.encode_flat = lime_flat_cached if mode == 'cached2' else orig_lime_flat
        run.model._graphs.clear()
        for _ in range(10): run.step()
        res[mode].append(round(timed(300), 4))
print(res)
