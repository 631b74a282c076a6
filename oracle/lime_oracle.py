"""CPU oracle: a functional restatement of LIME's candidate-scoring path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (lime_cikm25_amd/) may import this
file; it is used by tests/, by __graft_entry__.smoke() and by bench.py's ``cpu_baseline`` leg, as
the checker and the timed CPU baseline -- never as the thing shipped or measured as the product.

Pinned: every function below is checked in tests/test_oracle_golden.py against golden vectors
captured from the reference itself (tools/make_goldens.py imports /root/reference on CPU in the
build container; the vectors live in tests/golden/).  One boundary stays *parity unpinned*:
``torch_geometric.nn.GraphSAGE`` (userEncoders.py:54-58,153) is an absent, un-versioned third-party
dependency; ``graph_sage`` below restates PyG's documented SAGEConv and the goldens pin it only
against the same restatement wired into the imported reference (tools/ref_harness.py).

Plain torch fp32 on CPU (this is floating-point work), functional, keyed by the reference's
``state_dict`` names so a reference checkpoint drives it directly.  The two integer bucketisations
use fp32 threshold tables derived from the reference's own ``bucketize`` (see BUCKET_THRESHOLD_BITS).

All citations are file:line into the reference repository.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# Smallest fp32 x (as IEEE-754 bit patterns) whose bucket is >= k, k = 1..9, found by bisection over
# bit patterns with the reference's FreshnessEncoder.bucketize (newsEncoders.py:53-58) on torch CPU
# and verified monotone on +-1e5 ulp windows and 5e6 random samples (tools/derive_buckets.py).
# b(x) = min(trunc(log(max(x,1)) / log(86400) * (10/7)), 9) is evaluated there in fp32; comparing
# against these cut points reproduces it bit-exactly without depending on any libm's logf.
BUCKET_THRESHOLD_BITS = (
    0x45326B18,  # 2854.6934
    0x4AF8B232,  # 8149273.0
    0x50AD53E8,  # 2.3263658e10
    0x567199BD,  # 6.6410651e13
    0x5C2861F4,  # 1.8958199e17
    0x61EAB505,  # 5.4119774e20
    0x67A39429,  # 1.5449576e24
    0x6D6402D2,  # 4.4103745e27
    0x731EE960,  # 1.2590276e31
)
BUCKET_THRESHOLDS = np.array(BUCKET_THRESHOLD_BITS, dtype=np.uint32).view(np.float32)


def bucketize(x, num_buckets=10):
    """newsEncoders.py:53-58.  x: float tensor -> int64 buckets in [0, num_buckets-1].

    The reference default num_buckets=10 has a threshold table (pinned by the goldens); other counts evaluate the rule directly.
    Non-finite inputs index out of range in the reference; here NaN -> 0 and +inf -> 9.
    """
    if num_buckets != 10:                         # no threshold table: the rule itself, as torch evaluates it in fp32 on the CPU
        xf = torch.clamp(x.detach().cpu().float(), min=1)
        scaled = torch.log(xf) / torch.log(torch.tensor(60 * 60 * 24.0))
        return torch.clamp((scaled * (num_buckets / 7)).long(), max=num_buckets - 1)
    xn = x.detach().cpu().float().numpy()
    b = np.searchsorted(BUCKET_THRESHOLDS, xn, side='right').astype(np.int64)
    b[np.isnan(xn)] = 0
    return torch.from_numpy(b)


def positional_encoding(length, d_model):
    """newsEncoders.py:812-818 (the registered ``pe`` buffer, without the leading batch axis)."""
    position = torch.arange(0, length, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


# Test instrumentation (tests/test_bf16_budget.py): name -> callable applied to that intermediate of the token encoders, e.g. a
# round-to-bf16, to attribute the error of the bf16 build (BASELINE configs[2]) to its rounding points.  Empty: the reference's
# arithmetic, untouched.
ROUND = {}


def _tap(name, t):
    f = ROUND.get(name)
    return f(t) if f is not None else t


def encoder_layer(x, sd, p, nhead):
    """One post-LN ``nn.TransformerEncoderLayer`` (ReLU, eps 1e-5, no mask), as constructed at
    newsEncoders.py:244-247 and called at :316,:320.  x: [M, S, E]."""
    M, S, E = x.shape
    hd = E // nhead
    W = lambda k: _tap('w:' + k, _tap('weights', sd[p + k]))
    qkv = _tap('qkv', x @ W('self_attn.in_proj_weight').t() + sd[p + 'self_attn.in_proj_bias'])
    q, k, v = qkv.split(E, dim=-1)
    q = q.view(M, S, nhead, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = k.view(M, S, nhead, hd).transpose(1, 2)
    v = v.view(M, S, nhead, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    o = _tap('attn_out', (a @ v).transpose(1, 2).reshape(M, S, E))
    o = o @ W('self_attn.out_proj.weight').t() + sd[p + 'self_attn.out_proj.bias']
    x = layer_norm(x + o, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'])
    h = _tap('h', torch.relu(_tap('x1', x) @ W('linear1.weight').t() + sd[p + 'linear1.bias']))
    h = h @ W('linear2.weight').t() + sd[p + 'linear2.bias']
    return layer_norm(x + h, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'])


def additive_attention(feature, sd, p, mask=None):
    """layers.Attention.forward, layers.py:285-300.  feature [M, len, D] -> [M, D]."""
    a = torch.tanh(feature @ sd[p + 'affine1.weight'].t() + sd[p + 'affine1.bias'])
    a = (a @ sd[p + 'affine2.weight'].t()).squeeze(-1)
    if mask is not None:
        a = a.masked_fill(mask == 0, -1e9)
    alpha = torch.softmax(a, dim=1).unsqueeze(1)
    return torch.bmm(alpha, feature).squeeze(1)


def crown_news_encoder(sd, p, cfg, title_text, content_text, category, subCategory, taps=None):
    """newsEncoders.CROWN.forward, newsEncoders.py:302-373 (eval mode: dropouts are identity).
    The token masks are computed and never used there (:307-308).  -> [B, n, 900]."""
    B, n, T = title_text.shape
    L = content_text.shape[2]
    M = B * n
    E = cfg.word_embedding_dim
    emb = sd[p + 'word_embedding.weight']
    title = _tap('word_rows', emb[title_text.reshape(M, T).long()]) + positional_encoding(T, E)          # :311,:315
    body = _tap('word_rows', emb[content_text.reshape(M, L).long()]) + positional_encoding(L, E)        # :312,:319
    title_t, body_t = title, body
    for li in range(getattr(cfg, 'num_layers', 1)):                                  # config.py:70 (1 or 2), newsEncoders.py:244-247
        title_t = encoder_layer(title_t, sd, p + 'title_transformer.layers.%d.' % li, cfg.head_num)   # :316
        body_t = encoder_layer(body_t, sd, p + 'body_transformer.layers.%d.' % li, cfg.head_num)      # :320
    title_e = title_t.mean(dim=1)                                                    # :317
    body_e = body_t.mean(dim=1)                                                      # :321
    cat = sd[p + 'category_embedding.weight'][category.reshape(M).long()]
    sub = sd[p + 'subCategory_embedding.weight'][subCategory.reshape(M).long()]
    cat_rep = torch.cat([cat, sub], dim=1) @ sd[p + 'category_affine.weight'].t() + sd[p + 'category_affine.bias']  # :340-342

    def intents(e):                                                                  # :284-295
        x = torch.cat([e, cat_rep], dim=1)
        return torch.stack([torch.relu(x @ sd[p + 'intent_layers.%d.weight' % i].t() + sd[p + 'intent_layers.%d.bias' % i])
                            for i in range(cfg.intent_num)], dim=1)

    title_i = additive_attention(intents(title_e), sd, p + 'title_intent_attention.')  # :355
    body_i = additive_attention(intents(body_e), sd, p + 'body_intent_attention.')     # :356
    sim = (F.cosine_similarity(title_i, body_i, dim=1) + 1) / 2.0                     # :297-300
    rep = torch.cat([title_i, sim.unsqueeze(1) * body_i, cat, sub], dim=1)            # :369, :221-225
    if taps is not None:
        taps.setdefault('title_pooled', []).append(title_e)
        taps.setdefault('body_pooled', []).append(body_e)
    return rep.view(B, n, -1)


def multi_head_attention(x, sd, p, h, d_k, mask):
    """layers.MultiHeadAttention.forward, layers.py:222-238 (Q=K=V=x).  x [M, S, E] -> [M, S, h*d_k]."""
    M, S, _ = x.shape
    q = (x @ sd[p + 'W_Q.weight'].t() + sd[p + 'W_Q.bias']).view(M, S, h, d_k).transpose(1, 2)
    k = (x @ sd[p + 'W_K.weight'].t() + sd[p + 'W_K.bias']).view(M, S, h, d_k).transpose(1, 2)
    v = (x @ sd[p + 'W_V.weight'].t() + sd[p + 'W_V.bias']).view(M, S, h, d_k).transpose(1, 2)
    a = q @ k.transpose(-2, -1) / math.sqrt(float(d_k))
    a = a.masked_fill(mask.view(M, 1, 1, S) == 0, -1e9)
    return (torch.softmax(a, dim=-1) @ v).transpose(1, 2).reshape(M, S, h * d_k)


def mhsa_news_encoder(sd, p, cfg, title_text, title_mask, category, subCategory):
    """newsEncoders.MHSA.forward, newsEncoders.py:582-595 (title only).  -> [B, n, h*d_k + 100]."""
    B, n, T = title_text.shape
    M = B * n
    mask = title_mask.reshape(M, T)
    w = sd[p + 'word_embedding.weight'][title_text.reshape(M, T).long()]
    c = multi_head_attention(w, sd, p + 'multiheadAttention.', cfg.head_num, cfg.head_dim, mask)
    rep = additive_attention(c, sd, p + 'attention.', mask=mask)
    cat = sd[p + 'category_embedding.weight'][category.reshape(M).long()]
    sub = sd[p + 'subCategory_embedding.weight'][subCategory.reshape(M).long()]
    return torch.cat([rep, cat, sub], dim=1).view(B, n, -1)


def freshness_encoder(sd, p, cfg, freshness, lifetime):
    """FreshnessEncoder.forward, newsEncoders.py:60-83.  -> ([B, n, hidden], f_bucket, l_bucket)."""
    fb = bucketize(freshness, cfg.num_buckets)
    lb = bucketize(lifetime, cfg.num_buckets)
    x = torch.cat([sd[p + 'freshness_embedding.weight'][fb], sd[p + 'lifetime_embedding.weight'][lb]], dim=-1)
    return torch.tanh(x @ sd[p + 'dense.weight'].t() + sd[p + 'dense.bias']), fb, lb


def lime_news_encoder(sd, p, cfg, title_text, title_mask, content_text, category, subCategory, freshness, lifetime,
                      taps=None):
    """LIME.forward, newsEncoders.py:140-161: fusion 'concat' (+ project, :151-153) -> [B, n, 400]; 'add' (:154-155) and 'gated'
    (:156-159) -> [B, n, content dim]."""
    bp = p + 'base_news_encoder.'
    if cfg.content_encoder == 'CROWN':
        content = crown_news_encoder(sd, bp, cfg, title_text, content_text, category, subCategory, taps)
    elif cfg.content_encoder == 'MHSA':
        content = mhsa_news_encoder(sd, bp, cfg, title_text, title_mask, category, subCategory)
    else:
        raise ValueError('content encoder %r is outside the scoring path' % cfg.content_encoder)
    fresh, fb, lb = freshness_encoder(sd, p + 'freshness_encoder.', cfg, freshness, lifetime)
    if cfg.fusion_method == 'concat':
        out = torch.cat([content, fresh], dim=-1)
        if cfg.lime_output_dim:
            out = out @ sd[p + 'project.weight'].t() + sd[p + 'project.bias']                           # :152-153
    elif cfg.fusion_method == 'add':
        out = content + fresh                                                                            # :155
    elif cfg.fusion_method == 'gated':
        gate = torch.sigmoid(torch.cat([content, fresh], dim=-1) @ sd[p + 'gate.weight'].t() + sd[p + 'gate.bias'])   # :157-158
        out = gate * content + (1 - gate) * fresh                                                        # :159
    else:
        raise ValueError('Unknown fusion method: %s' % cfg.fusion_method)
    if taps is not None:
        taps.setdefault('content', []).append(content)
        taps.setdefault('freshness', []).append(fresh)
        taps.setdefault('f_bucket', []).append(fb)
        taps.setdefault('l_bucket', []).append(lb)
        taps.setdefault('news_out', []).append(out)
    return out


def topic_representation(sd, p, category, subCategory):
    """userEncoders.py:103-105,115-117: LIME's own (frozen) category tables + category_affine."""
    x = torch.cat([sd[p + 'category_embedding.weight'][category.long()],
                   sd[p + 'subCategory_embedding.weight'][subCategory.long()]], dim=-1)
    return x @ sd[p + 'category_affine.weight'].t() + sd[p + 'category_affine.bias']


def candidate_aware_attention(sd, p, hist, hist_topic, cand_topic, mask, num_heads=10, residual=True):
    """CandidateAware_ClickedNewsAttention.forward, layers.py:52-93 (dropout identity; the
    value_proj branch :68,:76-77 is dead).  -> ([B, H, D], agg [B, H])."""
    B, H, D = hist.shape
    N = cand_topic.shape[1]
    hd = D // num_heads
    Q = (cand_topic @ sd[p + 'query_proj.weight'].t() + sd[p + 'query_proj.bias']).view(B, N, num_heads, hd).transpose(1, 2)
    K = (hist_topic @ sd[p + 'key_proj.weight'].t() + sd[p + 'key_proj.bias']).view(B, H, num_heads, hd).transpose(1, 2)
    s = Q @ K.transpose(-2, -1) / (D ** 0.5)                                          # :70  (sqrt(D), not sqrt(hd))
    s = s.masked_fill(mask.view(B, 1, 1, H) == 0, -1e9)                               # :72
    a = torch.softmax(s, dim=-1)                                                      # :73  [B, heads, N, H]
    qw = torch.softmax(torch.norm(Q.transpose(1, 2).reshape(B, N, -1), dim=-1), dim=1)  # :79
    agg = torch.softmax((a.sum(dim=1) * qw.unsqueeze(-1)).sum(dim=1), dim=-1)         # :80-81
    wc = agg.unsqueeze(-1) * hist                                                     # :84
    if not residual:
        return wc, agg
    g = torch.sigmoid(wc @ sd[p + 'gate_proj.weight'].t() + sd[p + 'gate_proj.bias'])  # :87
    out = layer_norm(g * wc + (1 - g) * hist, sd[p + 'layernorm.weight'], sd[p + 'layernorm.bias'])  # :88-89
    return out, agg


def graph_sage(sd, p, hist, user_nodes, n_src):
    """userEncoders.py:121,151-157 in closed form.  The node axis is
    cat[hist (H), user_node_embedding (config.batch_size)]; create_bipartite_graph (:91-98) gives
    every target node i < H the source nodes 0..n_src-1 *of the same row*, n_src = rows per forward,
    so  g[b,h] = lin_l(mean_{u<n_src} X[b,u]) + lin_r(X[b,h]).  PyG semantics: parity unpinned."""
    B, H, D = hist.shape
    X = torch.cat([hist, user_nodes.unsqueeze(0).expand(B, -1, -1)], dim=1)
    assert n_src <= X.shape[1], 'index error in the reference when rows > H + config.batch_size'
    m = X[:, :n_src].mean(dim=1)
    l = m @ sd[p + 'convs.0.lin_l.weight'].t() + sd[p + 'convs.0.lin_l.bias']
    return l.unsqueeze(1) + hist @ sd[p + 'convs.0.lin_r.weight'].t()


def kq_attention(sd, p, g, cand, attention_dim):
    """userEncoders.py:158-169: unmasked softmax over the history.  -> [B, N, D]."""
    K = g @ sd[p + 'K.weight'].t()                                                    # :161
    Q = cand @ sd[p + 'Q.weight'].t() + sd[p + 'Q.bias']                              # :162
    a = torch.einsum('bha,bna->bnh', K, Q) / math.sqrt(float(attention_dim))          # :163
    return torch.softmax(a, dim=-1) @ g                                               # :164-168


def remaining_lifetime_weighting(cfg, user, news, remaining):
    """RemainingLifetimeWeighting.forward, util.py:23-49."""
    base = (user * news).sum(dim=-1)
    if not cfg.use_remaining_lifetime_weighting:
        return base
    if cfg.use_expired_penalty:
        w = torch.sigmoid(cfg.sigmoid_scaling_alpha * remaining)
        w = (remaining >= 0).float() * w + (remaining < 0).float() * cfg.penalty_scaling_beta * w
    else:
        w = torch.sigmoid(cfg.sigmoid_scaling_alpha * remaining.abs())
    return base * w


def model_forward(sd, cfg, inputs, eval_shape=False, taps=None, grad=False):
    """Model.forward, model.py:151-187, for LIME-{CROWN,MHSA}-CROWN with the dot-product predictor.

    ``inputs``: the 26 tensors in signature order (dict or sequence).  ``eval_shape``: candidates
    arrive without the N axis and are unsqueezed (model.py:158-169).  Returns logits [B, N];
    intermediates are appended to ``taps`` when given.  ``grad``: record torch's autograd graph (the checker of the
    training step: gradients w.r.t. the ``sd`` tensors that require them).
    """
    v = list(inputs.values()) if isinstance(inputs, dict) else list(inputs)
    (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, _ute, user_content_text, _ucm, _uce,
     user_freshness, user_lifetime, user_history_mask, _g, _cm, _ci, news_category, news_subCategory, news_title_text,
     news_title_mask, _nte, news_content_text, _ncm, _nce, news_freshness, news_lifetime, remaining) = v
    if eval_shape:
        (news_category, news_subCategory, news_title_text, news_title_mask, news_content_text, news_freshness,
         news_lifetime, remaining) = [t.unsqueeze(1) for t in (
             news_category, news_subCategory, news_title_text, news_title_mask, news_content_text, news_freshness,
             news_lifetime, remaining)]
    ne = 'news_encoder.'
    ue = 'user_encoder.'
    with torch.set_grad_enabled(bool(grad)):
        cand = lime_news_encoder(sd, ne, cfg, news_title_text, news_title_mask, news_content_text, news_category,
                                 news_subCategory, news_freshness, news_lifetime, taps)              # model.py:171-173
        hist = lime_news_encoder(sd, ne, cfg, user_title_text, user_title_mask, user_content_text, user_category,
                                 user_subCategory, user_freshness, user_lifetime, taps)              # userEncoders.py:110-112
        cand_topic = topic_representation(sd, ne, news_category, news_subCategory)                   # :103-105
        if cfg.use_candidate_ware_clicked_news_attention:
            hist_topic = topic_representation(sd, ne, user_category, user_subCategory)               # :115-117
            hist2, agg = candidate_aware_attention(sd, ue + 'candidate_aware_attn.', hist, hist_topic, cand_topic,
                                                   user_history_mask, residual=cfg.use_residual_connection)  # :119
        else:
            hist2, agg = hist, None
        g = graph_sage(sd, ue + 'graph_sage.', hist2, sd[ue + 'user_node_embedding'], n_src=hist.shape[0])  # :121,:151-157
        user = kq_attention(sd, ue, g, cand, cfg.attention_dim)                                       # :158-169
        logits = remaining_lifetime_weighting(cfg, user, cand, remaining)                             # model.py:181
    if taps is not None:
        taps['hist_refined'] = hist2
        taps['attn_weights_agg'] = agg
        taps['gcn_feature'] = g
        taps['user_representation'] = user
        taps['news_representation'] = cand
        taps['logits'] = logits
    return logits


# ---------------------------------------------------------------------------------------------------
# Batch assembly: numpy restatement of the reference's datasets, pinned by
# tests/golden/dataset_*.npz captured from the imported Train_Dataset / DevTest_Dataset.
# ---------------------------------------------------------------------------------------------------
def _pad_history_list(values, H):
    """dataset.py:125-128 / :202-204: the LAST H entries, then zeros up to H (a longer list is truncated, not padded)."""
    values = list(values)
    return values[-H:] + [0] * max(0, H - len(values))


def _news_fields(corpus, index):
    """The eight per-news arrays of corpus.py:360-367 gathered by news index (dataset.py:131-140)."""
    index = np.asarray(index)
    return [corpus.news_category[index], corpus.news_subCategory[index], corpus.news_title_text[index],
            corpus.news_title_mask[index], corpus.news_title_entity[index], corpus.news_abstract_text[index],
            corpus.news_abstract_mask[index], corpus.news_abstract_entity[index]]


def _assemble(corpus, behaviors, rows, cand_index, cand_freshness, cand_lifetime, fr_slot, lt_slot):
    H = corpus.max_history_num
    C = corpus.config.category_num
    B = len(rows)
    user_id = np.asarray([behaviors[i][0] for i in rows], dtype=np.int64)
    hist_index = np.stack([np.asarray(behaviors[i][1]) for i in rows])
    hist_mask = np.stack([np.asarray(behaviors[i][2]) for i in rows])
    user_fr = np.asarray([_pad_history_list(behaviors[i][fr_slot], H) for i in rows], dtype=np.float32)
    user_lt = np.asarray([_pad_history_list(behaviors[i][lt_slot], H) for i in rows], dtype=np.float32)
    out = [user_id] + _news_fields(corpus, hist_index) + [user_fr, user_lt, hist_mask,
           np.zeros((B, H, H), dtype=np.float32), np.zeros((B, C + 1), dtype=bool), np.zeros((B, H), dtype=np.int64)]    # dataset.py:119-121
    out += _news_fields(corpus, cand_index) + [np.asarray(cand_freshness, dtype=np.float32), np.asarray(cand_lifetime, dtype=np.float32)]
    return out


def assemble_train(corpus, train_samples, train_freshness, train_user_topic_lifetime, rows):
    """Train_Dataset.__getitem__ (dataset.py:105-141), default-collated over `rows`.  The three sampled tables are what
    negative_sampling (dataset.py:42-77) produced on the host.  Returns the 25 arrays in the reference's order."""
    rows = list(rows)
    return _assemble(corpus, corpus.train_behaviors, rows, np.asarray(train_samples)[rows], np.asarray(train_freshness)[rows],
                     np.asarray(train_user_topic_lifetime)[rows], 9, 10)


def assemble_devtest(corpus, mode, rows):
    """DevTest_Dataset.__getitem__ (dataset.py:192-227): one candidate per row, candidate tensors without the N axis."""
    rows = list(rows)
    beh = corpus.dev_behaviors if mode == 'dev' else corpus.test_behaviors
    return _assemble(corpus, beh, rows, np.asarray([beh[i][3] for i in rows]), [beh[i][5] for i in rows],
                     [beh[i][6] for i in rows], 7, 8)
