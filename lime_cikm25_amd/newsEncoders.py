"""Drop-in news encoders of the scoring path: FreshnessEncoder, LIME, CROWN and MHSA
(reference newsEncoders.py:38-161, 167-373, 566-595, 806-828).

The modules keep the reference's attribute and parameter names, so ``state_dict()`` has the same
183 keys and reference checkpoints load.  Standard torch containers
(nn.Linear, nn.Embedding, nn.TransformerEncoder ...) are used as *parameter holders* with their
constructor-default initialisation, exactly as the reference leaves them; their ``forward`` is never
called.  All arithmetic runs in hand-written HIP kernels (lime_cikm25_amd.ops -> liblime_hip.so):

  word gather + positional table      fused into the A-operand fetch of the in_proj GEMM
  in_proj / out_proj / FFN            exact-fp32 MFMA GEMM; bias, ReLU, residual and LayerNorm in the epilogue
  token attention                     one wave per 32 query rows, scores held in MFMA accumulators
  mean pool / intent tail / buckets   small fixed-order kernels
"""
import math
import os

import torch
import torch.nn as nn
from torch.nn import TransformerEncoder, TransformerEncoderLayer

from . import ops
from .layers import Attention, MultiHeadAttention

# tokens encoded per pass of the token encoder: bounds the activation workspace (9.2 KB / token)
MAX_TOKENS_PER_PASS = 4 * 1024 * 1024

# Encode only what differs (encode_tokens_compact): all-padding sequences share one representative, in_proj runs over the live
# tokens only.  Exact (same arithmetic per row, nothing cached between forwards); LIME_DENSE_TOKENS=1 (or DEDUP = False) runs
# every token of every slot through the layer as the reference does -- the A/B switch behind bench.py's "dense" figures.
DEDUP = os.environ.get('LIME_DENSE_TOKENS', '0') != '1'


def _no_train_dropout(module, p):
    if module.training and p > 0:
        raise NotImplementedError('encode_flat() is the fused scoring kernel chain (eval-mode children, or dropout_rate = 0): with '
                                  'training-mode dropout active call the module (forward) or Model.forward, which take the '
                                  'differentiable path of lime_cikm25_amd.training')


def _flat_inputs(title_text, title_mask, content_text, category, subCategory):
    B, n = title_text.shape[0], title_text.shape[1]
    return (_i32(title_text).reshape(B * n, -1).contiguous(), title_mask.reshape(B * n, -1).contiguous(),
            _i32(content_text).reshape(B * n, -1).contiguous(), _i32(category).reshape(-1).contiguous(),
            _i32(subCategory).reshape(-1).contiguous())


_SIDE = {}


# Independent branches of the forward CAN be forked onto side streams (fork / join with wait_stream, which is also how
# the fork is recorded into the HIP graph): branch 0 = freshness encoder, 2 = candidate-aware attention weights (small,
# latency-bound kernels that otherwise sit on the critical path in front of the token encoders), 1 = title chain beside the
# body chain, 3 = the body encoder's preparation under the title encoder, 4 = category representation + stacked intent weights, 5 = the
# body half of the intent attention's hidden GEMM, 6 = the candidate side of the interest match (same-box A/B at config 2b, ms per
# step: {0,2} 2.414, {0,2,3} 2.372, {0,2,4} 2.446 -- short kernels in front of a persistent GEMM delay some of its statically
# scheduled workgroups --, {0,2,5,6} 2.415, all 2.466), 7 = on the compacted path, the title encoder's chain beside the body
# encoder's: the compacted launches have few tiles (423 / 251 / 502 on 512 workgroup slots for the title, a half-empty last round
# for the body), so the two chains fill each other's gaps ({0,2,3} 2.288, {0,2,3,7} 2.210; on the dense path, whose launches fill
# every slot for ten rounds, the same fork -- branch 1 -- measured slower; beside the forked title chain branch 4 now pays:
# {0,2,3,7} 2.242, {0,2,3,4,7} 2.206; three interleaved 300-step runs each: {0,2,3,4,7} 2.195, + 5 2.175, + 6 2.170, + {5,6} 2.155).  OVERLAP_BRANCHES is the set that is forked (LIME_OVERLAP_STREAMS = 0: none, 2: the two small branches --
# the default, 4.557 vs 4.593 ms --, 1: all three).  The title / body fork is off by default: the big GEMMs hold two
# workgroups of 256 VGPRs x 4 waves and 61 KB LDS on every CU, so nothing else becomes resident beside them and that fork
# measured slower (4.651 ms).
def _branches(spec):
    named = {'0': frozenset(), '1': frozenset((0, 1, 2, 3, 4, 5, 6, 7)), '2': frozenset((0, 2, 3, 4, 5, 6, 7))}
    return named[spec] if spec in named else frozenset(int(x) for x in spec.split('+'))       # e.g. LIME_OVERLAP_STREAMS=0+2+4


OVERLAP_BRANCHES = _branches(os.environ.get('LIME_OVERLAP_STREAMS', '2'))
SERIAL_STREAMS = False          # bench.py's instrumented pass forces everything onto one stream


def _side_stream(device, which=0):
    if SERIAL_STREAMS or which not in OVERLAP_BRANCHES:
        return torch.cuda.current_stream(device)
    key = (device.type, device.index, which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


_ZERO_IDS = {}


def _zero_ids(n, device):
    """n zero word ids (the padding word), allocated once per device and never written: a forward does not pay a fill launch for them."""
    key = (device.type, device.index)
    z = _ZERO_IDS.get(key)
    if z is None or z.numel() < n:
        z = _ZERO_IDS[key] = torch.zeros(max(n, 512), dtype=torch.int32, device=device)
    return z[:n]


def _cat_rows(ts):
    """torch.cat(ts, dim=0) -- as a VIEW when the pieces already sit back to back in one storage (the Model's packed
    graph inputs are laid out that way), so the per-forward concatenation of candidates and history costs no kernel."""
    t0 = ts[0]
    ok = all(t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] and
             t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in ts)
    if ok:
        end = t0.storage_offset()
        for t in ts:
            ok = ok and t.storage_offset() == end
            end += t.numel()
    if not ok:
        return torch.cat(ts, dim=0)
    rows = sum(t.shape[0] for t in ts)
    size = (rows,) + tuple(t0.shape[1:])
    stride = t0.stride() if t0.dim() > 1 else (1,)
    return t0.as_strided(size, stride, t0.storage_offset())


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def reference_bucket(x, num_buckets):
    """The reference's bucket rule (newsEncoders.py:53-58) as torch evaluates it in fp32 on the CPU."""
    x = torch.clamp(x.float(), min=1)
    scaled = torch.log(x) / torch.log(torch.tensor(60 * 60 * 24.0))
    return torch.clamp((scaled * (num_buckets / 7)).long(), max=num_buckets - 1)


def bucket_cut_points(num_buckets):
    """The num_buckets - 1 fp32 cut points of the bucket rule: the smallest float whose bucket is >= k, for k = 1 .. num_buckets - 1,
    found by bisection over the bit patterns of the positive floats with the rule's own fp32 evaluation on the CPU (the rule is
    monotone; a +-64-ulp window around every cut is re-checked).  Comparing against them reproduces the rule bit-exactly on the
    device without depending on any logf (for num_buckets = 10 this is the table built into lime_bucketize_f32)."""
    import numpy as np
    top = int(np.array([np.finfo(np.float32).max], dtype=np.float32).view(np.uint32)[0])
    one = int(np.array([1.0], dtype=np.float32).view(np.uint32)[0])
    f = lambda bits: reference_bucket(torch.from_numpy(np.array(bits, dtype=np.uint32).view(np.float32).copy()), num_buckets)
    cuts = []
    for k in range(1, num_buckets):
        lo, hi = one, top                                   # bucket(lo) = 0 < k <= bucket(hi)
        if int(f([hi])[0]) < k:
            raise ValueError('bucket %d is never reached with num_buckets = %d' % (k, num_buckets))
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if int(f([mid])[0]) >= k:
                hi = mid
            else:
                lo = mid
        win = list(range(max(one, hi - 64), min(top, hi + 64) + 1))
        b = f(win)
        if not bool(((b >= k) == (torch.tensor(win) >= hi)).all()):
            raise ValueError('the bucket rule is not monotone around its cut point %d (num_buckets = %d)' % (k, num_buckets))
        cuts.append(hi)
    return torch.from_numpy(np.array(cuts, dtype=np.uint32).view(np.float32).copy())


class FreshnessEncoder(nn.Module):
    """newsEncoders.py:38-83.  hidden = content dim always: ``fusion_method == 'add' or 'gated'`` is truthy (:42-45)."""

    def __init__(self, config, base_news_encoder):
        super().__init__()
        embedding_dim = config.freshness_embedding_dim
        hidden_dim = base_news_encoder.news_embedding_dim
        self.num_buckets = config.num_buckets
        # num_buckets = 10 (config.py:59): the cut points are built into lime_bucketize_f32; any other count: derived here from the
        # rule's own fp32 evaluation (a plain attribute, not a buffer: the state_dict keeps the reference's keys)
        self._cuts = None if self.num_buckets == 10 else bucket_cut_points(self.num_buckets)
        self.freshness_embedding = nn.Embedding(self.num_buckets, embedding_dim)
        self.lifetime_embedding = nn.Embedding(self.num_buckets, embedding_dim)
        self.dense = nn.Linear(embedding_dim * 2, hidden_dim)
        self.activation = nn.Tanh()

    def bucketize(self, x):
        """int64 like the reference (newsEncoders.py:53-58); computed by threshold comparison on the device."""
        return self.buckets(x.float()).long()

    def buckets(self, x):
        """int32 buckets of a flat fp32 tensor."""
        if self._cuts is not None and self._cuts.device != x.device:
            self._cuts = self._cuts.to(x.device)
        return ops.bucketize(x, self._cuts)

    def encode_flat(self, freshness, lifetime, out):
        """freshness / lifetime: [M] fp32; out: [M, hidden] (may be a view of a wider buffer)."""
        M = freshness.numel()
        E = self.freshness_embedding.embedding_dim
        fb = self.buckets(freshness)
        lb = self.buckets(lifetime)
        cat = torch.empty((M, 2 * E), dtype=torch.float32, device=out.device)
        ops.gather_rows(fb, self.freshness_embedding.weight, cat[:, :E])
        ops.gather_rows(lb, self.lifetime_embedding.weight, cat[:, E:])
        return ops.linear(cat, self.dense.weight, self.dense.bias, act='tanh', out=out)

    def forward(self, news_freshness, news_user_topic_lifetime):
        if news_freshness.dim() == 1:
            news_freshness = news_freshness.unsqueeze(1)
        if news_user_topic_lifetime.dim() == 1:
            news_user_topic_lifetime = news_user_topic_lifetime.unsqueeze(1)
        if news_freshness.shape != news_user_topic_lifetime.shape:
            news_user_topic_lifetime = news_user_topic_lifetime.expand_as(news_freshness)
        B, n = news_freshness.shape
        out = torch.empty((B * n, self.dense.out_features), dtype=torch.float32, device=news_freshness.device)
        self.encode_flat(news_freshness.float().contiguous().view(-1), news_user_topic_lifetime.float().contiguous().view(-1), out)
        return out.view(B, n, -1)


class LIME(nn.Module):
    """newsEncoders.py:87-161: content + freshness, fused by 'concat' + project (the default, 400 columns), 'add' or 'gated' (the
    content encoder's 900 columns)."""

    def __init__(self, config, base_news_encoder):
        super().__init__()
        self.final_dim = config.lime_output_dim
        self.category_embedding = nn.Embedding(config.category_num, config.category_embedding_dim)
        self.category_embedding.weight.requires_grad = False
        self.subCategory_embedding = nn.Embedding(config.subCategory_num, config.subCategory_embedding_dim)
        self.subCategory_embedding.weight.requires_grad = False
        self.category_affine = nn.Linear(config.category_embedding_dim + config.subCategory_embedding_dim,
                                         config.category_embedding_dim)
        self.base_news_encoder = base_news_encoder
        self.freshness_encoder = FreshnessEncoder(config, base_news_encoder)
        self.fusion_method = config.fusion_method       # 'concat' (default), 'add' or 'gated' (newsEncoders.py:99, :111-126)
        self.auxiliary_loss = getattr(base_news_encoder, 'auxiliary_loss', None)      # newsEncoders.py:100-103
        content_dim = self.base_news_encoder.news_embedding_dim
        freshness_dim = content_dim                                                   # newsEncoders.py:106-107
        if self.fusion_method == 'concat':
            self.output_dim = content_dim + freshness_dim
            if self.final_dim:
                self.project = nn.Linear(self.output_dim, self.final_dim)
                self.output_dim = self.final_dim
            else:
                self.project = nn.Identity()
        elif self.fusion_method == 'add':
            self.output_dim = content_dim
            self.project = nn.Identity()
        elif self.fusion_method == 'gated':
            self.gate = nn.Linear(content_dim + freshness_dim, content_dim)
            self.output_dim = content_dim
            self.project = nn.Identity()
        else:
            raise ValueError('Unknown fusion method: %s' % self.fusion_method)
        self.news_embedding_dim = self.output_dim

    def initialize(self):
        if hasattr(self.base_news_encoder, 'initialize'):
            self.base_news_encoder.initialize()
        nn.init.xavier_uniform_(self.freshness_encoder.dense.weight)
        nn.init.zeros_(self.freshness_encoder.dense.bias)
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)
        nn.init.uniform_(self.subCategory_embedding.weight, -0.1, 0.1)
        nn.init.xavier_uniform_(self.category_affine.weight)
        nn.init.zeros_(self.category_affine.bias)

    def encode_flat(self, title_text, title_mask, content_text, category, subCategory, freshness, lifetime):
        """Flat batch of M news -> [M, output_dim].  title_text [M, T], content_text [M, L] int32; the rest [M]."""
        M = title_text.shape[0]
        cdim = self.base_news_encoder.news_embedding_dim
        main = torch.cuda.current_stream()
        side = _side_stream(title_text.device)
        if self.fusion_method in ('add', 'gated'):                                   # newsEncoders.py:154-159
            fused = torch.empty((M, 2 * cdim), dtype=torch.float32, device=title_text.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.freshness_encoder.encode_flat(freshness, lifetime, fused[:, cdim:])
            self.base_news_encoder.encode_flat(title_text, title_mask, content_text, category, subCategory, fused[:, :cdim])
            main.wait_stream(side)
            gate = ops.linear(fused, self.gate.weight, self.gate.bias, act='sigmoid') if self.fusion_method == 'gated' else None
            return ops.fuse_rows(fused[:, :cdim], fused[:, cdim:], gate)
        if isinstance(self.project, nn.Identity):
            fused = torch.empty((M, 2 * cdim), dtype=torch.float32, device=title_text.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.freshness_encoder.encode_flat(freshness, lifetime, fused[:, cdim:])
            self.base_news_encoder.encode_flat(title_text, title_mask, content_text, category, subCategory, fused[:, :cdim])
            main.wait_stream(side)
            return fused
        # project(cat(content, fresh)) = content W_c^T + (fresh W_f^T + b), and fresh = tanh(dense(cat(E_f[b1], E_l[b2]))) takes one of
        # num_buckets^2 values: the freshness half of `project` is a [100, 400] table G per forward (three GEMMs on 10 / 100-row
        # operands instead of one on M rows), gathered into the content GEMM as a residual by the bucket pair (newsEncoders.py:60-83,
        # :151-153; same sums, associated per half).  The branch depends on the inputs' buckets and the weights only: side stream.
        fe = self.freshness_encoder
        E, nb = fe.freshness_embedding.embedding_dim, fe.num_buckets
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pair = torch.add(fe.buckets(lifetime), fe.buckets(freshness), alpha=nb)                        # b_f * nb + b_l, int32 [M]
            t_f, t_l = ops.linear_group([dict(a=fe.freshness_embedding.weight, w=fe.dense.weight[:, :E], bias=None),       # [nb, cdim]
                                         dict(a=fe.lifetime_embedding.weight, w=fe.dense.weight[:, E:], bias=fe.dense.bias)])   # [nb, cdim]
            fresh = torch.tanh(t_f.unsqueeze(1) + t_l.unsqueeze(0)).view(nb * nb, -1)                     # row b_f * nb + b_l
            table = ops.linear(fresh, self.project.weight[:, cdim:], self.project.bias)                   # [nb^2, final_dim]
        content = torch.empty((M, cdim), dtype=torch.float32, device=title_text.device)
        self.base_news_encoder.encode_flat(title_text, title_mask, content_text, category, subCategory, content)
        main.wait_stream(side)
        return ops.linear(content, self.project.weight[:, :cdim], None, res=table, res_ids=pair)         # newsEncoders.py:152-153

    # ---- per-news content cache (eval: a news occurs in many impressions, its token encoders need to run once) ----------
    def build_content_cache(self, title_text, title_mask, content_text, category, subCategory, rows_per_pass=8192):
        """[n_news, output_dim]: the content half of every news pushed through its half of `project`.

        LIME's representation is project(cat(content(news), freshness(freshness, lifetime))) (newsEncoders.py:146-153) and
        project is linear, so  rep = content . W[:, :c]^T  +  freshness . W[:, c:]^T + b : the first term depends on the news
        alone (all the token-encoder work), the second on the occurrence (two bucket lookups and two small GEMMs).  The
        cache holds the first term per news id; ``encode_cached`` adds the second.  Rebuild it when weights change."""
        if self.fusion_method != 'concat':
            raise NotImplementedError("the per-news content cache splits `project` into a content and a freshness half: fusion_method "
                                      "'concat' only (got %r); score with Model.forward / util.compute_scores" % self.fusion_method)
        n = title_text.shape[0]
        cdim = self.base_news_encoder.news_embedding_dim
        dev = title_text.device
        ident = isinstance(self.project, nn.Identity)
        cache = torch.empty((n, cdim if ident else self.project.out_features), dtype=torch.float32, device=dev)
        for r0 in range(0, n, rows_per_pass):
            r1 = min(n, r0 + rows_per_pass)
            content = torch.empty((r1 - r0, cdim), dtype=torch.float32, device=dev)
            self.base_news_encoder.encode_flat(_i32(title_text[r0:r1]).contiguous(), title_mask[r0:r1].contiguous(),
                                               _i32(content_text[r0:r1]).contiguous(), _i32(category[r0:r1]).contiguous(),
                                               _i32(subCategory[r0:r1]).contiguous(), content)
            if ident:
                cache[r0:r1] = content
            else:
                ops.linear(content, self.project.weight[:, :cdim], None, out=cache[r0:r1])
        return cache

    def encode_cached(self, cache, news_index, freshness, lifetime):
        """Representations of the occurrences (news_index[r], freshness[r], lifetime[r]) -> [R, output_dim]."""
        cdim = self.base_news_encoder.news_embedding_dim
        R = news_index.numel()
        idx = _i32(news_index.reshape(-1)).contiguous()
        fresh = torch.empty((R, cdim), dtype=torch.float32, device=cache.device)
        self.freshness_encoder.encode_flat(freshness.float().reshape(-1).contiguous(), lifetime.float().reshape(-1).contiguous(), fresh)
        if isinstance(self.project, nn.Identity):
            return torch.cat([cache[idx.long()], fresh], dim=1)
        return ops.linear(fresh, self.project.weight[:, cdim:], self.project.bias, res=cache, res_ids=idx)

    def encode_many(self, groups):
        """Encode several [B, n, ...] groups (candidates, history) in ONE pass over the kernels.

        Each group: (title_text, title_mask, content_text, category, subCategory, freshness, lifetime).
        Returns one [B, n, output_dim] tensor per group.
        """
        shapes, flat = [], [[] for _ in range(7)]
        for (tt, tm, ct, cat, sub, fr, lt) in groups:
            B, n = tt.shape[0], tt.shape[1]
            shapes.append((B, n))
            if fr.dim() == 1:
                fr = fr.unsqueeze(1)
            if lt.dim() == 1:
                lt = lt.unsqueeze(1)
            if lt.shape != fr.shape:
                lt = lt.expand_as(fr)
            for dst, t in zip(flat, (_i32(tt).reshape(B * n, -1), tm.reshape(B * n, -1), _i32(ct).reshape(B * n, -1),
                                     _i32(cat).reshape(-1), _i32(sub).reshape(-1), fr.float().reshape(-1), lt.float().reshape(-1))):
                dst.append(t)
        cat_all = [t[0].contiguous() if len(t) == 1 else _cat_rows(t) for t in flat]
        out = self.encode_flat(*cat_all)
        res, r0 = [], 0
        for (B, n) in shapes:
            res.append(out[r0:r0 + B * n].view(B, n, -1))
            r0 += B * n
        return res

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory,
                user_embedding, news_freshness=None, news_user_topic_lifetime=None):
        """newsEncoders.py:140-161 -> [B, n, output_dim].  In training mode (autograd recording or dropout active) the call takes
        the differentiable path (``training.news_flat``): gradients reach every parameter the reference's do."""
        from . import training
        if training.wants_train_path(self, getattr(self.base_news_encoder, 'dropout_rate', 0.0)):
            B, n = title_text.shape[0], title_text.shape[1]
            fr, lt = news_freshness, news_user_topic_lifetime
            fr = fr.unsqueeze(1) if fr.dim() == 1 else fr
            lt = lt.unsqueeze(1) if lt.dim() == 1 else lt
            lt = lt.expand_as(fr) if lt.shape != fr.shape else lt
            rep = training.news_flat(self, *_flat_inputs(title_text, title_mask, content_text, category, subCategory),
                                     fr.float().reshape(-1).contiguous(), lt.float().reshape(-1).contiguous())
            return rep.view(B, n, -1)
        return self.encode_many([(title_text, title_mask, content_text, category, subCategory, news_freshness,
                                  news_user_topic_lifetime)])[0]


class NewsEncoder(nn.Module):
    """newsEncoders.py:167-225: shared tables.  The word table is filled by the caller (``load_state_dict`` or
    ``word_embedding.weight.data.copy_``); the reference unpickles it from the cwd at :173-174."""

    def __init__(self, config):
        super().__init__()
        self.word_embedding_dim = config.word_embedding_dim
        self.category_num = config.category_num
        self.word_embedding = nn.Embedding(num_embeddings=config.vocabulary_size, embedding_dim=self.word_embedding_dim)
        self.category_embedding = nn.Embedding(num_embeddings=config.category_num, embedding_dim=config.category_embedding_dim)
        self.category_embedding.weight.requires_grad = False
        self.subCategory_embedding = nn.Embedding(num_embeddings=config.subCategory_num,
                                                  embedding_dim=config.subCategory_embedding_dim)
        self.subCategory_embedding.weight.requires_grad = False
        self.dropout_rate = config.dropout_rate
        self.dropout = nn.Dropout(p=config.dropout_rate, inplace=True)
        self.dropout_ = nn.Dropout(p=config.dropout_rate, inplace=False)
        self.auxiliary_loss = None
        self.affine = nn.Linear(config.word_embedding_dim, config.word_embedding_dim, bias=True)   # unused upstream too

    def initialize(self):
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)
        nn.init.uniform_(self.subCategory_embedding.weight, -0.1, 0.1)
        nn.init.zeros_(self.subCategory_embedding.weight[0])
        nn.init.xavier_uniform_(self.affine.weight)
        nn.init.zeros_(self.affine.bias)

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory,
                user_embedding, news_freshness=None, news_user_topic_lifetime=None):
        B, n = title_text.shape[0], title_text.shape[1]
        flat = _flat_inputs(title_text, title_mask, content_text, category, subCategory)
        from . import training
        if training.wants_train_path(self, self.dropout_rate):
            return training.content_flat(self, *flat).view(B, n, -1)
        out = torch.empty((B * n, self.news_embedding_dim), dtype=torch.float32, device=title_text.device)
        self.encode_flat(*flat, out)
        return out.view(B, n, -1)


class PositionalEncoding(nn.Module):
    """newsEncoders.py:806-828: the sinusoid table, a registered buffer (part of the state_dict)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))

    def table(self):
        return self.pe[0]


class _MAB(nn.Module):
    """Parameter holder for newsEncoders.py:395-422 (never called on the scoring path)."""

    def __init__(self, dim_Q, dim_K, dim_V, num_heads, ln=False):
        super().__init__()
        self.fc_q = nn.Linear(dim_Q, dim_V)
        self.fc_k = nn.Linear(dim_K, dim_V)
        self.fc_v = nn.Linear(dim_K, dim_V)
        if ln:
            self.ln0 = nn.LayerNorm(dim_V)
            self.ln1 = nn.LayerNorm(dim_V)
        self.fc_o = nn.Linear(dim_V, dim_V)


class ISAB(nn.Module):
    """Parameter holder: the reference constructs ISAB (newsEncoders.py:250-254) and never calls it (:325-333 are
    commented out), but its 25 tensors are part of the checkpoint."""

    def __init__(self, dim_in, dim_out, num_heads, num_inds, ln=False):
        super().__init__()
        self.I = nn.Parameter(torch.Tensor(1, num_inds, dim_out))
        nn.init.xavier_uniform_(self.I)
        self.mab0 = _MAB(dim_out, dim_in, dim_out, num_heads, ln=ln)
        self.mab1 = _MAB(dim_in, dim_out, dim_out, num_heads, ln=ln)


class CategoryPredictor(nn.Module):
    """Parameter holder for newsEncoders.py:375-393: under LIME the auxiliary loss is dead."""

    def __init__(self, title_embedding, category_num):
        super().__init__()
        self.fc = nn.Linear(title_embedding, category_num)


def encode_tokens(ids, table, pe, transformer, nhead, pooled_out=None):
    """Word gather + positional table + the post-LN encoder layer(s) of newsEncoders.py:311-320.

    ids: [M, S] int32 (every id must be in [0, V): unchecked, as on nn.Embedding's device path);
    returns the layer output [M * S, E].  Five launches per layer, activations stay fp32.
    With ``pooled_out`` ([M, E]) the token mean pooling of :317 / :321 is taken in the last GEMM's epilogue (pool32: means
    over 32-token blocks; a longer sequence is finished by a mean over its S / 32 block rows) and the layer output never
    reaches HBM; returns None then.
    """
    M, S = ids.shape
    E = table.shape[1]
    hd = E // nhead
    hs = 32 if hd <= 32 else hd                    # heads padded to 32 columns in the in_proj output: the 128 x 320 GEMM
    W = nhead * hs                                 # tiles compute those columns anyway, and attention gets 16-byte loads
    flat = ids.reshape(-1)
    x = None
    for li, layer in enumerate(transformer.layers):
        sa = layer.self_attn
        w_in = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, hs) if hs != hd else sa.in_proj_weight
        b_in = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, hs) if hs != hd else sa.in_proj_bias
        if li == 0:
            # (E[ids] + PE) W^T + b = E[ids] W^T + (PE W^T + b)[t]: the positional term is an [S, 3W] table added as a
            # periodic residual, so the A operand is a pure row gather (which the LDS-DMA GEMM can stage directly)
            pew = ops.linear(pe[:S], w_in, b_in)
            qkv = ops.linear(table, w_in, None, a_ids=flat, res=pew, res_mod=S, n_alg=3 * E)
        else:
            qkv = ops.linear(x, w_in, b_in, n_alg=3 * E)
        attn = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, 1.0 / math.sqrt(hd),
                                   head_stride=hs)
        ln1 = (layer.norm1.weight, layer.norm1.bias)
        if li == 0:
            x1 = ops.linear(attn, sa.out_proj.weight, sa.out_proj.bias, res=table, res_ids=flat, res_pe=pe, res_period=S,
                            ln=ln1, ln_eps=layer.norm1.eps)
        else:
            x1 = ops.linear(attn, sa.out_proj.weight, sa.out_proj.bias, res=x, ln=ln1, ln_eps=layer.norm1.eps)
        h = ops.linear(x1, layer.linear1.weight, layer.linear1.bias, act='relu')
        last = li == len(transformer.layers) - 1
        if last and pooled_out is not None and transformer.norm is None and S % 32 == 0 and M * S >= 4096:
            blocks = ops.linear(h, layer.linear2.weight, layer.linear2.bias, res=x1, ln=(layer.norm2.weight, layer.norm2.bias),
                                ln_eps=layer.norm2.eps, pool32=True, out=pooled_out if S == 32 else None)
            if S != 32:
                ops.mean_pool(blocks, M, S // 32, out=pooled_out)
            return None
        x = ops.linear(h, layer.linear2.weight, layer.linear2.bias, res=x1, ln=(layer.norm2.weight, layer.norm2.bias),
                       ln_eps=layer.norm2.eps)
    if transformer.norm is not None:
        raise NotImplementedError('a final encoder norm is not used by the reference (newsEncoders.py:245,247)')
    if pooled_out is not None:
        ops.mean_pool(x, M, S, out=pooled_out)
        return None
    return x


def encode_tokens_compact(ids, table, pe, transformer, nhead, pooled_out):
    """``encode_tokens(..., pooled_out)`` without the repetitions of a padded batch (csrc/compact.hip): the history slots of an
    impression are padded with the all-zero <PAD> news (corpus.py:476-477, dataset.py:105-141) and every text with the padding
    word behind it; the reference encodes them all (newsEncoders.py:311-321).

      * sequence level: an all-padding sequence pools to the same vector wherever it stands (no mask, no cross-sequence term in
        the layer), so the live sequences + ONE all-padding representative go through the layer (n_c = live + 1 sequences) and
        ``pooled_out[s] = pooled_c[seq_inv[s]]``;
      * token level: the in_proj row of a padding token is (E[0] + PE[t]) W^T + b, a function of its position alone: S table
        rows computed once; in_proj runs over the live tokens (rows scattered to their compact positions) and attention looks
        every token's q / k / v row up through ``row_map``.  From the attention output on every position is distinct.

    Counts live in device memory (lime_linear_args.m_dev / n_seq_dev), buffers have their full-batch size: the forward stays one
    HIP graph.  Same kernels, same per-row arithmetic as the dense path; results differ from it only through the S padding rows
    coming from the small-M GEMM kernel (different k order, ~1e-7).  Returns None (the result is ``pooled_out``).
    """
    return compact_run(compact_prepare(ids, table, pe, transformer, nhead), table, pe, transformer, nhead, pooled_out)


def compact_prepare(ids, table, pe, transformer, nhead):
    """Everything of ``encode_tokens_compact`` in front of the big GEMMs -- index lists, padded in_proj weights, the positional
    table through in_proj, the S rows shared by the padding tokens: a dozen short launches that depend on the ids and the
    weights only, so the caller can run them on a side stream under another encoder's GEMMs."""
    return compact_prepare_many([(ids, pe, transformer)], table, nhead)[0]


def compact_prepare_many(encoders, table, nhead):
    """``compact_prepare`` for several token encoders over one word table ((ids, pe, transformer) each): their positional tables go
    through in_proj in ONE grouped launch, their padding rows in another (independent, latency-bound GEMMs)."""
    E = table.shape[1]
    hd = E // nhead
    W = nhead * 32
    parts = []
    for ids, pe, transformer in encoders:
        M, S = ids.shape
        sa = transformer.layers[0].self_attn
        cap = (M + 1) * S
        cmp = ops.compact_sequences(ids)                                       # pad rows live at qkv[cap : cap + S]
        w_in = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, 32)
        b_in = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, 32)
        qkv = torch.empty((cap + S, 3 * W), dtype=torch.float32, device=ids.device)
        parts.append((cmp, w_in, b_in, qkv, pe, S, cap))
    pews = ops.linear_group([dict(a=pe[:S], w=w_in, bias=b_in) for (_, w_in, b_in, _, pe, S, _) in parts])   # [S, 3W]: positional term + bias
    ops.linear_group([dict(a=table, w=w_in, bias=None, a_ids=_zero_ids(S, table.device), res=pew, res_mod=S, out=qkv[cap:])
                      for (_, w_in, _, qkv, _, S, cap), pew in zip(parts, pews)])                             # the S padding rows
    return [(cmp, w_in, pew, qkv) for (cmp, w_in, _, qkv, _, _, _), pew in zip(parts, pews)]


_IDENT = {}


def _identity_rows(n, device):
    """arange(n) int32, allocated once per device: the row map of an encoder layer behind the first (every compact row is its own)."""
    key = (device.type, device.index)
    z = _IDENT.get(key)
    if z is None or z.numel() < n:
        z = _IDENT[key] = torch.arange(max(n, 4096), dtype=torch.int32, device=device)
    return z[:n]


def compact_run(prep, table, pe, transformer, nhead, pooled_out):
    cmp, w_in, pew, qkv = prep
    M, S, cap = cmp.n_seq, cmp.S, cmp.cap
    E = table.shape[1]
    hd = E // nhead
    W = nhead * 32
    n_layers = len(transformer.layers)
    x = None
    for li, layer in enumerate(transformer.layers):
        sa = layer.self_attn
        ln1 = (layer.norm1.weight, layer.norm1.bias)
        if li == 0:
            ops.linear(table, w_in, None, a_ids=cmp.tok_ids, res=pew, res_mod=S, out=qkv[:cap], m_dev=cmp.n_live_tokens,
                       c_ids=cmp.tok_rows, n_alg=3 * E)                                                    # live tokens only
            attn = ops.token_attention_rows(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], cmp.row_map, cmp.n_compact, M + 1, S, nhead, hd,
                                            1.0 / math.sqrt(hd))
            x1 = ops.linear(attn, sa.out_proj.weight, sa.out_proj.bias, res=table, res_ids=cmp.ids_c, res_pe=pe, res_period=S,
                            ln=ln1, ln_eps=layer.norm1.eps, m_dev=cmp.n_rows)
        else:
            # a layer behind the first (config.py:70 allows num_layers = 2): its input rows are all distinct, so only the sequence-level
            # sharing is left -- the layer runs over the n_rows compact rows (device-side count), attention through the identity map
            w_l = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, 32)
            b_l = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, 32)
            qkv_l = ops.linear(x, w_l, b_l, m_dev=cmp.n_rows, n_alg=3 * E)
            attn = ops.token_attention_rows(qkv_l[:, :W], qkv_l[:, W:2 * W], qkv_l[:, 2 * W:], _identity_rows(cap, x.device), cmp.n_compact,
                                            M + 1, S, nhead, hd, 1.0 / math.sqrt(hd))
            x1 = ops.linear(attn, sa.out_proj.weight, sa.out_proj.bias, res=x, ln=ln1, ln_eps=layer.norm1.eps, m_dev=cmp.n_rows)
        h = ops.linear(x1, layer.linear1.weight, layer.linear1.bias, act='relu', m_dev=cmp.n_rows)
        if li + 1 < n_layers:
            x = ops.linear(h, layer.linear2.weight, layer.linear2.bias, res=x1, ln=(layer.norm2.weight, layer.norm2.bias),
                           ln_eps=layer.norm2.eps, m_dev=cmp.n_rows)
            continue
        blocks = ops.linear(h, layer.linear2.weight, layer.linear2.bias, res=x1, ln=(layer.norm2.weight, layer.norm2.bias),
                            ln_eps=layer.norm2.eps, pool32=True, m_dev=cmp.n_rows)                     # [cap / 32, E] block means
    pooled_c = blocks if S == 32 else ops.mean_pool(blocks, M + 1, S // 32, n_seq_dev=cmp.n_compact)
    ops.gather_rows(cmp.seq_inv, pooled_c, pooled_out)
    return None


def compact_applicable(ids, table, transformer, nhead):
    """The compacted path covers the shapes of the big-M kernels: post-LN layers without a final norm, head_dim <= 32,
    S a multiple of 32 that the row-map attention is built for, at least 4096 token rows, 16-byte aligned table rows."""
    M, S = ids.shape
    E = table.shape[1]
    return (DEDUP and len(transformer.layers) >= 1 and transformer.norm is None and E % nhead == 0 and E // nhead <= 32 and
            E % 4 == 0 and S % 32 == 0 and (S // 32 <= 4 or S // 32 in (8, 16)) and (M + 1) * S >= 4096 and
            ids.dtype == torch.int32 and ids.is_contiguous())


FUSED_FFN = os.environ.get('LIME_BF16_FUSED_FFN', '1') != '0'        # 0: linear1 / linear2 as two lime_linear_bf16 launches (A/B runs)
FUSED_BLOCK = FUSED_FFN and os.environ.get('LIME_BF16_FUSED_BLOCK', '1') != '0'     # 0: out_proj + norm1 as its own launch


def _inproj_applicable(N, K):
    """lime_inproj_bf16: q / k / v columns in passes of 320 (ten heads padded to 32), K <= 320."""
    return FUSED_FFN and N % 320 == 0 and K <= 320 and K % 8 == 0


def _ffn_fused_applicable(layer, E, EP):
    """lime_encoder_ffn_bf16 is built for the reference's encoder shape: E = 300 carried as 304, hidden width a multiple of 128."""
    return (FUSED_FFN and EP == ops.ffn_model_columns() and EP - 15 <= E < EP and layer.linear1.out_features % 128 == 0 and
            layer.linear1.bias is not None and layer.linear2.bias is not None)


def encode_tokens_bf16(ids, table_bf16, pe, transformer, nhead, pooled_out):
    """encode_tokens + mean pool on the bf16 matrix cores (BASELINE config 3).

    table_bf16: the word table converted with ``ops.to_bf16`` ([V, E rounded up to 8], zero padded).  Activations between
    the launches are bf16; accumulation, bias, residual adds, softmax and LayerNorm are fp32.  E = 300 is carried as 304
    columns (zero weights / bias / gamma / beta in the pad, so the pad stays exactly zero through the layer).
    """
    M, S = ids.shape
    EP = table_bf16.shape[1]
    E = pe.shape[1]
    hd = E // nhead
    if hd > 32 or S % 32 != 0:
        raise NotImplementedError('the bf16 encoder path needs head_dim <= 32 and S a multiple of 32 (got %d, %d)' % (hd, S))
    W = nhead * 32
    flat = ids.reshape(-1)
    dev = ids.device
    padv = lambda v: torch.cat([v, v.new_zeros(EP - E)])
    x = None
    for li, layer in enumerate(transformer.layers):
        sa = layer.self_attn
        w_in = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, 32)                  # fp32 [3W, E]
        b_in = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, 32)
        if _inproj_applicable(3 * W, EP):
            # activation-stationary q / k / v projection (csrc/inproj_bf16.hip): the tile is read once for all 3 W columns
            w_in_p = ops.inproj_pack_bf16(w_in, EP)
            qkv = torch.empty((M * S, 3 * W), dtype=torch.bfloat16, device=dev)
            if li == 0:
                ops.inproj_bf16(table_bf16, w_in_p, ops.linear(pe[:S], w_in, b_in), 3 * W, qkv, a_ids=flat)     # + positional term + bias
            else:
                ops.inproj_bf16(x, w_in_p, b_in.view(1, -1), 3 * W, qkv)
        elif li == 0:
            pew = ops.linear(pe[:S], w_in, b_in)                                     # fp32 [S, 3W]: positional term + bias
            qkv = ops.linear_bf16(table_bf16, ops.to_bf16(w_in, cols_out=EP), None, a_ids=flat, res=pew, res_kind=1, res_mod=S, n_alg=3 * E, k_alg=E)
        else:
            qkv = ops.linear_bf16(x, ops.to_bf16(w_in, cols_out=EP), b_in, n_alg=3 * E, k_alg=E)
        attn = ops.token_attention_bf16(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, 1.0 / math.sqrt(hd), out_cols=EP)
        last = li == len(transformer.layers) - 1
        pool = last and transformer.norm is None              # token mean pooling in the epilogue: fp32 means over 32-token blocks
        if li == 0:
            pe_p = torch.cat([pe[:S], pe.new_zeros(S, EP - E)], dim=1)
        if FUSED_BLOCK and _ffn_fused_applicable(layer, E, EP) and E % 4 == 0:
            # everything behind the attention core in ONE launch (csrc/ffn_bf16.hip): out_proj + residual + norm1 written into the
            # stationary LDS tile the feed-forward half then reads
            w1p, w2p = ops.ffn_pack_bf16(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight)
            res_kw = (dict(res=table_bf16, res_kind=2, res_ids=flat, add_rows=pe[:S] + sa.out_proj.bias) if li == 0 else
                      dict(res=x, res_kind=3, add_rows=sa.out_proj.bias.view(1, E)))
            y = ops.encoder_block_bf16(attn, ops.oproj_pack_bf16(sa.out_proj.weight), ln1=(layer.norm1.weight, layer.norm1.bias),
                                       ln1_eps=layer.norm1.eps, w1p=w1p, w2p=w2p, b2=layer.linear2.bias, ln2=(layer.norm2.weight, layer.norm2.bias),
                                       ln2_eps=layer.norm2.eps, E=E, pool32=pool, **res_kw)
            if pool:
                return ops.mean_pool(y[:, :E], M, S // 32, out=pooled_out)
            x = y
            continue
        w_o = ops.to_bf16(sa.out_proj.weight, rows_out=EP, cols_out=EP)
        ln1 = (padv(layer.norm1.weight), padv(layer.norm1.bias))
        if li == 0:
            x1 = ops.linear_bf16(attn, w_o, padv(sa.out_proj.bias), res=table_bf16, res_kind=2, res_ids=flat, res_pe=pe_p,
                                 res_period=S, ln=ln1, ln_eps=layer.norm1.eps, ln_count=E, n_alg=E, k_alg=E)
        else:
            x1 = ops.linear_bf16(attn, w_o, padv(sa.out_proj.bias), res=x, res_kind=3, ln=ln1, ln_eps=layer.norm1.eps, ln_count=E, n_alg=E, k_alg=E)
        if _ffn_fused_applicable(layer, E, EP):
            # linear1 + ReLU + linear2 + residual + norm2 in ONE launch: the hidden state stays in registers (csrc/ffn_bf16.hip)
            w1p, w2p = ops.ffn_pack_bf16(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight)
            y = ops.encoder_ffn_bf16(x1, w1p, w2p, layer.linear2.bias, (layer.norm2.weight, layer.norm2.bias), layer.norm2.eps, E, pool32=pool)
            if pool:
                return ops.mean_pool(y[:, :E], M, S // 32, out=pooled_out)
            x = y
            continue
        h = ops.linear_bf16(x1, ops.to_bf16(layer.linear1.weight, cols_out=EP), layer.linear1.bias, act='relu', k_alg=E)
        if pool:
            blocks = ops.linear_bf16(h, ops.to_bf16(layer.linear2.weight, rows_out=EP), padv(layer.linear2.bias), res=x1, res_kind=3,
                                     ln=(padv(layer.norm2.weight), padv(layer.norm2.bias)), ln_eps=layer.norm2.eps, ln_count=E,
                                     pool32=True, n_alg=E)
            return ops.mean_pool(blocks[:, :E], M, S // 32, out=pooled_out)
        x = ops.linear_bf16(h, ops.to_bf16(layer.linear2.weight, rows_out=EP), padv(layer.linear2.bias), res=x1, res_kind=3,
                            ln=(padv(layer.norm2.weight), padv(layer.norm2.bias)), ln_eps=layer.norm2.eps, ln_count=E, n_alg=E)
    if transformer.norm is not None:
        raise NotImplementedError('a final encoder norm is not used by the reference (newsEncoders.py:245,247)')
    return ops.mean_pool_bf16(x, M, S, E, out=pooled_out)


def encode_tokens_bf16_compact(ids, table_bf16, pe, transformer, nhead, pooled_out, shared=None):
    """``encode_tokens_bf16`` on the compacted batch (see ``encode_tokens_compact``): live sequences + one all-padding
    representative through the layer, in_proj over the live tokens, the row-map bf16 attention.  The S padding rows come from
    the same bf16 GEMM kernel as the live rows (lime_linear_bf16 takes any M)."""
    M, S = ids.shape
    EP = table_bf16.shape[1]
    E = pe.shape[1]
    hd = E // nhead
    W = nhead * 32
    layer = transformer.layers[0]
    sa = layer.self_attn
    dev = ids.device
    cap = (M + 1) * S
    padv = lambda v: torch.cat([v, v.new_zeros(EP - E)])
    cmp = ops.compact_sequences(ids)
    fused = FUSED_BLOCK and _ffn_fused_applicable(layer, E, EP) and E % 4 == 0 and _inproj_applicable(3 * W, EP)
    if fused and shared is not None and 'w_in_p' in shared:
        # the passes of one forward over the same encoder share the packed weights (`shared`: a dict that lives for one forward)
        w_in_p, pew, w0p, w1p, w2p, add_rows = (shared[k] for k in ('w_in_p', 'pew', 'w0p', 'w1p', 'w2p', 'add_rows'))
    else:
        w_in = ops.pad_heads(sa.in_proj_weight, 3 * nhead, hd, 32)
        b_in = ops.pad_heads(sa.in_proj_bias, 3 * nhead, hd, 32)
        pew = ops.linear(pe[:S], w_in, b_in)                                   # fp32 [S, 3W]
        if fused:
            w_in_p = ops.inproj_pack_bf16(w_in, EP)
            w1p, w2p = ops.ffn_pack_bf16(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight)
            w0p, add_rows = ops.oproj_pack_bf16(sa.out_proj.weight), pe[:S] + sa.out_proj.bias
            if shared is not None:
                shared.update(w_in_p=w_in_p, pew=pew, w0p=w0p, w1p=w1p, w2p=w2p, add_rows=add_rows)
    qkv = torch.empty((cap + S, 3 * W), dtype=torch.bfloat16, device=dev)
    if fused:
        # three launches: in_proj (the live tokens and, behind them in the token list, the S padding rows the row map points at),
        # attention, and everything behind it
        ops.inproj_bf16(table_bf16, w_in_p, pew, 3 * W, qkv, a_ids=cmp.tok_ids, c_ids=cmp.tok_rows, m_dev=cmp.n_tokens_and_pad_rows)
        attn = ops.token_attention_rows_bf16(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], cmp.row_map, cmp.n_compact, M + 1, S, nhead, hd,
                                             1.0 / math.sqrt(hd), out_cols=EP)
        blocks = ops.encoder_block_bf16(attn, w0p, add_rows, (layer.norm1.weight, layer.norm1.bias), layer.norm1.eps, res=table_bf16,
                                        res_kind=2, res_ids=cmp.ids_c, w1p=w1p, w2p=w2p, b2=layer.linear2.bias,
                                        ln2=(layer.norm2.weight, layer.norm2.bias), ln2_eps=layer.norm2.eps, E=E, pool32=True,
                                        m_dev=cmp.n_rows)                       # fp32 [cap / 32, EP] block means
        pooled_c = ops.mean_pool(blocks[:, :E], M + 1, S // 32, n_seq_dev=cmp.n_compact)
        ops.gather_rows(cmp.seq_inv, pooled_c, pooled_out)
        return None
    if _inproj_applicable(3 * W, EP):
        ops.inproj_bf16(table_bf16, ops.inproj_pack_bf16(w_in, EP), pew, 3 * W, qkv, a_ids=cmp.tok_ids, c_ids=cmp.tok_rows,
                        m_dev=cmp.n_tokens_and_pad_rows)
    else:
        zeros = _zero_ids(S, dev)
        w_in_b = ops.to_bf16(w_in, cols_out=EP)
        ops.linear_bf16(table_bf16, w_in_b, None, a_ids=zeros, res=pew, res_kind=1, res_mod=S, out=qkv[cap:])
        ops.linear_bf16(table_bf16, w_in_b, None, a_ids=cmp.tok_ids, res=pew, res_kind=1, res_mod=S, out=qkv[:cap], m_dev=cmp.n_live_tokens,
                        c_ids=cmp.tok_rows, n_alg=3 * E, k_alg=E)
    attn = ops.token_attention_rows_bf16(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], cmp.row_map, cmp.n_compact, M + 1, S, nhead, hd,
                                         1.0 / math.sqrt(hd), out_cols=EP)
    pe_p = torch.cat([pe[:S], pe.new_zeros(S, EP - E)], dim=1)
    if FUSED_BLOCK and _ffn_fused_applicable(layer, E, EP) and E % 4 == 0:
        w1p, w2p = ops.ffn_pack_bf16(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight)
        blocks = ops.encoder_block_bf16(attn, ops.oproj_pack_bf16(sa.out_proj.weight), pe[:S] + sa.out_proj.bias, (layer.norm1.weight, layer.norm1.bias),
                                        layer.norm1.eps, res=table_bf16, res_kind=2, res_ids=cmp.ids_c, w1p=w1p,
                                        w2p=w2p, b2=layer.linear2.bias, ln2=(layer.norm2.weight, layer.norm2.bias), ln2_eps=layer.norm2.eps,
                                        E=E, pool32=True, m_dev=cmp.n_rows)     # fp32 [cap / 32, EP] block means
        pooled_c = ops.mean_pool(blocks[:, :E], M + 1, S // 32, n_seq_dev=cmp.n_compact)
        ops.gather_rows(cmp.seq_inv, pooled_c, pooled_out)
        return None
    w_o = ops.to_bf16(sa.out_proj.weight, rows_out=EP, cols_out=EP)
    x1 = ops.linear_bf16(attn, w_o, padv(sa.out_proj.bias), res=table_bf16, res_kind=2, res_ids=cmp.ids_c, res_pe=pe_p, res_period=S,
                         ln=(padv(layer.norm1.weight), padv(layer.norm1.bias)), ln_eps=layer.norm1.eps, ln_count=E, n_alg=E, k_alg=E,
                         m_dev=cmp.n_rows)
    if _ffn_fused_applicable(layer, E, EP):
        w1p, w2p = ops.ffn_pack_bf16(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight)
        blocks = ops.encoder_ffn_bf16(x1, w1p, w2p, layer.linear2.bias, (layer.norm2.weight, layer.norm2.bias), layer.norm2.eps, E,
                                      pool32=True, m_dev=cmp.n_rows)           # fp32 [cap / 32, EP] block means
    else:
        h = ops.linear_bf16(x1, ops.to_bf16(layer.linear1.weight, cols_out=EP), layer.linear1.bias, act='relu', k_alg=E, m_dev=cmp.n_rows)
        blocks = ops.linear_bf16(h, ops.to_bf16(layer.linear2.weight, rows_out=EP), padv(layer.linear2.bias), res=x1, res_kind=3,
                                 ln=(padv(layer.norm2.weight), padv(layer.norm2.bias)), ln_eps=layer.norm2.eps, ln_count=E, pool32=True,
                                 n_alg=E, m_dev=cmp.n_rows)
    pooled_c = ops.mean_pool(blocks[:, :E], M + 1, S // 32, n_seq_dev=cmp.n_compact)
    ops.gather_rows(cmp.seq_inv, pooled_c, pooled_out)
    return None


def compact_applicable_bf16(ids, transformer, nhead, E):
    M, S = ids.shape
    return (DEDUP and len(transformer.layers) == 1 and transformer.norm is None and E % nhead == 0 and E // nhead <= 32 and
            (E // nhead) % 2 == 0 and S in (32, 64, 128) and (M + 1) * S >= 4096 and ids.dtype == torch.int32 and ids.is_contiguous())


class CROWN(NewsEncoder):
    """newsEncoders.py:228-373: title/body transformer encoders, mean pooling, category-aware k-intent
    disentanglement, intent attention, title-body similarity, feature fusion.  -> [B, n, 900]."""

    def __init__(self, config):
        super().__init__(config)
        self.max_title_length = config.max_title_length
        self.max_body_length = config.max_abstract_length
        self.compute_dtype = getattr(config, 'compute_dtype', 'fp32')            # 'fp32' | 'bf16' (not a reference option)
        if self.compute_dtype not in ('fp32', 'bf16'):
            raise ValueError('compute_dtype must be fp32 or bf16')
        self.max_history_num = config.max_history_num
        self.category_embedding_dim = config.category_embedding_dim
        self.intent_embedding_dim = config.intent_embedding_dim
        self.category_embedding = nn.Embedding(config.category_num, config.category_embedding_dim)     # trainable again (:237)
        self.news_embedding_dim = config.intent_embedding_dim * 2 + config.category_embedding_dim + config.subCategory_embedding_dim
        self.head_num = config.head_num
        self.title_pos_encoder = PositionalEncoding(config.word_embedding_dim, config.dropout_rate, config.max_title_length)
        self.body_pos_encoder = PositionalEncoding(config.word_embedding_dim, config.dropout_rate, config.max_abstract_length)
        title_encoder_layers = TransformerEncoderLayer(config.word_embedding_dim, config.head_num, config.feedforward_dim,
                                                       config.dropout_rate, batch_first=True)
        self.title_transformer = TransformerEncoder(title_encoder_layers, config.num_layers)
        body_encoder_layers = TransformerEncoderLayer(config.word_embedding_dim, config.head_num, config.feedforward_dim,
                                                      config.dropout_rate, batch_first=True)
        self.body_transformer = TransformerEncoder(body_encoder_layers, config.num_layers)
        self.ISAB = ISAB(dim_in=config.word_embedding_dim, dim_out=config.word_embedding_dim, num_heads=config.isab_num_heads,
                         num_inds=config.isab_num_inds, ln=True)
        self.category_affine = nn.Linear(config.category_embedding_dim + config.subCategory_embedding_dim,
                                         config.category_embedding_dim)
        self.intent_num = config.intent_num
        self.alpha = config.alpha
        self.title_intent_attention = Attention(config.intent_embedding_dim, config.attention_dim)
        self.body_intent_attention = Attention(config.intent_embedding_dim, config.attention_dim)
        self.intent_layers = nn.ModuleList([nn.Linear(config.word_embedding_dim + config.category_embedding_dim,
                                                      config.intent_embedding_dim, bias=True) for _ in range(self.intent_num)])
        self.category_predictor = CategoryPredictor(config.intent_embedding_dim, config.category_num)

    def initialize(self):
        super().initialize()
        self.title_intent_attention.initialize()
        self.body_intent_attention.initialize()
        nn.init.xavier_uniform_(self.category_affine.weight)
        nn.init.zeros_(self.category_affine.bias)
        for intent_layer in self.intent_layers:
            nn.init.xavier_uniform_(intent_layer.weight)
            nn.init.zeros_(intent_layer.bias)
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)

    def encode_flat(self, title_text, title_mask, content_text, category, subCategory, out):
        """M news -> out [M, 900] (out may be a column slice of a wider buffer).  The token masks are computed and never
        used by the reference (:307-308); title_mask is accepted and ignored."""
        _no_train_dropout(self, self.dropout_rate)
        M, T = title_text.shape
        L = content_text.shape[1]
        if T != self.max_title_length or L != self.max_body_length:
            raise ValueError('token tensors must be [*, %d] / [*, %d]' % (self.max_title_length, self.max_body_length))
        E, Dc, D = self.word_embedding_dim, self.category_embedding_dim, self.intent_embedding_dim
        k = self.intent_num
        dev = title_text.device
        table = self.word_embedding.weight
        kin = E + Dc
        ldx = (kin + 3) // 4 * 4                                   # 352: keeps the rows 16-byte aligned
        # rows [0, M): [title_pooled | category_rep], rows [M, 2M): [body_pooled | category_rep]   (:343-344)
        xin = torch.empty((2 * M, ldx), dtype=torch.float32, device=dev)
        # The title and body encoders are independent chains of five GEMM / attention launches each (optionally on two
        # streams, see SERIAL_STREAMS).
        bf16 = self.compute_dtype == 'bf16'
        if bf16:                                                   # the word table in bf16, rows padded to 8 columns
            table_b = ops.to_bf16(table, cols_out=(E + 7) // 8 * 8)
        main = torch.cuda.current_stream()
        # branch 4: the category representation (:340-342), the raw category / subCategory rows of feature_fusion (:221-225) and
        # the stacked intent weights -- they depend on ids and weights only, and run beside the token encoders
        side4 = _side_stream(dev, 4)
        side4.wait_stream(main)
        with torch.cuda.stream(side4):
            sub_table = self.subCategory_embedding.weight
            ops.topic_rep(category, subCategory, self.category_embedding.weight, sub_table, self.category_affine.weight,
                          self.category_affine.bias, out=xin[:M, E:kin], emb_out=out[:, 2 * D:2 * D + Dc + sub_table.shape[1]])
            ops.topic_rep(category, subCategory, self.category_embedding.weight, sub_table, self.category_affine.weight,
                          self.category_affine.bias, out=xin[M:, E:kin])
            # the k intent weight matrices stacked row-wise, K = 350 carried as ldx = 352 zero-padded columns (16-byte rows: the
            # LDS-DMA GEMM kernels take it; the two pad columns of xin are zeroed to match)
            w_int = torch.nn.functional.pad(torch.cat([lin.weight for lin in self.intent_layers], dim=0), (0, ldx - kin))
            b_int = torch.cat([lin.bias for lin in self.intent_layers], dim=0)
            if ldx > kin:
                xin[:, kin:] = 0.0
        encoders = ((title_text, self.title_pos_encoder, self.title_transformer, T),
                    (content_text, self.body_pos_encoder, self.body_transformer, L))
        step_of = lambda S: min(max(1, MAX_TOKENS_PER_PASS // S),
                                # the compacted in_proj scatters rows with 32-bit byte offsets from the base of qkv ([rows, 3 * 320])
                                max(1, (0x7FFFFFF0 // (3 * self.head_num * 32 * (2 if bf16 else 4))) // S - 2) if DEDUP else M,
                                # the fused bf16 entry points address a pass's rows with 32-bit byte offsets ([rows, 304] bf16)
                                max(1, (0x7FFFFFF0 // (((E + 7) // 8 * 8) * 2)) // S - 2) if bf16 else M)
        one_pass = (not bf16 and all(M <= step_of(S) and compact_applicable(ids, table, tr, self.head_num)
                                     for ids, pos, tr, S in encoders))
        if one_pass:
            # both encoders on the compacted path in one pass each: the body's preparation (index lists, padded weights, padding
            # rows: a dozen short launches) runs on branch 3 under the title encoder's GEMMs
            (t_ids, t_pos, t_tr, _), (b_ids, b_pos, b_tr, _) = encoders
            side3 = _side_stream(dev, 3)
            side3.wait_stream(main)
            with torch.cuda.stream(side3):              # both encoders' preparation together: their small GEMMs share launches
                prep_t, prep_b = compact_prepare_many([(t_ids, t_pos.table(), t_tr), (b_ids, b_pos.table(), b_tr)], table, self.head_num)
            side1 = _side_stream(dev, 7)                   # branch 7: the (short) title chain beside the body chain
            side1.wait_stream(side3)
            with torch.cuda.stream(side1):
                compact_run(prep_t, table, t_pos.table(), t_tr, self.head_num, xin[:M, :E])                    # :311-317
            main.wait_stream(side3)
            compact_run(prep_b, table, b_pos.table(), b_tr, self.head_num, xin[M:, :E])                        # :312-321
            main.wait_stream(side1)
        else:
            side = _side_stream(dev, 7 if DEDUP else 1)   # compacted chunks: title beside body (branch 7); dense: branch 1 (off)
            side.wait_stream(main)
            for half, (ids, pos, tr, S) in enumerate(encoders):
                step = step_of(S)
                with torch.cuda.stream(side if half == 0 else main):
                    shared = {}                                # packed weights of this encoder, for the passes of this forward
                    for m0 in range(0, M, step):
                        m1 = min(M, m0 + step)
                        if bf16:
                            if compact_applicable_bf16(ids[m0:m1], tr, self.head_num, E):
                                encode_tokens_bf16_compact(ids[m0:m1], table_b, pos.table(), tr, self.head_num,
                                                           xin[half * M + m0:half * M + m1, :E], shared=shared)
                            else:
                                encode_tokens_bf16(ids[m0:m1], table_b, pos.table(), tr, self.head_num, xin[half * M + m0:half * M + m1, :E])
                            continue
                        if compact_applicable(ids[m0:m1], table, tr, self.head_num):
                            encode_tokens_compact(ids[m0:m1], table, pos.table(), tr, self.head_num,
                                                  pooled_out=xin[half * M + m0:half * M + m1, :E])
                            continue
                        encode_tokens(ids[m0:m1], table, pos.table(), tr, self.head_num,
                                      pooled_out=xin[half * M + m0:half * M + m1, :E])                                   # :311-321
            main.wait_stream(side)
        main.wait_stream(side4)
        # k intent layers (:284-295): [2M, 350] x [400, 350]^T each, ReLU fused, written side by side
        # (one GEMM against the k weight matrices stacked row-wise: the k layers share their input)
        intents = ops.linear(xin, w_int, b_int, act='relu')
        # intent attention (:355-356): tanh(affine1) on the GEMM (title on the main stream, body on branch 5), the rest in the fuse kernel
        A = self.title_intent_attention.affine1.out_features
        hidden = torch.empty((2 * M * k, A), dtype=torch.float32, device=dev)
        iv = intents.view(2 * M * k, D)
        # the two halves do not depend on each other: one grouped launch (they were two launches on two streams)
        ops.linear_group([dict(a=iv[half * M * k:(half + 1) * M * k], w=att.affine1.weight, bias=att.affine1.bias, act='tanh',
                               out=hidden[half * M * k:(half + 1) * M * k])
                          for half, att in enumerate((self.title_intent_attention, self.body_intent_attention))])
        ops.intent_fuse(iv, hidden, self.title_intent_attention.affine2.weight.view(-1),
                        self.body_intent_attention.affine2.weight.view(-1), out, M, k, D, A)          # :355-371
        return out


class MHSA(NewsEncoder):
    """newsEncoders.py:566-595: title-only multi-head self-attention + additive attention.  -> [B, n, 300]."""

    def _encode_compact(self, ids, mask, out):
        """The title encoder over the live sequences + ONE representative of the padding news' title (see
        ``encode_tokens_compact``; sequence level only: this encoder masks its padding tokens, so their rows are needed).  A
        sequence repeats the representative when its ids are all zero AND its mask is the padding news' mask (first position
        set, corpus.py:476-477); an all-zero sequence with any other mask counts as live."""
        n, T = ids.shape
        mha = self.multiheadAttention
        dev = ids.device
        if mask.dtype not in (torch.bool, torch.uint8):
            mask = mask.bool()
        mask = mask.contiguous()
        # all-zero ids under a mask that is not the padding news' mask: live (a sentinel in the first id); behind the compaction the
        # sentinel goes back to the padding word and the key mask is gathered into compact order -- two launches (they were ~18 torch ops)
        cmp = ops.compact_sequences(ops.mhsa_live_ids(ids, mask))
        mask_c = ops.mhsa_compact_mask(cmp, mask)
        ids_c = cmp.ids_c
        qkv = mha.project(table=self.word_embedding.weight, ids=ids_c, m_dev=cmp.n_rows)
        c = mha.attend(qkv, n + 1, T, mask_c, n_seq_dev=cmp.n_compact)
        hidden = ops.linear(c, self.attention.affine1.weight, self.attention.affine1.bias, act='tanh', m_dev=cmp.n_rows)
        pooled_c = ops.additive_pool(hidden, self.attention.affine2.weight.view(-1), c, n + 1, T, mask=mask_c, n_seq_dev=cmp.n_compact)
        ops.gather_rows(cmp.seq_inv, pooled_c, out)

    def _compact_applicable(self, ids, n, T):
        """The compacted title path rests on the device-count forms of the kernels, which are stricter than the dense ones:
        lime_linear_f32 with m_dev needs 16-byte friendly operands (K and N multiples of 4), lime_token_attention with a device count
        head_dim <= 32 and T <= 512.  Anything else takes the dense branch."""
        mha = self.multiheadAttention
        return (DEDUP and (n + 1) * T >= 4096 and ids.dtype == torch.int32 and ids.is_contiguous() and mha.d_model % 4 == 0 and
                (3 * mha.h * mha.d_k) % 4 == 0 and mha.d_k <= 32 and mha.d_v <= 32 and T <= 512)

    def __init__(self, config):
        super().__init__(config)
        self.max_sentence_length = config.max_title_length
        self.feature_dim = config.head_num * config.head_dim
        self.multiheadAttention = MultiHeadAttention(config.head_num, config.word_embedding_dim, config.max_title_length,
                                                     config.max_title_length, config.head_dim, config.head_dim)
        self.attention = Attention(config.head_num * config.head_dim, config.attention_dim)
        self.news_embedding_dim = config.head_num * config.head_dim + config.category_embedding_dim + config.subCategory_embedding_dim
        self.category_embedding = nn.Embedding(config.category_num, config.category_embedding_dim)

    def initialize(self):
        super().initialize()
        self.multiheadAttention.initialize()
        self.attention.initialize()
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)

    def encode_flat(self, title_text, title_mask, content_text, category, subCategory, out):
        _no_train_dropout(self, self.dropout_rate)
        M, T = title_text.shape
        F = self.feature_dim
        mask = title_mask.contiguous()
        mha = self.multiheadAttention
        step = max(1, MAX_TOKENS_PER_PASS // T)
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            ids = title_text[m0:m1]
            if self._compact_applicable(ids, m1 - m0, T):
                self._encode_compact(ids, mask[m0:m1], out[m0:m1, :F])
                continue
            qkv = mha.project(table=self.word_embedding.weight, ids=ids.reshape(-1))                    # :588 + layers.py:224-226
            c = mha.attend(qkv, m1 - m0, T, mask[m0:m1])                                                # layers.py:227-237
            hidden = ops.linear(c, self.attention.affine1.weight, self.attention.affine1.bias, act='tanh')
            ops.additive_pool(hidden, self.attention.affine2.weight.view(-1), c, m1 - m0, T, mask=mask[m0:m1],
                              out=out[m0:m1, :F])                                                       # :592
        ops.topic_rep(category, subCategory, self.category_embedding.weight, self.subCategory_embedding.weight,
                      emb_out=out[:, F:])                                                               # :594
        return out
