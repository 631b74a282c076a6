"""Drop-in ``Model`` for the LIME-{CROWN,MHSA}-CROWN scoring path (reference model.py:11-187)."""
import torch
import torch.nn as nn

from . import newsEncoders, ops, training, userEncoders
from .util import RemainingLifetimeWeighting

# positions (in the 26-tensor signature, model.py:151-154) of the inputs the scoring path reads; the others
# (user_ID, *_entity, content masks, user_history_graph / category_mask / category_indices) are ignored by the
# reference too (last bullet)
_USED = (1, 2, 3, 4, 6, 9, 10, 11, 15, 16, 17, 18, 20, 23, 24, 25)
# (candidate tensor, history tensor) pairs the news encoder concatenates: title_text, title_mask, content_text, category,
# subCategory, freshness, user_topic_lifetime
_PAIRS = ((17, 3), (18, 4), (20, 6), (15, 1), (16, 2), (23, 9), (24, 10))


class Model(nn.Module):
    """Same constructor, attributes (``model_name``, ``config``, ``news_encoder``, ``user_encoder``,
    ``news_embedding_dim``), ``initialize()`` and 26-tensor ``forward`` as the reference's Model
    (model.py:12-187); ``state_dict()`` has the reference's key set.  ``forward`` returns logits [B, N].

    Scoring (eval mode, or any call under ``torch.no_grad()``): the forward pass runs entirely in hand-written HIP
    kernels and records no autograd graph.  In training mode with grad enabled (``model.train(); model(...)``, what
    trainer.py:131 does) ``forward`` takes the differentiable path of ``lime_cikm25_amd.training``.  Candidates and history
    are encoded in one pass over the news-encoder kernels (the reference encodes them in two calls, model.py:171 and userEncoders.py:110).

    The ~75 kernel launches of a forward are captured once per input signature into a HIP graph and replayed
    (``use_graph``, on by default): the forward is launch-bound from Python otherwise (2.5 ms of gaps on 6 ms of
    kernels at batch 32).  Inputs are copied into the graph's static buffers on every call; parameters are read in
    place, so in-place updates (``load_state_dict``, optimizer steps) need no re-capture, while ``.to()/.cuda()``
    drop the captured graphs.
    """

    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.news_encoder != 'LIME':
            raise NotImplementedError('news_encoder %r: the MI355X path covers LIME (config.py:25)' % config.news_encoder)
        if config.content_encoder == 'CROWN':
            base_encoder = newsEncoders.CROWN(config)
        elif config.content_encoder == 'MHSA':
            base_encoder = newsEncoders.MHSA(config)
        else:
            raise NotImplementedError('content_encoder %r is a baseline outside the scoring path' % config.content_encoder)
        self.news_encoder = newsEncoders.LIME(config=config, base_news_encoder=base_encoder)
        if config.user_encoder != 'CROWN':
            raise NotImplementedError('user_encoder %r is a baseline outside the scoring path' % config.user_encoder)
        self.user_encoder = userEncoders.CROWN(self.news_encoder, config)
        self.model_name = f"{config.news_encoder}-{config.content_encoder}-{config.user_encoder}"
        self.news_embedding_dim = self.news_encoder.news_embedding_dim
        self.dropout = nn.Dropout(p=config.dropout_rate)
        self.use_user_embedding = False
        self.click_predictor = config.click_predictor
        if self.click_predictor != 'dot_product':
            raise NotImplementedError('click_predictor %r: LIME uses dot_product (config.py:109)' % self.click_predictor)
        self.remaining_lifetime_weighting = RemainingLifetimeWeighting(config)
        self.use_graph = True
        self._graphs = {}

    def _apply(self, fn, *args, **kwargs):
        self._graphs = {}                      # parameter storage moves: captured pointers are stale
        return super()._apply(fn, *args, **kwargs)

    def initialize(self):
        self.news_encoder.initialize()
        self.user_encoder.initialize()
        self.remaining_lifetime_weighting.initialize()

    def forward(self, user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity,
                user_content_text, user_content_mask, user_content_entity, user_freshness, user_user_topic_lifetime,
                user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
                news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask,
                news_content_entity, news_freshness, news_user_topic_lifetime, remaining_lifetime):
        args = (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity,
                user_content_text, user_content_mask, user_content_entity, user_freshness, user_user_topic_lifetime,
                user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
                news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask,
                news_content_entity, news_freshness, news_user_topic_lifetime, remaining_lifetime)
        if self.training and (torch.is_grad_enabled() or self.news_encoder.training or self.user_encoder.training):
            # trainer.py:131-145: model.train(); logits = model(...); loss.backward() -- the differentiable path.  It is also the
            # path with the training-mode dropouts (model.train() under no_grad: the layer's p = 0.2 dropout of layers.py:74 is
            # active whatever config.dropout_rate says); `model.eval(); model.training = True` keeps the [B, K] shape on the
            # fused scoring kernels (children in eval mode)
            return training.forward_train(self, user_category, user_subCategory, user_title_text, user_title_mask,
                                          user_content_text, user_freshness, user_user_topic_lifetime, user_history_mask,
                                          news_category, news_subCategory, news_title_text, news_title_mask, news_content_text,
                                          news_freshness, news_user_topic_lifetime, remaining_lifetime)
        if (self.use_graph and ops.PROFILE is None and user_category.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            return self._forward_graphed(args)
        return self._forward_impl(*args)

    def _forward_graphed(self, args):
        used = [args[i] for i in _USED]
        key = (self.training,) + tuple((tuple(t.shape), t.dtype) for t in used)
        entry = self._graphs.get(key)
        if entry is None:
            static = list(args)
            for i, t in zip(_USED, self._packed_like(args)):
                static[i] = t
            ops.multi_copy([(static[i], args[i]) for i in _USED])
            with torch.no_grad():
                self._forward_impl(*static)                 # eager warm-up: lazy one-time set-up stays out of the capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward_impl(*static)
            entry = (graph, [static[i] for i in _USED], out)
            self._graphs[key] = entry
        graph, static_used, out = entry
        keep = ops.multi_copy(list(zip(static_used, used)))          # one launch for all the inputs
        graph.replay()
        del keep
        return out.clone()

    @staticmethod
    def _packed_like(args):
        """Static input buffers for the graph: ONE allocation, with every candidate tensor directly in front of its
        history counterpart (news_title_text | user_title_text, ...) so that the encoder's `cat` of the two groups is a
        view of the buffer instead of a copy kernel (newsEncoders.LIME.encode_many)."""
        order = list(_USED)
        for n_i, u_i in _PAIRS:                                      # news tensor, then the user tensor right behind it
            order.remove(u_i)
            order.insert(order.index(n_i) + 1, u_i)
        offs, total = {}, 0
        for i in order:
            t = args[i]
            adjacent = any(i == u and args[n].dtype == t.dtype for n, u in _PAIRS)
            if not adjacent:
                total = (total + 255) // 256 * 256                   # 256-byte aligned unless glued to its partner
            offs[i] = total
            total += t.numel() * t.element_size()
        buf = torch.empty(total + 256, dtype=torch.uint8, device=args[_USED[0]].device)
        base = (-buf.data_ptr()) % 256
        out = []
        for i in _USED:
            t = args[i]
            nbytes = t.numel() * t.element_size()
            out.append(buf[base + offs[i]:base + offs[i] + nbytes].view(t.dtype).view(t.shape))
        return out

    @torch.no_grad()
    def score_impressions(self, user_category, user_subCategory, user_title_text, user_title_mask, user_content_text,
                          user_freshness, user_user_topic_lifetime, user_history_mask, news_category, news_subCategory,
                          news_title_text, news_title_mask, news_content_text, news_freshness, news_user_topic_lifetime,
                          remaining_lifetime, n_src=None, rows_per_pass=16384):
        """Scoring-only layout of BASELINE config 5: B impressions with K candidates each -> logits [B, K] with the
        reference's EVAL semantics (util.py:86-111: every (impression, candidate) pair is its own row with N = 1, Q16),
        but every history is encoded ONCE instead of once per candidate.

        user_* [B, H, ...], news_* [B, K, ...] (news_user_topic_lifetime may be [B] or [B, K]), remaining_lifetime [B, K].
        ``n_src``: the GraphSAGE closed form averages over the first n_src node slots, and the reference takes n_src =
        rows of the forward, i.e. its eval batch size (Q7); default: B * K capped at H + config.batch_size, which is what
        one reference forward over all pairs would use.  The result equals ``forward`` in eval mode on the B * K expanded
        rows (tested), the history encoder just runs B * H instead of B * K * H times.
        """
        B, K = news_category.shape
        H = user_category.shape[1]
        ne, ue = self.news_encoder, self.user_encoder
        if news_user_topic_lifetime.dim() == 1:
            news_user_topic_lifetime = news_user_topic_lifetime.unsqueeze(1).expand(B, K)
        if news_freshness.dim() == 1:
            news_freshness = news_freshness.unsqueeze(1).expand(B, K)
        cand, hist = ne.encode_many([
            (news_title_text, news_title_mask, news_content_text, news_category, news_subCategory, news_freshness.contiguous(),
             news_user_topic_lifetime.contiguous()),
            (user_title_text, user_title_mask, user_content_text, user_category, user_subCategory, user_freshness,
             user_user_topic_lifetime)])                                              # [B, K, D], [B, H, D]
        rows = B * K
        if n_src is None:
            n_src = min(rows, H + ue.user_node_embedding.shape[0])
        out = torch.empty((B, K), dtype=torch.float32, device=cand.device)
        per = max(1, rows_per_pass // K)                                              # impressions per pass ...
        n_pass = (B + per - 1) // per
        per = (B + n_pass - 1) // n_pass          # ... evened out: the passes then take the same kernels (a short last pass would
                                                  # fall under the row counts from which the big-M GEMM kernels take over)
        rl = remaining_lifetime.float()
        gate_y = ue.gate_projection(hist)                                             # once for every pass's histories (one GEMM shape)
        for b0 in range(0, B, per):
            b1 = min(B, b0 + per)
            n = (b1 - b0) * K
            # the history side (embeddings, topic ids, mask) is shared by the K candidate rows of an impression: hist_div
            _, logits = ue.match(hist[b0:b1], news_category[b0:b1].reshape(n, 1), news_subCategory[b0:b1].reshape(n, 1),
                                 user_category[b0:b1], user_subCategory[b0:b1], user_history_mask[b0:b1],
                                 cand[b0:b1].reshape(n, 1, -1), remaining_lifetime=rl[b0:b1].reshape(n, 1),
                                 weighting=self.remaining_lifetime_weighting, n_src=n_src, hist_div=K,
                                 gate_y=None if gate_y is None else gate_y[b0 * H:b1 * H])
            out[b0:b1] = logits.view(b1 - b0, K)
        return out

    @torch.no_grad()
    def build_news_cache(self, device_corpus, rows_per_pass=8192):
        """Content cache for ``score_behaviors``: every news of a ``DeviceCorpus`` through the token encoders ONCE."""
        c = device_corpus
        return self.news_encoder.build_content_cache(c.news_title_text, c.news_title_mask, c.news_abstract_text, c.news_category,
                                                     c.news_subCategory, rows_per_pass=rows_per_pass)

    @torch.no_grad()
    def score_behaviors(self, behaviors, rows, news_cache, n_src=None):
        """Scores of the (impression, candidate) rows `rows` of a dev / test ``DeviceBehaviors`` -- the function of
        util.compute_scores' forward (util.py:86-111, eval mode, N = 1 per row) -- from the news cache: no token encoder
        runs, the history and the candidate of a row are looked up by news index and only their freshness half, the user
        encoder and the match are computed.  ``n_src`` as in score_impressions (default: the number of rows, capped)."""
        b = behaviors
        dev = news_cache.device
        rows = torch.as_tensor(rows, device=dev).long().reshape(-1)
        R, H = rows.numel(), b.hist_index.shape[1]
        ne, ue, c = self.news_encoder, self.user_encoder, b.corpus
        hist_idx, cand_idx = b.hist_index[rows], b.cand_index[rows]                       # [R, H], [R, 1]
        hist = ne.encode_cached(news_cache, hist_idx, b.user_freshness[rows], b.user_lifetime[rows]).view(R, H, -1)
        cand = ne.encode_cached(news_cache, cand_idx, b.cand_freshness[rows], b.cand_lifetime[rows]).view(R, 1, -1)
        flat_h, flat_c = hist_idx.reshape(-1).long(), cand_idx.reshape(-1).long()
        # the remaining lifetime per config.lifetime_type, as util.py:98-106 derives it from the batch
        lt = getattr(self.config, 'lifetime_type', 'user_topic')
        if lt == 'fixed':
            remaining = self.config.fixed_lifetime - b.cand_freshness[rows]
        elif lt == 'topic_wise':
            cmap = torch.as_tensor(self.config.category_lifetime_map, dtype=torch.float32, device=dev)
            remaining = cmap[c.news_category[cand_idx.reshape(-1).long()].long()].view_as(b.cand_freshness[rows]) - b.cand_freshness[rows]
        elif lt == 'user_topic':
            remaining = (b.cand_lifetime[rows] - b.cand_freshness[rows])
        else:
            raise ValueError('Invalid lifetime_type')
        if n_src is None:
            n_src = min(R, H + ue.user_node_embedding.shape[0])
        _, logits = ue.match(hist, c.news_category[flat_c].view(R, 1), c.news_subCategory[flat_c].view(R, 1),
                             c.news_category[flat_h].view(R, H), c.news_subCategory[flat_h].view(R, H), b.hist_mask[rows],
                             cand, remaining_lifetime=remaining.view(R, 1), weighting=self.remaining_lifetime_weighting, n_src=n_src)
        return logits.view(R)

    def _forward_impl(self, user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity,
                      user_content_text, user_content_mask, user_content_entity, user_freshness, user_user_topic_lifetime,
                      user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices,
                      news_category, news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text,
                      news_content_mask, news_content_entity, news_freshness, news_user_topic_lifetime, remaining_lifetime):
        if not self.training:                                                    # model.py:158-169
            news_category = news_category.unsqueeze(1)
            news_subCategory = news_subCategory.unsqueeze(1)
            news_title_text = news_title_text.unsqueeze(1)
            news_title_mask = news_title_mask.unsqueeze(1)
            news_content_text = news_content_text.unsqueeze(1)
            news_freshness = news_freshness.unsqueeze(1)
            news_user_topic_lifetime = news_user_topic_lifetime.unsqueeze(1)
            remaining_lifetime = remaining_lifetime.unsqueeze(1)
        with torch.no_grad():
            # candidate-aware attention weights depend on topic ids and the history mask only: side stream, joined below
            main = torch.cuda.current_stream()
            side2 = newsEncoders._side_stream(user_category.device, 2)
            side2.wait_stream(main)
            with torch.cuda.stream(side2):
                agg = self.user_encoder.attention_weights(news_category, news_subCategory, user_category, user_subCategory,
                                                          user_history_mask)
            news_representation, history_embedding = self.news_encoder.encode_many([
                (news_title_text, news_title_mask, news_content_text, news_category, news_subCategory, news_freshness,
                 news_user_topic_lifetime),                                      # model.py:171-173
                (user_title_text, user_title_mask, user_content_text, user_category, user_subCategory, user_freshness,
                 user_user_topic_lifetime)])                                     # userEncoders.py:110-112
            main.wait_stream(side2)
            _, logits = self.user_encoder.match(history_embedding, news_category, news_subCategory, user_category,
                                                user_subCategory, user_history_mask, news_representation,
                                                remaining_lifetime=remaining_lifetime.float(),
                                                weighting=self.remaining_lifetime_weighting, agg=agg)   # model.py:174-181
        return logits
