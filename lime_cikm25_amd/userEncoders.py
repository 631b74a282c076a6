"""Drop-in CROWN user encoder of the scoring path (reference userEncoders.py:16-175)."""
import math

import torch
import torch.nn as nn

from . import ops
from .layers import CandidateAware_ClickedNewsAttention


class SAGEConv(nn.Module):
    """Parameter holder with PyG's SAGEConv names: ``lin_l`` (bias) on the aggregate, ``lin_r`` (no bias) on the root."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)


class GraphSAGE(nn.Module):
    """``torch_geometric.nn.GraphSAGE(in, hidden, num_layers=1, out_channels=...)`` as the reference constructs it
    (userEncoders.py:54-58): one SAGEConv, nothing after it.  PyG is an absent, un-versioned dependency; the
    aggregation is restated from its documented semantics (parity unpinned at this boundary)."""

    def __init__(self, in_channels, hidden_channels, num_layers, out_channels=None, dropout=0.0):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError('the reference uses num_layers = 1 (userEncoders.py:56)')
        self.convs = nn.ModuleList([SAGEConv(in_channels, out_channels or hidden_channels)])

    def forward_closed_form(self, hist, user_nodes, n_src):
        """hist [B, H, D] (node slots 0..H-1 of every row), user_nodes [n_user, D] (slots H..), n_src = rows per forward.

        create_bipartite_graph (userEncoders.py:91-98) yields edges (u -> i) for u < n_src, i < H applied along the node
        axis of each row, so  out[b, i] = lin_l(mean_{u < n_src} X[b, u]) + lin_r(X[b, i])  for the H history slots that
        survive the slice at :157.
        """
        B, H, D = hist.shape
        conv = self.convs[0]
        flat = hist.reshape(B * H, D)
        m = ops.sage_mean(flat, user_nodes.contiguous(), B, H, n_src, D)
        l = ops.linear(m, conv.lin_l.weight, conv.lin_l.bias)
        g = ops.linear(flat, conv.lin_r.weight, None, res=l, res_div=H)
        return g.view(B, H, D)


class _NoParams(nn.Module):
    """LightGCN / LGConv are constructed and never called by the reference (userEncoders.py:59-62); they hold no
    checkpoint entries."""


class UserEncoder(nn.Module):
    def __init__(self, news_encoder, config):
        super().__init__()
        self.news_embedding_dim = news_encoder.news_embedding_dim
        self.news_encoder = news_encoder
        self.device = torch.device('cuda')
        self.auxiliary_loss = None
        self.word_embedding_dim = config.word_embedding_dim
        self.batch_size = config.batch_size


class CROWN(UserEncoder):
    """userEncoders.py:49-175: history encoding -> candidate-aware clicked-news attention -> GraphSAGE step ->
    history-vs-candidate attention.  ``forward`` keeps the reference's 18-argument signature."""

    def __init__(self, news_encoder, config):
        super().__init__(news_encoder, config)
        self.attention_dim = config.attention_dim
        self.graph_sage = GraphSAGE(in_channels=self.news_embedding_dim, hidden_channels=self.news_embedding_dim, num_layers=1,
                                    out_channels=self.news_embedding_dim, dropout=config.dropout_rate)
        self.lightgcn = _NoParams()
        self.lgconv = _NoParams()
        self.user_node_embedding = nn.Parameter(torch.zeros([config.batch_size, self.news_embedding_dim]))
        self.K = nn.Linear(self.news_embedding_dim, self.attention_dim, bias=False)
        self.Q = nn.Linear(self.news_embedding_dim, self.attention_dim, bias=True)
        self.max_history_num = config.max_history_num
        self.attention_scalar = math.sqrt(float(self.attention_dim))
        self.affine = nn.Linear(self.news_embedding_dim, self.news_embedding_dim, bias=True)      # unused upstream (:171)
        self.dropout_rate = config.dropout_rate
        self.dropout = nn.Dropout(p=config.dropout_rate, inplace=True)
        self.dropout_ = nn.Dropout(p=config.dropout_rate, inplace=False)
        self.use_candidate_aware_attn = config.use_candidate_ware_clicked_news_attention
        if self.use_candidate_aware_attn:
            self.candidate_aware_attn = CandidateAware_ClickedNewsAttention(config, news_encoder)

    def initialize(self):
        nn.init.zeros_(self.user_node_embedding)
        nn.init.xavier_uniform_(self.K.weight)
        nn.init.xavier_uniform_(self.Q.weight)
        nn.init.zeros_(self.Q.bias)
        nn.init.xavier_uniform_(self.affine.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.affine.bias)
        if self.use_candidate_aware_attn:
            self.candidate_aware_attn.initialize()

    def _topic(self, category, subCategory):
        """userEncoders.py:103-105 / :115-117: LIME's own frozen tables + category_affine."""
        ne = self.news_encoder
        shape = category.shape
        cat = category.reshape(-1)
        sub = subCategory.reshape(-1)
        cat = cat if cat.dtype == torch.int32 else cat.to(torch.int32)
        sub = sub if sub.dtype == torch.int32 else sub.to(torch.int32)
        rep = ops.topic_rep(cat.contiguous(), sub.contiguous(), ne.category_embedding.weight, ne.subCategory_embedding.weight,
                            ne.category_affine.weight, ne.category_affine.bias)
        return rep.view(*shape, -1)

    def attention_weights(self, category, subCategory, user_category, user_subCategory, user_history_mask, hist_div=1):
        """The candidate-aware attention weights agg [B, H] (layers.py:66-81).  They depend on the topic ids and the
        history mask only -- not on any news embedding -- so the model computes them on a side stream while the token
        encoders run."""
        if not self.use_candidate_aware_attn:
            return None
        if self.training and self.candidate_aware_attn.dropout.p > 0:
            raise NotImplementedError('attention_weights() is the scoring kernel: in training mode the layer\'s p = 0.2 dropout '
                                      '(layers.py:36,74) runs on the differentiable path (user_encoder(...) / Model.forward)')
        cand_topic = self._topic(category, subCategory)
        hist_topic = self._topic(user_category, user_subCategory)
        return self.candidate_aware_attn.attention_weights(hist_topic, cand_topic, user_history_mask, hist_div=hist_div)

    def match(self, history_embedding, category, subCategory, user_category, user_subCategory, user_history_mask,
              candidate_news_representation, remaining_lifetime=None, weighting=None, agg=None, n_src=None, hist_div=1,
              gate_y=None):
        """Everything after the history has been encoded (userEncoders.py:103-105, :114-169).

        Returns (user_representation [B, N, D], logits [B, N] or None).  With ``weighting`` (the model's
        RemainingLifetimeWeighting) the dot-product match and the lifetime weight are fused into the last kernel.
        ``agg``: precomputed ``attention_weights(...)``.  ``n_src``: how many node slots the GraphSAGE mean runs over
        (Q7: the reference uses the number of rows of the forward; default B).
        ``hist_div`` > 1 (Model.score_impressions): ``history_embedding`` is [B / hist_div, H, D] -- ONE copy of a history for the
        hist_div consecutive rows (candidates) that share it, and so are the history's topic ids and mask ([B / hist_div, H]); the
        candidates' own ids and ``agg`` (if given) are per row [B, ...].
        ``gate_y``: ``gate_projection(history_embedding)`` computed by the caller (one GEMM over all passes' histories).
        """
        if self.training and (self.dropout_rate > 0 or (self.use_candidate_aware_attn and self.candidate_aware_attn.dropout.p > 0)):
            raise NotImplementedError('match() is the fused scoring kernel chain: with training-mode dropouts active (userEncoders.py:121, '
                                      'layers.py:74) call user_encoder(...) or Model.forward, which take the differentiable path')
        Bh, H, D = history_embedding.shape
        B = Bh * hist_div
        N = candidate_news_representation.shape[1]
        cand = candidate_news_representation.contiguous()
        n_src = B if n_src is None else n_src
        caa = self.candidate_aware_attn if self.use_candidate_aware_attn else None
        fused = caa is not None and caa.use_residual_connection and D <= 512
        main = torch.cuda.current_stream()
        if fused and gate_y is None:
            # Q(candidates) (:162) and gate_proj(history) (layers.py:87) do not depend on each other: one grouped launch
            qp, gate_y = ops.linear_group([dict(a=cand.view(B * N, D), w=self.Q.weight, bias=self.Q.bias),
                                           dict(a=history_embedding.reshape(Bh * H, D), w=caa.gate_proj.weight, bias=None)])
            side6 = None
        else:
            # the candidate side of the match (:162) needs the candidates only: branch 6, beside the history chain
            from .newsEncoders import _side_stream
            side6 = _side_stream(cand.device, 6)
            side6.wait_stream(main)
            with torch.cuda.stream(side6):
                qp = ops.linear(cand.view(B * N, D), self.Q.weight, self.Q.bias)                             # :162
        if caa is not None and agg is None:
            # (hist_div > 1: user_category / user_subCategory / user_history_mask are [B / hist_div, H] -- one history per hist_div rows)
            agg = self.attention_weights(category, subCategory, user_category, user_subCategory, user_history_mask, hist_div=hist_div)
        conv = self.graph_sage.convs[0]
        if fused:
            # gate_proj sees the history alone (the row scale commutes: layers.py:85-87), so it runs once per HISTORY; the gated
            # residual + LayerNorm of each row's H history rows and the SAGEConv mean over them are one launch, which reads a
            # shared history through row / hist_div (no per-candidate copies) and takes the user-node part of the mean -- the same
            # vector for every row -- as one precomputed sum                                                  :121,:151-157
            hist = history_embedding.reshape(Bh * H, D)
            y = gate_y
            node_const = self.user_node_embedding[:n_src - H].sum(dim=0) if n_src > H else None
            refined, m = ops.gate_ln_sage(y, hist, agg.reshape(-1), caa.gate_proj.bias, caa.layernorm.weight, caa.layernorm.bias,
                                          caa.layernorm.eps, B, H, D, hist_div, n_src, node_const)
            l = ops.linear(m, conv.lin_l.weight, conv.lin_l.bias)
            g = ops.linear(refined, conv.lin_r.weight, None, res=l, res_div=H).view(B, H, D)
        else:
            if hist_div > 1:
                history_embedding = history_embedding.repeat_interleave(hist_div, dim=0)
            if caa is not None:
                history_embedding = caa.refine(history_embedding, agg)
            g = self.graph_sage.forward_closed_form(history_embedding, self.user_node_embedding, n_src=n_src)   # :121,:151-157
        kp = ops.linear(g.view(B * H, D), self.K.weight, None)                                               # :161
        if side6 is not None:
            main.wait_stream(side6)
        w = weighting
        user, logits = ops.interest_match(
            kp, qp, g.reshape(-1), cand.reshape(-1), remaining_lifetime, B, N, H, self.attention_dim, D,
            1.0 / self.attention_scalar,
            w.alpha if w is not None else 0.0, w.beta if w is not None else 0.0,
            bool(w.use_remaining_lifetime_weighting) if w is not None else False,
            bool(w.use_expired_penalty) if w is not None else False,
            want_logits=w is not None, want_user=True)
        return user, logits

    def gate_projection(self, history_embedding):
        """W_g x of the candidate-aware attention's gate (layers.py:87) for histories [*, H, D] -> [* H, D], or None when ``match``
        does not take the fused path."""
        caa = self.candidate_aware_attn if self.use_candidate_aware_attn else None
        D = history_embedding.shape[-1]
        if caa is None or not caa.use_residual_connection or D > 512:
            return None
        return ops.linear(history_embedding.reshape(-1, D), caa.gate_proj.weight, None)

    def forward(self, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
                user_content_entity, category, subCategory, user_category, user_subCategory, user_history_mask,
                user_history_graph, user_history_category_mask, user_history_category_indices, user_embedding,
                candidate_news_representation, user_freshness, user_user_topic_lifetime):
        history_embedding = self.news_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text,
                                              user_content_mask, user_content_entity, user_category, user_subCategory,
                                              user_embedding, user_freshness, user_user_topic_lifetime)        # :110-112
        from . import training
        p_any = max(self.dropout_rate, self.candidate_aware_attn.dropout.p if self.use_candidate_aware_attn else 0.0)
        if training.wants_train_path(self, p_any):
            # training mode (trainer.py:87): the differentiable path; the candidate representation may carry a graph
            i32 = lambda t: (t if t.dtype == torch.int32 else t.to(torch.int32)).contiguous()
            return training.user_representation(self, history_embedding, candidate_news_representation.float(), i32(category),
                                                i32(subCategory), i32(user_category), i32(user_subCategory),
                                                user_history_mask.contiguous())
        user, _ = self.match(history_embedding, category, subCategory, user_category, user_subCategory, user_history_mask,
                             candidate_news_representation)
        return user
