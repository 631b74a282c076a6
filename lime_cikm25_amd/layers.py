"""Drop-ins for the three attention primitives of the scoring path (reference layers.py:15-93,
192-238, 269-300).  The modules own parameters under the reference's names (so reference
checkpoints load); ``forward`` runs the hand-written HIP kernels through lime_cikm25_amd.ops.
"""
import math

import torch
import torch.nn as nn

from . import ops


class CandidateAware_ClickedNewsAttention(nn.Module):
    """layers.py:15-93.  Q/K topic projections on the fp32 MFMA GEMM, per-head softmax + query-weighted
    aggregation in one workgroup per impression row, gate_proj on the GEMM, gated residual + LayerNorm in one
    row kernel.  ``value_proj`` is kept for the state_dict only: its branch is dead in the
    reference (layers.py:68,76-77)."""

    def __init__(self, config, news_encoder):
        super().__init__()
        self.topic_embedding_dim = config.category_embedding_dim
        self.use_residual_connection = config.use_residual_connection
        self.news_embedding_dim = news_encoder.news_embedding_dim
        self.num_heads = 10
        self.head_dim = self.news_embedding_dim // self.num_heads
        assert self.news_embedding_dim % self.num_heads == 0, 'embedding_dim must be divisible by num_heads'
        self.query_proj = nn.Linear(self.topic_embedding_dim, self.news_embedding_dim)
        self.key_proj = nn.Linear(self.topic_embedding_dim, self.news_embedding_dim)
        self.value_proj = nn.Linear(self.news_embedding_dim, self.news_embedding_dim)
        self.scale = self.news_embedding_dim ** 0.5
        self.dropout = nn.Dropout(p=0.2)
        self.gate_proj = nn.Linear(self.news_embedding_dim, self.news_embedding_dim)
        self.layernorm = nn.LayerNorm(self.news_embedding_dim)

    def initialize(self):
        nn.init.xavier_uniform_(self.query_proj.weight)
        nn.init.xavier_uniform_(self.key_proj.weight)
        nn.init.zeros_(self.query_proj.bias)
        nn.init.zeros_(self.key_proj.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.gate_proj.weight)
        nn.init.zeros_(self.gate_proj.bias)

    def attention_weights(self, clicked_news_topic_embeddings, candidate_topic_embeddings, mask=None, hist_div=1):
        """agg [B, H] of layers.py:66-81: topic projections, per-head masked softmax, query-norm weighting, outer softmax.
        ``hist_div`` > 1: the history side ([B / hist_div, H, .], mask [B / hist_div, H]) is shared by hist_div consecutive candidate rows."""
        Bh, H, _ = clicked_news_topic_embeddings.shape
        B = Bh * hist_div
        N = candidate_topic_embeddings.shape[1]
        D = self.news_embedding_dim
        if mask is None:
            mask = torch.ones(Bh, H, dtype=torch.bool, device=clicked_news_topic_embeddings.device)
        qp = ops.linear(candidate_topic_embeddings.reshape(B * N, -1), self.query_proj.weight, self.query_proj.bias)
        kp = ops.linear(clicked_news_topic_embeddings.reshape(Bh * H, -1), self.key_proj.weight, self.key_proj.bias)
        return ops.cand_attn_weights(qp, kp, mask, B, N, H, D, self.num_heads, hist_div=hist_div)

    def refine(self, clicked_news_embeddings, agg):
        """layers.py:83-91: weighted history, gated residual, LayerNorm."""
        B, H, D = clicked_news_embeddings.shape
        hist = clicked_news_embeddings.reshape(B * H, D)
        if self.use_residual_connection:
            y = ops.linear(hist, self.gate_proj.weight, None)                       # W_g x; the row scale commutes
            out = ops.gate_ln(y, hist, agg.view(-1), self.gate_proj.bias, self.layernorm.weight, self.layernorm.bias,
                              self.layernorm.eps)
        else:
            out = ops.row_scale(hist, agg.view(-1))
        return out.view(B, H, D)

    def forward(self, clicked_news_embeddings, clicked_news_topic_embeddings, candidate_topic_embeddings, mask=None):
        """-> (refined history [B, H, D], attn_weights_agg [B, H]).  In training mode (autograd recording, or the layer's own
        p = 0.2 dropout active, layers.py:36,74) the call takes the differentiable kernels of ``training.candidate_aware``."""
        from . import training
        if training.wants_train_path(self, self.dropout.p):
            return training.candidate_aware(self, clicked_news_embeddings.float(), clicked_news_topic_embeddings.float(),
                                            candidate_topic_embeddings.float(), mask)
        agg = self.attention_weights(clicked_news_topic_embeddings, candidate_topic_embeddings, mask)
        return self.refine(clicked_news_embeddings, agg), agg


class MultiHeadAttention(nn.Module):
    """layers.py:192-238 (self-attention use, Q = K = V): three projections + the masked token-attention kernel."""

    def __init__(self, h, d_model, len_q, len_k, d_k, d_v):
        super().__init__()
        self.h, self.d_model, self.len_q, self.len_k, self.d_k, self.d_v = h, d_model, len_q, len_k, d_k, d_v
        self.out_dim = self.h * self.d_v
        self.attention_scalar = math.sqrt(float(self.d_k))
        self.W_Q = nn.Linear(d_model, self.h * self.d_k, bias=True)
        self.W_K = nn.Linear(d_model, self.h * self.d_k, bias=True)
        self.W_V = nn.Linear(d_model, self.h * self.d_v, bias=True)

    def initialize(self):
        for lin in (self.W_Q, self.W_K, self.W_V):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def project(self, x2d=None, table=None, ids=None, m_dev=None):
        """[tokens, 3*h*d_k] packed q|k|v; the operand is either a dense [tokens, d_model] matrix or a gather.  m_dev: optional
        device-side row count (a compacted batch)."""
        assert self.d_k == self.d_v
        hd = self.h * self.d_k
        tokens = x2d.shape[0] if ids is None else ids.numel()
        src = x2d if ids is None else table
        # ONE GEMM against W_Q / W_K / W_V stacked row-wise (they share the input: one gather of the word rows instead of three,
        # and N = 3 h d_k = 600 fills two 320-column tiles where 200 wasted a fifth of a 256-column one)
        w = torch.cat([self.W_Q.weight, self.W_K.weight, self.W_V.weight], dim=0)
        b = torch.cat([self.W_Q.bias, self.W_K.bias, self.W_V.bias], dim=0)
        qkv = ops.linear(src, w, b, a_ids=ids, m_dev=m_dev)
        return qkv

    def attend(self, qkv, n_seq, S, mask, n_seq_dev=None):
        hd = self.h * self.d_k
        return ops.token_attention(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], n_seq, S, self.h, self.d_k,
                                   1.0 / self.attention_scalar, key_mask=mask, n_seq_dev=n_seq_dev)

    def forward(self, Q, K, V, mask=None):
        if not (Q is K and K is V):
            raise NotImplementedError('only the self-attention use (newsEncoders.py:590) is on the scoring path')
        n_seq, S, _ = Q.shape
        qkv = self.project(Q.reshape(n_seq * S, -1))
        return self.attend(qkv, n_seq, S, mask).view(n_seq, S, self.out_dim)


class Attention(nn.Module):
    """layers.py:269-300: additive attention pooling.  affine1 + tanh on the GEMM, the score / masked softmax /
    weighted sum in one workgroup per sequence."""

    def __init__(self, feature_dim, attention_dim):
        super().__init__()
        self.affine1 = nn.Linear(feature_dim, attention_dim, bias=True)
        self.affine2 = nn.Linear(attention_dim, 1, bias=False)

    def initialize(self):
        nn.init.xavier_uniform_(self.affine1.weight, gain=nn.init.calculate_gain('tanh'))
        nn.init.zeros_(self.affine1.bias)
        nn.init.xavier_uniform_(self.affine2.weight)

    def forward(self, feature, mask=None):
        n_seq, S, D = feature.shape
        x = feature.reshape(n_seq * S, D)
        hidden = ops.linear(x, self.affine1.weight, self.affine1.bias, act='tanh')
        return ops.additive_pool(hidden, self.affine2.weight.view(-1), x, n_seq, S, mask=mask)
