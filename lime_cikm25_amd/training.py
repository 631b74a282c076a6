"""The training step of LIME-CROWN-CROWN: a differentiable ``Model.forward`` for
``loss.backward()`` (reference trainer.py:131-148), and the native step that replaces trainer.py:143-148 --
``zero_grad / backward / clip_grad_norm_ / Adam.step`` -- plus the gradient all-reduce of a data-parallel run.

Where the arithmetic runs
  * token encoders (word gather + positional table + post-LN encoder layer + mean pool, 97 % of a step's FLOPs):
    hand-written HIP forward AND backward behind one autograd node (``_TokenEncoder``): the forward of the scoring path
    with the LayerNorm rstd kept, backward = LayerNorm / attention / ReLU backward kernels, the MFMA weight-gradient GEMM,
    input gradients on lime_linear_f32 with transposed weights, word-table scatter-add.
  * every nn.Linear of the tail and the user encoder: ``_Linear`` (HIP GEMMs forward and backward).
  * the fused tail kernels of the scoring path with hand-written backward kernels (csrc/tail_backward_f32.hip): intent attention
    + cosine similarity + concat, gated residual + LayerNorm, history-vs-candidate attention + dot product + lifetime weight.
  * candidate-aware attention weights (per-head softmax with the layer's p = 0.2 dropout, query-norm softmax, aggregate
    softmax) forward and backward in one kernel per impression row (cand_attn_train_kernel).
  * still torch ops with torch's autograd: the GraphSAGE mean, a few concatenations and the two feature-fusion dropouts.

Dropout (the reference trains with dropout_rate 0.2): the six dropouts inside a token encoder run on the dropout kernels
with counter-based masks (csrc/dropout.h; torch's Philox stream is not reproduced -- the arithmetic is pinned against a torch
statement of the layer fed with the same masks, tests/test_dropout_gpu.py), and so does the hard-coded p = 0.2 of the
candidate-aware attention.  The dropouts left in the torch glue (feature_fusion, user_node_embedding) use torch's generator.
"""
import math

import torch
import torch.nn as nn

from . import distributed, ops


# ---------------------------------------------------------------------------------------------------------------------
# autograd nodes over the HIP kernels
# ---------------------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) on lime_linear_f32; backward: dX on lime_linear_f32 (W^T as the weight operand), dW on the
    weight-gradient GEMM, db as a column sum."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x if x.stride(-1) == 1 else x.contiguous()
        y = ops.linear(x, w, b, act=act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act == 'relu':
            dy = ops.relu_bwd_(dy.clone(), y)
        elif ctx.act == 'tanh':
            dy = dy * (1.0 - y * y)
        elif ctx.act == 'sigmoid':
            dy = dy * y * (1.0 - y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy, w.t().contiguous(), None)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(dy, x, want_bias=want_b)
            if want_b:
                dw, db = dw
        elif want_b:
            db = ops.colsum(dy)
        return dx, dw, db, None


def linear(x, lin, act=None):
    """nn.Linear ``lin`` applied to the last axis of x (any leading shape)."""
    shape = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), lin.weight, lin.bias, act)
    return y.view(*shape, -1)


class _GatherRows(torch.autograd.Function):
    """table[idx] (nn.Embedding forward) on the gather kernel; backward = the scatter-add kernel."""

    @staticmethod
    def forward(ctx, idx, table, hot_id):
        out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
        ops.gather_rows(idx, table, out)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(table.shape)
        ctx.hot_id = hot_id
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.shape, dtype=torch.float32, device=dy.device)
        ops.embed_bwd(idx, dy.contiguous(), dtable, hot_id=ctx.hot_id)
        return None, dtable, None


def embedding(table, idx, hot_id=-1):
    """table[idx] for an integer tensor idx of any shape -> [*idx.shape, dim].  ``hot_id``: a row that a large share of the
    indices hit (the padding word), summed per wave before it reaches memory in the backward."""
    flat = idx.reshape(-1)
    flat = (flat if flat.dtype == torch.int32 else flat.to(torch.int32)).contiguous()
    if table.requires_grad and torch.is_grad_enabled():
        out = _GatherRows.apply(flat, table, hot_id)
    else:
        out = torch.empty((flat.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
        ops.gather_rows(flat, table, out)
    return out.view(*idx.shape, table.shape[1])


class _IntentFuse(torch.autograd.Function):
    """[title_i | sim * body_i] [M, 2D] from the k intents of title and body (newsEncoders.py:355-371) on the fused kernel."""

    @staticmethod
    def forward(ctx, intents, hidden, a2_t, a2_b, M, k):
        D, A = intents.shape[1], hidden.shape[1]
        intents, hidden = intents.contiguous(), hidden.contiguous()
        a2_t, a2_b = a2_t.reshape(-1).contiguous(), a2_b.reshape(-1).contiguous()
        out = torch.empty((M, 2 * D), dtype=torch.float32, device=intents.device)
        ops.intent_fuse(intents, hidden, a2_t, a2_b, out, M, k, D, A)
        ctx.dims = (M, k, D, A)
        ctx.save_for_backward(intents, hidden, a2_t, a2_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        intents, hidden, a2_t, a2_b = ctx.saved_tensors
        M, k, D, A = ctx.dims
        d_int, d_hid, da_t, da_b = ops.intent_fuse_bwd(intents, hidden, a2_t, a2_b, dout.contiguous(), M, k, D, A)
        return d_int, d_hid, da_t, da_b, None, None


class _GateLN(torch.autograd.Function):
    """LayerNorm(g s x + (1 - g) x), g = sigmoid(s y + bias) (layers.py:84-89); y = W_g x comes from the caller's GEMM."""

    @staticmethod
    def forward(ctx, y, x, scale, bias, gamma, beta, eps):
        y, x, scale = y.contiguous(), x.contiguous(), scale.contiguous()
        out = ops.gate_ln(y, x, scale, bias, gamma, beta, eps)
        ctx.eps = eps
        ctx.save_for_backward(y, x, scale, bias, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, x, scale, bias, gamma, beta = ctx.saved_tensors
        dy, dx, dscale, dbias, dgamma, dbeta = ops.gate_ln_bwd(y, x, scale, bias, gamma, beta, ctx.eps, dout)
        return dy.view_as(y), dx.view_as(x), dscale.view_as(scale), dbias, dgamma, dbeta, None


class _CandAttnWeights(torch.autograd.Function):
    """agg [B, H] of CandidateAware_ClickedNewsAttention (layers.py:66-81) from the topic projections, with the layer's own
    dropout (p = 0.2, layers.py:36,74) on the per-head probabilities in training mode."""

    @staticmethod
    def forward(ctx, qp, kp, mask, dims, n_head, p, seed):
        B, N, H, D = dims
        qp, kp, mask = qp.contiguous(), kp.contiguous(), mask.contiguous()
        agg = ops.cand_attn_weights_train(qp.view(-1), kp.view(-1), mask, B, N, H, D, n_head, p, seed, 0)
        ctx.cfg = (dims, n_head, p, seed)
        ctx.save_for_backward(qp, kp, mask)
        return agg

    @staticmethod
    def backward(ctx, dagg):
        qp, kp, mask = ctx.saved_tensors
        (B, N, H, D), n_head, p, seed = ctx.cfg
        dqp, dkp = ops.cand_attn_weights_bwd(qp.view(-1), kp.view(-1), mask, dagg, B, N, H, D, n_head, p, seed, 0)
        return dqp.view_as(qp), dkp.view_as(kp), None, None, None, None, None


class _Dropout(torch.autograd.Function):
    """nn.Dropout on the counter-based masks (csrc/dropout.h): the backward re-applies the mask of (seed, site)."""

    @staticmethod
    def forward(ctx, x, p, seed, site):
        ctx.cfg = (p, seed, site)
        return ops.dropout(x.contiguous(), p, seed, site)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site = ctx.cfg
        return ops.dropout(dy.contiguous(), p, seed, site), None, None, None


class _MaskedAttention(torch.autograd.Function):
    """The attention core of layers.MultiHeadAttention (layers.py:227-237): softmax(Q K^T / sqrt(d_k), key mask -1e9) V on the
    packed [tokens, 3 h d_k] projections."""

    @staticmethod
    def forward(ctx, qkv, mask, n_seq, S, h, dk):
        W = h * dk
        out = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], n_seq, S, h, dk, 1.0 / math.sqrt(float(dk)), key_mask=mask)
        ctx.dims = (n_seq, S, h, dk)
        ctx.save_for_backward(qkv, mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, mask = ctx.saved_tensors
        n_seq, S, h, dk = ctx.dims
        W = h * dk
        dqkv = ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dout.contiguous(), n_seq, S, h, dk,
                                       1.0 / math.sqrt(float(dk)), key_mask=mask)
        return dqkv, None, None, None, None, None


class _AdditivePool(torch.autograd.Function):
    """layers.Attention over the tokens of a sequence (layers.py:285-300 as newsEncoders.py:591-592 calls it): rep [M, D] =
    sum_t softmax_t(mask(hidden_t . a2)) x_t, one kernel each way (lime_additive_pool_f32 / _bwd_f32)."""

    @staticmethod
    def forward(ctx, hidden, a2, x, mask, n_seq, S):
        hidden, x, a2 = hidden.contiguous(), x.contiguous(), a2.contiguous()
        ctx.dims = (n_seq, S)
        ctx.save_for_backward(hidden, a2, x, mask)
        return ops.additive_pool(hidden, a2, x, n_seq, S, mask=mask)

    @staticmethod
    def backward(ctx, dout):
        hidden, a2, x, mask = ctx.saved_tensors
        n_seq, S = ctx.dims
        dh, da2, dx = ops.additive_pool_bwd(hidden, a2, x, dout.contiguous(), n_seq, S, mask=mask)
        return dh, da2, dx, None, None, None


class _InterestMatch(torch.autograd.Function):
    """logits [B, N] = (softmax_h(kp . qp / sqrt(A)) g) . cand * lifetime weight (userEncoders.py:158-169, util.py:23-49)."""

    @staticmethod
    def forward(ctx, kp, qp, g, cand, remaining, dims, scale, alpha, beta, use_weight, use_penalty):
        B, N, H, A, D = dims
        kp, qp, g, cand = kp.contiguous(), qp.contiguous(), g.contiguous(), cand.contiguous()
        remaining = remaining.contiguous()
        _, logits = ops.interest_match(kp.view(-1), qp.view(-1), g.view(-1), cand.view(-1), remaining, B, N, H, A, D, scale, alpha, beta,
                                       use_weight, use_penalty, want_logits=True, want_user=False)
        ctx.cfg = (dims, scale, alpha, beta, use_weight, use_penalty)
        ctx.save_for_backward(kp, qp, g, cand, remaining)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        kp, qp, g, cand, remaining = ctx.saved_tensors
        (B, N, H, A, D), scale, alpha, beta, use_weight, use_penalty = ctx.cfg
        dkp, dqp, dg, dcand = ops.interest_match_bwd(kp.view(-1), qp.view(-1), g.view(-1), cand.view(-1), remaining, dlogits, B, N, H, A, D,
                                                     scale, alpha, beta, use_weight, use_penalty)
        return dkp.view_as(kp), dqp.view_as(qp), dg.view_as(g), dcand.view_as(cand), None, None, None, None, None, None, None


def _unpad_heads(t, groups, hd, hs):
    """[groups * hs, ...] -> [groups * hd, ...]: drop the padding rows ``ops.pad_heads`` inserted."""
    if hs == hd:
        return t
    rest = t.shape[1:]
    return t.view(groups, hs, *rest)[:, :hd].reshape(groups * hd, *rest)


# dropout sites of one encoder call (csrc/dropout.h): the mask of a site is a function of (seed, site, element index)
_SITE_EMB, _SITE_PE, _SITE_ATTN, _SITE_DROP1, _SITE_FF, _SITE_DROP2 = range(6)


class _TokenEncoder(torch.autograd.Function):
    """pooled [M, E] = mean_S(EncoderLayer(E[ids] + PE))  (newsEncoders.py:311-321 for one of title / body).

    p = 0: the fused forward of the scoring path (word gather inside the in_proj GEMM, residual + LayerNorm in the GEMM
    epilogues).  p > 0 (training-mode dropout, seeded per call): the same layer with its six dropouts -- word embeddings and
    positional sum (:311-312, :827), attention probabilities, dropout1 / dropout / dropout2 of the encoder layer -- on the
    dropout kernels; masks are regenerated in the backward from (seed, site)."""

    @staticmethod
    def forward(ctx, ids, nhead, eps1, eps2, p, seed, table, pe, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w,
                n2_b, *live):
        # live (optional, p = 0): (ids, rows) of the non-padding tokens (int32 device tensors, lime_compact_sequences' tok_ids / tok_rows cut
        # to their count): in_proj then runs over those only and the padding tokens' q / k / v rows -- a function of the position alone --
        # are copied in from S rows (the reference's batch is 72 % padding tokens: newsEncoders.py:311-312 embeds them all the same)
        ctx.n_extra = len(live)
        live = live[0] if live else None
        M, S = ids.shape
        E = table.shape[1]
        hd = E // nhead
        hs = 32 if hd <= 32 else hd
        if hs > 32 or S > 512:
            raise NotImplementedError('the attention kernels cover head_dim <= 32 and S <= 512 (got %d, %d)' % (hd, S))
        W = nhead * hs
        flat = ids.reshape(-1).contiguous()
        tok = M * S
        dev = table.device
        scale = 1.0 / math.sqrt(hd)
        w_in = ops.pad_heads(in_w, 3 * nhead, hd, hs) if hs != hd else in_w
        b_in = ops.pad_heads(in_b, 3 * nhead, hd, hs) if hs != hd else in_b
        x0 = None
        if p > 0:
            x0 = ops.embed_pe_dropout(flat, table, pe, S, p, seed, _SITE_EMB, _SITE_PE)
            qkv = ops.linear(x0, w_in, b_in, n_alg=3 * E)
            ao = ops.token_attention_dropout(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, scale, p, seed, _SITE_ATTN,
                                             head_stride=hs)
            x1, rstd1 = ops.dropout_add_layernorm(ops.linear(ao, out_w, out_b), x0, n1_w, n1_b, eps1, p, seed, _SITE_DROP1)
            h = ops.linear(x1, l1_w, l1_b, act='relu', dropout=(p, seed, _SITE_FF))     # the dropout behind the ReLU in the GEMM's epilogue
            y, rstd2 = ops.dropout_add_layernorm(ops.linear(h, l2_w, l2_b), x1, n2_w, n2_b, eps2, p, seed, _SITE_DROP2)
        else:
            pew = ops.linear(pe[:S], w_in, b_in)
            if live is not None and live[0].numel() >= 4096:
                tok_ids, tok_rows = live
                qkv = torch.empty((tok, 3 * W), dtype=torch.float32, device=dev)
                ops.linear(table, w_in, None, a_ids=tok_ids, res=pew, res_mod=S, n_alg=3 * E, c_ids=tok_rows, out=qkv)
                pad_rows = ops.linear(table, w_in, None, a_ids=torch.zeros(S, dtype=torch.int32, device=dev), res=pew, res_mod=S)
                ops.fill_pad_rows(flat, pad_rows, qkv, S)
            else:
                qkv = ops.linear(table, w_in, None, a_ids=flat, res=pew, res_mod=S, n_alg=3 * E)
            # S > 128: the forward keeps its softmax statistics, the blocked backward does not recompute them
            lse = torch.empty(tok * nhead, dtype=torch.float32, device=dev) if S > 128 else None
            ao = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, scale, head_stride=hs, lse=lse)
            rstd1 = torch.empty(tok, dtype=torch.float32, device=dev)
            x1 = ops.linear(ao, out_w, out_b, res=table, res_ids=flat, res_pe=pe, res_period=S, ln=(n1_w, n1_b), ln_eps=eps1,
                            ln_rstd=rstd1)
            h = ops.linear(x1, l1_w, l1_b, act='relu')
            rstd2 = torch.empty(tok, dtype=torch.float32, device=dev)
            y = ops.linear(h, l2_w, l2_b, res=x1, ln=(n2_w, n2_b), ln_eps=eps2, ln_rstd=rstd2)
        pooled = ops.mean_pool(y, M, S)
        ctx.dims = (M, S, E, nhead, hd, hs, p, seed)
        ctx.lse = lse if p == 0 else None
        ctx.save_for_backward(flat, table, pe, w_in, out_w, l1_w, l2_w, n1_w, n1_b, n2_w, n2_b, qkv, ao, x1, rstd1, h, y, rstd2, x0)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        (flat, table, pe, w_in, out_w, l1_w, l2_w, n1_w, n1_b, n2_w, n2_b, qkv, ao, x1, rstd1, h, y, rstd2, x0) = ctx.saved_tensors
        M, S, E, nhead, hd, hs, p, seed = ctx.dims
        W = nhead * hs
        drop = p > 0
        keep_scale = 1.0 / (1.0 - p) if drop else 1.0
        dpooled = dpooled.contiguous()
        # norm2 <- mean pool: every token of a sequence receives dpooled / S
        if drop:                                                                   # dt2: the branch through dropout2 into linear2, same pass
            dz2, dn2_w, dn2_b, dl2_b, dt2 = ops.layernorm_bwd(dpooled, y, n2_w, n2_b, rstd2, dy_div=S, dy_scale=1.0 / S,
                                                             dropout=(p, seed, _SITE_DROP2))            # dl2_b: column sums of dt2
        else:
            dz2, dn2_w, dn2_b, dl2_b = ops.layernorm_bwd(dpooled, y, n2_w, n2_b, rstd2, dy_div=S, dy_scale=1.0 / S)
            dt2 = dz2
        del y
        dl2_w = ops.linear_wgrad(dt2, h)
        # dH = dT2 W2 with the ReLU (and dropout) gradient in the GEMM's epilogue: h > 0 <=> ReLU passed and the mask kept
        dh = ops.linear(dt2, l2_w.t().contiguous(), None, act='relu_grad', res=h, act_scale=keep_scale)
        del h, dt2
        dl1_w, dl1_b = ops.linear_wgrad(dh, x1, want_bias=True)
        dx1 = ops.linear(dh, l1_w.t().contiguous(), None, res=dz2)                 # through linear1 + the residual branch
        del dh, dz2
        if drop:
            dz1, dn1_w, dn1_b, dout_b, dt1 = ops.layernorm_bwd(dx1, x1, n1_w, n1_b, rstd1, dropout=(p, seed, _SITE_DROP1))
        else:
            dz1, dn1_w, dn1_b, dout_b = ops.layernorm_bwd(dx1, x1, n1_w, n1_b, rstd1)
            dt1 = dz1
        del dx1, x1
        dout_w = ops.linear_wgrad(dt1, ao)
        dao = ops.linear(dt1, out_w.t().contiguous(), None)
        dqkv = ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dao, M, S, nhead, hd, 1.0 / math.sqrt(hd),
                                       head_stride=hs, out=ao, dropout=(p, seed, _SITE_ATTN) if drop else None, lse=ctx.lse)
        del dao, qkv, ao, dt1
        if x0 is None:
            x0 = ops.embed_pe(flat, table, pe, S)                                  # the layer input, re-gathered
        din_w, din_b = ops.linear_wgrad(dqkv, x0, want_bias=True)
        din_w, din_b = _unpad_heads(din_w, 3 * nhead, hd, hs), _unpad_heads(din_b, 3 * nhead, hd, hs)
        del x0
        dx0 = ops.linear(dqkv, w_in.t().contiguous(), None, res=dz1)               # through in_proj + the residual branch
        dtable = None
        if ctx.needs_input_grad[6]:
            if drop:                                                               # back through the two input dropouts, one pass
                ops.dropout2(dx0, p, seed, _SITE_PE, _SITE_EMB, out=dx0)
            dtable = torch.zeros_like(table)
            ops.embed_bwd(flat, dx0, dtable, hot_id=0)
        return (None, None, None, None, None, None, dtable, None, din_w, din_b, dout_w, dout_b, dl1_w, dl1_b, dl2_w, dl2_b, dn1_w,
                dn1_b, dn2_w, dn2_b) + (None,) * ctx.n_extra


class _EmbedPE(torch.autograd.Function):
    """x0 [M S, E] = drop(drop(E[ids]) + PE[t]) (newsEncoders.py:311-312, :827): the layer input, materialised -- the entry of the
    num_layers > 1 path (the single-layer node above gathers it inside its in_proj GEMM)."""

    @staticmethod
    def forward(ctx, ids, table, pe, p, seed):
        M, S = ids.shape
        flat = ids.reshape(-1).contiguous()
        ctx.cfg = (S, p, seed)
        ctx.save_for_backward(flat, table)
        if p > 0:
            return ops.embed_pe_dropout(flat, table, pe, S, p, seed, _SITE_EMB, _SITE_PE)
        return ops.embed_pe(flat, table, pe, S)

    @staticmethod
    def backward(ctx, dx0):
        flat, table = ctx.saved_tensors
        S, p, seed = ctx.cfg
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        dx0 = dx0.contiguous()
        if p > 0:                                                                  # back through the two input dropouts
            dx0 = ops.dropout2(dx0, p, seed, _SITE_PE, _SITE_EMB)
        dtable = torch.zeros_like(table)
        ops.embed_bwd(flat, dx0, dtable, hot_id=0)
        return None, dtable, None, None, None


class _EncoderLayer(torch.autograd.Function):
    """y [M S, E] = one post-LN TransformerEncoderLayer over a MATERIALISED input x (newsEncoders.py:244-247): what a layer behind the
    first needs (config.py:70 allows num_layers = 2), and the first one on that path too.  Same kernels as ``_TokenEncoder`` -- in_proj,
    attention, out_proj + residual + LayerNorm, linear1 + ReLU, linear2 + residual + LayerNorm, with the four in-layer dropouts on the
    counter-based masks when p > 0 -- with a dense residual instead of the gathered one, and dx as the input gradient."""

    @staticmethod
    def forward(ctx, x, M, S, nhead, eps1, eps2, p, seed, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b):
        x = x.contiguous()
        E = x.shape[1]
        hd = E // nhead
        hs = 32 if hd <= 32 else hd
        if hs > 32 or S > 512:
            raise NotImplementedError('the attention kernels cover head_dim <= 32 and S <= 512 (got %d, %d)' % (hd, S))
        W = nhead * hs
        tok = M * S
        dev = x.device
        scale = 1.0 / math.sqrt(hd)
        w_in = ops.pad_heads(in_w, 3 * nhead, hd, hs) if hs != hd else in_w
        b_in = ops.pad_heads(in_b, 3 * nhead, hd, hs) if hs != hd else in_b
        qkv = ops.linear(x, w_in, b_in, n_alg=3 * E)
        if p > 0:
            ao = ops.token_attention_dropout(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, scale, p, seed, _SITE_ATTN,
                                             head_stride=hs)
            x1, rstd1 = ops.dropout_add_layernorm(ops.linear(ao, out_w, out_b), x, n1_w, n1_b, eps1, p, seed, _SITE_DROP1)
            h = ops.linear(x1, l1_w, l1_b, act='relu', dropout=(p, seed, _SITE_FF))     # the dropout behind the ReLU in the GEMM's epilogue
            y, rstd2 = ops.dropout_add_layernorm(ops.linear(h, l2_w, l2_b), x1, n2_w, n2_b, eps2, p, seed, _SITE_DROP2)
        else:
            lse = torch.empty(tok * nhead, dtype=torch.float32, device=dev) if S > 128 else None
            ao = ops.token_attention(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], M, S, nhead, hd, scale, head_stride=hs, lse=lse)
            rstd1 = torch.empty(tok, dtype=torch.float32, device=dev)
            x1 = ops.linear(ao, out_w, out_b, res=x, ln=(n1_w, n1_b), ln_eps=eps1, ln_rstd=rstd1)
            h = ops.linear(x1, l1_w, l1_b, act='relu')
            rstd2 = torch.empty(tok, dtype=torch.float32, device=dev)
            y = ops.linear(h, l2_w, l2_b, res=x1, ln=(n2_w, n2_b), ln_eps=eps2, ln_rstd=rstd2)
        ctx.dims = (M, S, E, nhead, hd, hs, p, seed)
        ctx.lse = lse if p == 0 else None
        ctx.save_for_backward(x, w_in, out_w, l1_w, l2_w, n1_w, n1_b, n2_w, n2_b, qkv, ao, x1, rstd1, h, y, rstd2)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, w_in, out_w, l1_w, l2_w, n1_w, n1_b, n2_w, n2_b, qkv, ao, x1, rstd1, h, y, rstd2) = ctx.saved_tensors
        M, S, E, nhead, hd, hs, p, seed = ctx.dims
        W = nhead * hs
        drop = p > 0
        keep_scale = 1.0 / (1.0 - p) if drop else 1.0
        if drop:                                                                   # dt2: the branch through dropout2 into linear2, same pass
            dz2, dn2_w, dn2_b, dl2_b, dt2 = ops.layernorm_bwd(dy.contiguous(), y, n2_w, n2_b, rstd2, dropout=(p, seed, _SITE_DROP2))
        else:
            dz2, dn2_w, dn2_b, dl2_b = ops.layernorm_bwd(dy.contiguous(), y, n2_w, n2_b, rstd2)
            dt2 = dz2
        dl2_w = ops.linear_wgrad(dt2, h)
        # dH = dT2 W2 with the ReLU (and dropout) gradient in the GEMM's epilogue: h > 0 <=> ReLU passed and the mask kept
        dh = ops.linear(dt2, l2_w.t().contiguous(), None, act='relu_grad', res=h, act_scale=keep_scale)
        dl1_w, dl1_b = ops.linear_wgrad(dh, x1, want_bias=True)
        dx1 = ops.linear(dh, l1_w.t().contiguous(), None, res=dz2)                 # through linear1 + the residual branch
        del dh, dz2
        if drop:
            dz1, dn1_w, dn1_b, dout_b, dt1 = ops.layernorm_bwd(dx1, x1, n1_w, n1_b, rstd1, dropout=(p, seed, _SITE_DROP1))
        else:
            dz1, dn1_w, dn1_b, dout_b = ops.layernorm_bwd(dx1, x1, n1_w, n1_b, rstd1)
            dt1 = dz1
        del dx1
        dout_w = ops.linear_wgrad(dt1, ao)
        dao = ops.linear(dt1, out_w.t().contiguous(), None)
        dqkv = ops.token_attention_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dao, M, S, nhead, hd, 1.0 / math.sqrt(hd),
                                       head_stride=hs, out=ao, dropout=(p, seed, _SITE_ATTN) if drop else None, lse=ctx.lse)
        del dao, dt1
        din_w, din_b = ops.linear_wgrad(dqkv, x, want_bias=True)
        din_w, din_b = _unpad_heads(din_w, 3 * nhead, hd, hs), _unpad_heads(din_b, 3 * nhead, hd, hs)
        dx = ops.linear(dqkv, w_in.t().contiguous(), None, res=dz1)                # through in_proj + the residual branch
        return (dx, None, None, None, None, None, None, None, din_w, din_b, dout_w, dout_b, dl1_w, dl1_b, dl2_w, dl2_b, dn1_w, dn1_b,
                dn2_w, dn2_b)


class _SeqExpand(torch.autograd.Function):
    """pooled[s] = pooled_c[seq_inv[s]]: the compact sequences' pooled vectors back on every slot (all-padding slots share the
    representative's).  Backward: every live compact row has exactly one slot (a gather by seq_src), the representative's
    gradient is the sum over the all-padding slots -- a fixed-order torch reduction, no atomics."""

    @staticmethod
    def forward(ctx, pooled_c, seq_inv, seq_src, n_live):
        out = torch.empty((seq_inv.numel(), pooled_c.shape[1]), dtype=torch.float32, device=pooled_c.device)
        ops.gather_rows(seq_inv, pooled_c.contiguous(), out)
        ctx.save_for_backward(seq_inv, seq_src)
        ctx.n_live = n_live
        return out

    @staticmethod
    def backward(ctx, dout):
        seq_inv, seq_src = ctx.saved_tensors
        n_live = ctx.n_live
        dout = dout.contiguous()
        d = torch.empty((n_live + 1, dout.shape[1]), dtype=torch.float32, device=dout.device)
        if n_live:
            ops.gather_rows(seq_src[:n_live].contiguous(), dout, d[:n_live])
        pad = (seq_inv == n_live).to(dout.dtype).unsqueeze(1)
        d[n_live] = (dout * pad).sum(dim=0)
        return d, None, None, None


# Training with every dropout off (p = 0): an all-padding sequence pools to the same vector in every slot, so the encoder runs on
# the live sequences + one representative, forward and backward (the representative collects the gradients of all its slots).
# With dropout on every slot draws its own masks (the reference's behaviour) and nothing is shared.  LIME_DENSE_TOKENS=1 turns
# it off, as on the scoring path.
def _dedup_sequences(ids):
    from . import newsEncoders
    if not newsEncoders.DEDUP or ids.shape[0] < 64:
        return None
    cmp = ops.compact_sequences(ids)
    n_c, _, n_tok, n_live = (int(v) for v in cmp.counts[:4].tolist())  # one host read per encoder call (the step is eager)
    if n_c >= ids.shape[0]:
        return None                                                    # nothing repeats
    return cmp, n_c, n_live, n_tok


def _draw_seed():
    """A fresh 62-bit seed from torch's CPU generator (torch.manual_seed makes a run repeatable)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def encode_tokens(ids, table, pos_encoder, transformer, nhead, p_embedding=0.0):
    """``p_embedding``: the probability of the inplace dropout the caller's encoder applies to the word embeddings (0 in eval
    mode).  The HIP layer applies ONE probability to its six dropout sites, which is how the reference builds it
    (config.dropout_rate everywhere); anything else is refused."""
    if transformer.norm is not None:
        raise NotImplementedError('a final encoder norm is not used by the reference (newsEncoders.py:245,247)')
    layers = list(transformer.layers)
    ps = [p_embedding, pos_encoder.dropout.p if pos_encoder.training else 0.0]
    for layer in layers:
        ps += [layer.self_attn.dropout, layer.dropout1.p, layer.dropout.p, layer.dropout2.p] if layer.training else [0.0] * 4
    if max(ps) != min(ps):
        raise NotImplementedError('the HIP encoder layer applies one dropout probability to all of its sites; got %s (embedding, '
                                  'positional, then attention, dropout1, dropout, dropout2 per layer): put the news encoder into one mode' % ps)
    p = float(ps[0])
    ids = ids.contiguous()
    dd = _dedup_sequences(ids) if p == 0 else None
    if len(layers) == 1:
        layer = layers[0]
        sa = layer.self_attn
        run = lambda rows, *live: _TokenEncoder.apply(rows, nhead, layer.norm1.eps, layer.norm2.eps, p, _draw_seed() if p > 0 else 0, table,
                                                      pos_encoder.table(), sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight,
                                                      sa.out_proj.bias, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight,
                                                      layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight,
                                                      layer.norm2.bias, *live)
    else:
        def run(rows):                                  # num_layers = 2 (config.py:70): materialised input, one node per layer, mean pool
            M, S = rows.shape
            x = _EmbedPE.apply(rows, table, pos_encoder.table(), p, _draw_seed() if p > 0 else 0)
            for layer in layers:
                sa = layer.self_attn
                x = _EncoderLayer.apply(x, M, S, nhead, layer.norm1.eps, layer.norm2.eps, p, _draw_seed() if p > 0 else 0,
                                        sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.linear1.weight,
                                        layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias,
                                        layer.norm2.weight, layer.norm2.bias)
            return x.view(M, S, -1).mean(dim=1)                                                                 # :317 / :321
    if dd is None:
        return run(ids)
    cmp, n_c, n_live, n_tok = dd
    S = ids.shape[1]
    if len(layers) == 1:                                 # the live tokens' (id, compact row) lists: in_proj over them only
        pooled_c = run(cmp.ids_c[:n_c * S].view(n_c, S), (cmp.tok_ids[:n_tok], cmp.tok_rows[:n_tok]))
    else:
        pooled_c = run(cmp.ids_c[:n_c * S].view(n_c, S))
    return _SeqExpand.apply(pooled_c, cmp.seq_inv, cmp.seq_src, n_live)


# ---------------------------------------------------------------------------------------------------------------------
# the differentiable forward
# ---------------------------------------------------------------------------------------------------------------------
def crown_tail(enc, title_p, body_p, category, subCategory):
    """newsEncoders.CROWN.forward after the token encoders (newsEncoders.py:340-373): pooled title / body [M, 300] -> [M, 900]."""
    cat_e = embedding(enc.category_embedding.weight, category)
    sub_e = embedding(enc.subCategory_embedding.weight, subCategory)
    cat_rep = linear(torch.cat([cat_e, sub_e], dim=1), enc.category_affine)                                     # :340-342
    w_int = torch.cat([lin.weight for lin in enc.intent_layers], dim=0)
    b_int = torch.cat([lin.bias for lin in enc.intent_layers], dim=0)
    k, D = enc.intent_num, enc.intent_embedding_dim

    M = title_p.shape[0]
    xin = torch.cat([torch.cat([title_p, cat_rep], dim=1), torch.cat([body_p, cat_rep], dim=1)], dim=0)         # :343-344
    iv = _Linear.apply(xin, w_int, b_int, 'relu').view(2 * M * k, D)                                            # :284-295
    hidden = torch.cat([linear(iv[:M * k], enc.title_intent_attention.affine1, act='tanh'),
                        linear(iv[M * k:], enc.body_intent_attention.affine1, act='tanh')], dim=0)              # layers.py:288
    fused = _IntentFuse.apply(iv, hidden, enc.title_intent_attention.affine2.weight.view(-1),
                              enc.body_intent_attention.affine2.weight.view(-1), M, k)                                                                             # :355-371
    return torch.cat([fused, enc.dropout(cat_e.clone()), enc.dropout(sub_e.clone())], dim=1)                    # :221-225


def pooled_tokens(enc, title_text, content_text):
    """The two token encoders of the CROWN content encoder ``enc`` (newsEncoders.py:311-321) on M flat news -> 2 x [M, 300]."""
    table = enc.word_embedding.weight
    p_emb = enc.dropout.p if enc.training else 0.0                                                              # :311-312
    title_p = encode_tokens(title_text, table, enc.title_pos_encoder, enc.title_transformer, enc.head_num, p_emb)   # :311-317
    body_p = encode_tokens(content_text, table, enc.body_pos_encoder, enc.body_transformer, enc.head_num, p_emb)    # :312-321
    return title_p, body_p


def mhsa_content(enc, title_text, title_mask, category, subCategory):
    """newsEncoders.MHSA.forward (newsEncoders.py:582-595; title only) on M flat news -> [M, h d_k + 100]."""
    M, T = title_text.shape
    if T > 128:
        raise NotImplementedError('the masked attention backward covers sequences of at most 128 tokens (got %d)' % T)
    mha, att = enc.multiheadAttention, enc.attention
    if mha.d_k != mha.d_v:
        raise NotImplementedError('d_k != d_v')
    p = float(enc.dropout.p) if enc.training else 0.0
    seed = _draw_seed() if p > 0 else 0
    x = embedding(enc.word_embedding.weight, title_text, hot_id=0).view(M * T, -1)                              # :588
    if p > 0:
        x = _Dropout.apply(x, p, seed, 0)
    w = torch.cat([mha.W_Q.weight, mha.W_K.weight, mha.W_V.weight], dim=0)
    b = torch.cat([mha.W_Q.bias, mha.W_K.bias, mha.W_V.bias], dim=0)
    mask = title_mask.contiguous()
    c = _MaskedAttention.apply(_Linear.apply(x, w, b, None), mask, M, T, mha.h, mha.d_k)                         # :590, layers.py:222-238
    if p > 0:
        c = _Dropout.apply(c, p, seed, 1)
    hidden = linear(c, att.affine1, act='tanh')                                                                 # layers.py:288
    rep = _AdditivePool.apply(hidden, att.affine2.weight.view(-1), c, mask, M, T)                               # :592, layers.py:289-300
    cat_e = embedding(enc.category_embedding.weight, category)
    sub_e = embedding(enc.subCategory_embedding.weight, subCategory)
    return torch.cat([rep, enc.dropout(cat_e.clone()), enc.dropout(sub_e.clone())], dim=1)                      # :594


def lime_tail(ne, content, freshness, lifetime):
    """LIME.forward, fusion 'concat' (newsEncoders.py:140-153), from the content encoder's output -> [M, 400]."""
    fe = ne.freshness_encoder
    fb = fe.buckets(freshness)
    lb = fe.buckets(lifetime)
    fresh = linear(torch.cat([embedding(fe.freshness_embedding.weight, fb), embedding(fe.lifetime_embedding.weight, lb)], dim=1),
                   fe.dense, act='tanh')
    if ne.fusion_method == 'add':                                                                                   # :154-155
        return content + fresh
    fused = torch.cat([content, fresh], dim=1)
    if ne.fusion_method == 'gated':                                                                                 # :156-159
        gate = linear(fused, ne.gate, act='sigmoid')
        return gate * content + (1 - gate) * fresh
    return fused if isinstance(ne.project, nn.Identity) else linear(fused, ne.project)


def _topic(ne, category, subCategory):
    """userEncoders.py:103-105 / :115-117."""
    x = torch.cat([embedding(ne.category_embedding.weight, category), embedding(ne.subCategory_embedding.weight, subCategory)], dim=-1)
    return linear(x, ne.category_affine)


def candidate_aware(att, hist, hist_topic, cand_topic, mask):
    """CandidateAware_ClickedNewsAttention.forward (layers.py:52-93) -> (refined history [B, H, D], attn_weights_agg [B, H]); the
    value_proj branch is dead there."""
    B, H, D = hist.shape
    N = cand_topic.shape[1]
    if mask is None:
        mask = torch.ones(B, H, dtype=torch.bool, device=hist.device)
    p = float(att.dropout.p) if att.training else 0.0                                                       # layers.py:36,74
    agg = _CandAttnWeights.apply(linear(cand_topic, att.query_proj).reshape(B * N, D), linear(hist_topic, att.key_proj).reshape(B * H, D),
                                 mask, (B, N, H, D), att.num_heads, p, _draw_seed() if p > 0 else 0)           # :66-81
    if not att.use_residual_connection:
        return agg.unsqueeze(-1) * hist, agg
    ln = att.layernorm
    flat = hist.reshape(B * H, D)
    y = _Linear.apply(flat, att.gate_proj.weight, None, None)                       # W_g x; the row scale agg commutes (:87)
    return _GateLN.apply(y, flat, agg.reshape(-1), att.gate_proj.bias, ln.weight, ln.bias, ln.eps).view(B, H, D), agg


def wants_train_path(module, p=0.0):
    """A sub-module called on its own (``news_encoder(...)``, ``user_encoder(...)``, ``candidate_aware_attn(...)``) takes the
    differentiable path of this file when it is in training mode and either autograd is recording or a dropout is active;
    otherwise the fused scoring kernels compute the same function."""
    return module.training and (torch.is_grad_enabled() or p > 0)


def _interest_inputs(ue, hist, cand, category, subCategory, user_category, user_subCategory, user_history_mask, n_src=None):
    """userEncoders.py:103-105, :114-162: candidate-aware refinement, user nodes + GraphSAGE closed form, K / Q projections."""
    B, H, D = hist.shape
    ne = ue.news_encoder
    if ue.use_candidate_aware_attn:
        hist, _ = candidate_aware(ue.candidate_aware_attn, hist, _topic(ne, user_category, user_subCategory),
                                  _topic(ne, category, subCategory), user_history_mask)
    nodes = ue.dropout_(ue.user_node_embedding.unsqueeze(0).expand(B, -1, -1))                                    # :121
    n_src = B if n_src is None else n_src
    if n_src > H + nodes.shape[1]:
        raise IndexError('rows per forward (%d) exceed the node slots H + config.batch_size (SURVEY Q7)' % n_src)
    X = torch.cat([hist, nodes], dim=1)
    conv = ue.graph_sage.convs[0]
    g = linear(X[:, :n_src].mean(dim=1), conv.lin_l).unsqueeze(1) + linear(hist, conv.lin_r)                      # :151-157
    return g, linear(g, ue.K), linear(cand, ue.Q)                                                                 # :161-162


def user_logits(ue, weighting, hist, cand, category, subCategory, user_category, user_subCategory, user_history_mask,
                remaining_lifetime, n_src=None):
    """userEncoders.CROWN.forward after the history is encoded (userEncoders.py:103-105, :114-169) and the lifetime-weighted dot
    product of model.py:181 / util.py:23-49 -> logits [B, N]."""
    B, H, D = hist.shape
    N = cand.shape[1]
    g, kp, qp = _interest_inputs(ue, hist, cand, category, subCategory, user_category, user_subCategory, user_history_mask, n_src)
    A = kp.shape[-1]
    w = weighting
    return _InterestMatch.apply(kp.reshape(B * H, A), qp.reshape(B * N, A), g.reshape(B * H, D), cand.reshape(B * N, D),
                                remaining_lifetime, (B, N, H, A, D), 1.0 / ue.attention_scalar, float(w.alpha), float(w.beta),
                                bool(w.use_remaining_lifetime_weighting), bool(w.use_expired_penalty))               # :163-168


def user_representation(ue, hist, cand, category, subCategory, user_category, user_subCategory, user_history_mask, n_src=None):
    """``user_encoder(...)`` called on its own in training mode (userEncoders.py:101-175) -> user_representation [B, N, D] with
    autograd.  The model's own forward never materialises this tensor (``user_logits`` fuses the attention with the dot product);
    the last two steps (:163-168: softmax over the history, weighted sum) are two batched torch matmuls here."""
    g, kp, qp = _interest_inputs(ue, hist, cand, category, subCategory, user_category, user_subCategory, user_history_mask, n_src)
    a = torch.softmax(torch.bmm(qp, kp.transpose(1, 2)) / ue.attention_scalar, dim=2)                             # [B, N, H]
    return torch.bmm(a, g)


def lifetime_weighted_logits(w, user, news, remaining_lifetime):
    """RemainingLifetimeWeighting.forward (util.py:23-49) as torch ops with autograd (the module called on its own)."""
    base = (user * news).sum(dim=-1)
    if not w.use_remaining_lifetime_weighting:
        return base
    r = remaining_lifetime.float()
    weight = torch.sigmoid(w.alpha * r)
    if w.use_expired_penalty:
        weight = torch.where(r < 0, w.beta * weight, weight)
    return base * weight


def content_flat(enc, title_text, title_mask, content_text, category, subCategory):
    """The base content encoder (CROWN: newsEncoders.py:302-373, MHSA: :582-595) on M flat news with autograd -> [M, dim]."""
    from .newsEncoders import CROWN, MHSA
    if getattr(enc, 'compute_dtype', 'fp32') != 'fp32':
        raise NotImplementedError("compute_dtype %r is a scoring option (BASELINE config 3); the training step is fp32: build the "
                                  "model with compute_dtype='fp32' to train" % enc.compute_dtype)
    if isinstance(enc, CROWN):
        title_p, body_p = pooled_tokens(enc, title_text, content_text)
        return crown_tail(enc, title_p, body_p, category, subCategory)
    if isinstance(enc, MHSA):
        return mhsa_content(enc, title_text, title_mask, category, subCategory)
    raise NotImplementedError('the training path covers the CROWN and MHSA content encoders')


def news_flat(ne, title_text, title_mask, content_text, category, subCategory, freshness, lifetime):
    """LIME.forward (newsEncoders.py:140-161) on M flat news with autograd -> [M, 400]."""
    content = content_flat(ne.base_news_encoder, title_text, title_mask, content_text, category, subCategory)
    return lime_tail(ne, content, freshness, lifetime)


def forward_train(model, user_category, user_subCategory, user_title_text, user_title_mask, user_content_text, user_freshness,
                  user_user_topic_lifetime, user_history_mask, news_category, news_subCategory, news_title_text, news_title_mask,
                  news_content_text, news_freshness, news_user_topic_lifetime, remaining_lifetime):
    """Model.forward with [B, N] candidates (model.py:171-187), recording the autograd graph.  Candidates and history go
    through the news encoder as one flat batch of B (N + H) news.  After the token encoders come the content tail, freshness
    and project per news, then the CROWN user encoder and the lifetime-weighted dot product: 3 % of the FLOPs in some 300
    short launches forward + backward (4 ms of kernel time at config 2b -- short kernels, not idle gaps: capturing them into
    HIP graphs with torch.cuda.make_graphed_callables was measured and changed nothing)."""
    ne = model.news_encoder
    B, N = news_category.shape
    H = user_category.shape[1]
    i32 = lambda t: t if t.dtype == torch.int32 else t.to(torch.int32)
    flat2 = lambda c, u: torch.cat([c.reshape(B * N, -1), u.reshape(B * H, -1)], dim=0)
    flat1 = lambda c, u: torch.cat([c.reshape(-1), u.reshape(-1)], dim=0)
    if news_freshness.dim() == 1:
        news_freshness = news_freshness.unsqueeze(1).expand(B, N)
    if news_user_topic_lifetime.dim() == 1:
        news_user_topic_lifetime = news_user_topic_lifetime.unsqueeze(1).expand(B, N)
    category, subCategory = i32(flat1(news_category, user_category)), i32(flat1(news_subCategory, user_subCategory))
    rep = news_flat(ne, i32(flat2(news_title_text, user_title_text)), flat2(news_title_mask, user_title_mask),
                    i32(flat2(news_content_text, user_content_text)), category, subCategory,
                    flat1(news_freshness.float(), user_freshness.float()).contiguous(),
                    flat1(news_user_topic_lifetime.float(), user_user_topic_lifetime.float()).contiguous())
    cand = rep[:B * N].view(B, N, -1)
    hist = rep[B * N:].view(B, H, -1)
    return user_logits(model.user_encoder, model.remaining_lifetime_weighting, hist, cand, i32(news_category).contiguous(),
                       i32(news_subCategory).contiguous(), i32(user_category).contiguous(), i32(user_subCategory).contiguous(),
                       user_history_mask.contiguous(), remaining_lifetime.float().contiguous())


# ---------------------------------------------------------------------------------------------------------------------
# the native step: flat parameter / gradient / Adam-state buffers, one clip + one Adam launch, one all-reduce
# ---------------------------------------------------------------------------------------------------------------------
def negative_log_softmax(logits):
    """trainer.py:71-73 on the HIP loss kernel (forward and gradient in one launch)."""
    return _NllSoftmax.apply(logits)


class _NllSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        loss, d = ops.nll_softmax(logits.contiguous())
        ctx.save_for_backward(d)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (d,) = ctx.saved_tensors
        return d * dloss


# parameters the reference constructs and never uses on this path: autograd leaves their .grad at None, Adam and
# clip_grad_norm_ skip them, and so does the flat bucket
_DEAD = ('base_news_encoder.affine.', '.ISAB.', 'category_predictor.', 'user_encoder.affine.', 'candidate_aware_attn.value_proj.')


class TrainStep:
    """trainer.py:33 + :143-148 for one process (one GPU): Adam(lr, weight_decay) over the parameters that receive
    gradients, ``clip_grad_norm_(gradient_clip_norm)``, and -- under torch.distributed -- ONE all-reduce (RCCL over xGMI on
    ROCm) of the flat gradient bucket before the clip, averaging over ranks as DistributedDataParallel does.

    Parameters are re-pointed into one flat fp32 buffer (``p.data`` becomes a view, values preserved) and their ``.grad``
    into a second one, so zeroing, the norm, the all-reduce and the Adam update are one launch each.

    One difference from torch's Adam, visible only with ``weight_decay > 0`` (the reference trains with 0, config.py:47): every
    bucket parameter is decayed every step, including one whose gradient happened to be all zero in that step; torch skips a
    parameter whose ``.grad`` is None (the never-used parameters of SURVEY Q20 are outside the bucket and skipped here too).
    """

    def __init__(self, model, lr=1e-4, weight_decay=0.0, gradient_clip_norm=4.0, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None):
        self.model = model
        self.lr, self.weight_decay, self.clip, self.betas, self.eps = lr, weight_decay, gradient_clip_norm, betas, eps
        self.group = process_group
        named, seen = [], set()
        for name, p in model.named_parameters():
            if not p.requires_grad or id(p) in seen or any(d in name for d in _DEAD):
                continue
            seen.add(id(p))
            named.append((name, p))
        self.names = [n for n, _ in named]
        self._params = named
        # every parameter starts on a 256-byte boundary of the bucket (the LDS-DMA GEMM wants 16-byte aligned operands);
        # the padding stays zero in all four buffers
        align = lambda n: (n + 63) // 64 * 64
        total = sum(align(p.numel()) for _, p in named)
        dev = named[0][1].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for _, p in named:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += align(n)
        distributed.broadcast_(self.flat, 0, process_group)      # replicas start from rank 0's parameters (as under DDP)
        if hasattr(model, '_graphs'):
            model._graphs = {}                  # captured scoring graphs hold the parameters' old addresses
        self.step_count = 0
        self.last_norm = None

    def backward(self, loss):
        """``loss.backward()`` with the gradients landing in the flat bucket.  The parameters' ``.grad`` views are taken away for
        the duration of the backward pass, so autograd hands every parameter its gradient tensor as it is (no accumulation
        kernel: with the views in place AccumulateGrad runs one ``grad += g`` launch per parameter, 72 a step, and the bucket has
        to be zeroed first); the tensors are then gathered into the bucket with one ``lime_multi_copy`` launch per 32
        parameters and the views are put back.  A parameter that received no gradient in this step gets zeros."""
        self._check_bucket()
        views = [p.grad for _, p in self._params]
        for _, p in self._params:
            p.grad = None
        try:
            loss.backward()
            pairs = []
            for (_, p), v in zip(self._params, views):
                if p.grad is None:
                    v.zero_()
                else:
                    pairs.append((v, p.grad))
            keep = ops.multi_copy(pairs)
        finally:
            for (_, p), v in zip(self._params, views):
                p.grad = v
        del keep

    def backward_and_update(self, loss):
        """loss.backward() into the flat bucket, all-reduce, clip, Adam.  Returns the (device) gradient norm."""
        self.backward(loss)
        distributed.allreduce_mean_(self.grad, self.group)
        return self.update()

    def _check_bucket(self):
        """Parameters and their .grad must still be the views into the flat buckets this object set up: ``model.zero_grad()``
        (set_to_none), ``model.to(...)`` or a foreign optimizer would silently detach them."""
        pb, gb = self.flat.data_ptr(), self.grad.data_ptr()
        for name, p in self._params:
            if p.grad is None or p.grad.data_ptr() - gb != p.data_ptr() - pb or not (0 <= p.data_ptr() - pb < self.flat.numel() * 4):
                raise RuntimeError('TrainStep: parameter %s (or its .grad) no longer lives in the flat bucket -- do not call '
                                   'model.zero_grad() / model.to() after constructing TrainStep (it zeroes the bucket itself)' % name)

    def update(self):
        self.step_count += 1
        coef = ops.grad_clip_coef(self.grad, self.clip)
        ops.adam_step_(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas, self.eps,
                       self.weight_decay, grad_scale=coef[1:])
        self.last_norm = coef[:1]
        return self.last_norm

    # ---- checkpoint / resume ------------------------------------------------------------------------------------------
    def state_dict(self):
        """Optimizer state for resuming: Adam's moments per parameter NAME (so it survives a different bucket layout), the step
        count and the hyper-parameters.  The model itself is saved the reference's way: {model_name: model.state_dict()}
        (trainer.py:220), see ``save_checkpoint``."""
        out = {'step': self.step_count, 'lr': self.lr, 'weight_decay': self.weight_decay, 'betas': tuple(self.betas), 'eps': self.eps,
               'gradient_clip_norm': self.clip, 'exp_avg': {}, 'exp_avg_sq': {}}
        for name, (off, n, shape) in self._slots().items():
            out['exp_avg'][name] = self.exp_avg[off:off + n].view(shape).clone()
            out['exp_avg_sq'][name] = self.exp_avg_sq[off:off + n].view(shape).clone()
        return out

    def load_state_dict(self, state):
        slots = self._slots()
        missing = sorted(set(slots) - set(state['exp_avg']))
        if missing:
            raise KeyError('optimizer state lacks %d parameters, e.g. %s' % (len(missing), missing[:3]))
        with torch.no_grad():
            for name, (off, n, shape) in slots.items():
                self.exp_avg[off:off + n].copy_(state['exp_avg'][name].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(state['exp_avg_sq'][name].reshape(-1))
        self.step_count = int(state['step'])
        self.lr, self.weight_decay, self.eps = state['lr'], state['weight_decay'], state['eps']
        self.betas, self.clip = tuple(state['betas']), state['gradient_clip_norm']

    def _slots(self):
        named = dict(self.model.named_parameters())
        base = self.flat.data_ptr()
        return {n: ((named[n].data_ptr() - base) // 4, named[n].numel(), tuple(named[n].shape)) for n in self.names}

    def step(self, *batch):
        """One training step on a batch in ``Model.forward``'s 26-tensor order; returns the (device) loss."""
        logits = self.model(*batch)
        loss = negative_log_softmax(logits)
        self.backward_and_update(loss)
        return loss.detach()


def save_checkpoint(path, model, train_step=None):
    """The reference's checkpoint file -- torch.save({model.model_name: model.state_dict()}) (trainer.py:220), which main.py:45,
    60 load with ``torch.load(path)[model.model_name]`` -- plus, under the extra key 'optimizer', the TrainStep state when
    one is given (the reference saves none: its runs cannot resume mid-training)."""
    payload = {model.model_name: {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if train_step is not None:
        st = train_step.state_dict()
        for key in ('exp_avg', 'exp_avg_sq'):
            st[key] = {k: v.cpu() for k, v in st[key].items()}
        payload['optimizer'] = st
    torch.save(payload, path)


def load_checkpoint(path, model, train_step=None, map_location='cpu'):
    """Load a checkpoint written by ``save_checkpoint`` or by the reference's trainer (tensors only: weights_only=True).
    Parameters keep their storage (``load_state_dict`` copies in place), so a TrainStep's flat bucket stays valid."""
    payload = torch.load(path, map_location=map_location, weights_only=True)
    if model.model_name not in payload:
        raise KeyError('checkpoint holds %s, not %r' % (sorted(payload), model.model_name))
    model.load_state_dict(payload[model.model_name])
    if train_step is not None and 'optimizer' in payload:
        train_step.load_state_dict(payload['optimizer'])
    return payload
