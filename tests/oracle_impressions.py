"""Checker for the scoring-only layout of BASELINE.json configs[4] (B impressions x K candidates, eval semantics): the oracle's
own stage functions composed so that every history and candidate is ENCODED ONCE, then the user side run per (impression,
candidate) row with N = 1 -- which is what ``oracle.model_forward(..., eval_shape=True)`` computes on the B * K expanded rows
(the reference's layout, util.py:86-111), because no op couples rows except the GraphSAGE source count ``n_src``.
tests/test_oracle_golden.py ties this composition to ``model_forward`` on expanded rows; the GPU tests use it at sizes where
re-encoding the history per candidate on the CPU would take minutes.  Test infrastructure only."""
import torch

from oracle import lime_oracle as O


def score_impressions(sd, cfg, batch, n_src, rows_per_pass=800):
    b = batch
    B, K = b['news_category'].shape
    H = b['user_category'].shape[1]
    ne, ue = 'news_encoder.', 'user_encoder.'
    fr = b['news_freshness'] if b['news_freshness'].dim() == 2 else b['news_freshness'].unsqueeze(1).expand(B, K)
    lt = b['news_user_topic_lifetime'] if b['news_user_topic_lifetime'].dim() == 2 else b['news_user_topic_lifetime'].unsqueeze(1).expand(B, K)
    with torch.no_grad():
        cand = O.lime_news_encoder(sd, ne, cfg, b['news_title_text'], b['news_title_mask'], b['news_content_text'], b['news_category'],
                                   b['news_subCategory'], fr, lt)                                  # [B, K, D]
        hist = O.lime_news_encoder(sd, ne, cfg, b['user_title_text'], b['user_title_mask'], b['user_content_text'], b['user_category'],
                                   b['user_subCategory'], b['user_freshness'], b['user_user_topic_lifetime'])   # [B, H, D]
        hist_topic = O.topic_representation(sd, ne, b['user_category'], b['user_subCategory'])       # [B, H, 50]
        out = torch.empty(B, K)
        per = max(1, rows_per_pass // K)
        for b0 in range(0, B, per):
            b1 = min(B, b0 + per)
            n = (b1 - b0) * K
            rep = lambda t: t[b0:b1].repeat_interleave(K, dim=0)
            c = cand[b0:b1].reshape(n, 1, -1)
            ct = O.topic_representation(sd, ne, b['news_category'][b0:b1].reshape(n, 1), b['news_subCategory'][b0:b1].reshape(n, 1))
            h = rep(hist)
            if cfg.use_candidate_ware_clicked_news_attention:
                h, _ = O.candidate_aware_attention(sd, ue + 'candidate_aware_attn.', h, rep(hist_topic), ct, rep(b['user_history_mask']),
                                                   residual=cfg.use_residual_connection)
            g = O.graph_sage(sd, ue + 'graph_sage.', h, sd[ue + 'user_node_embedding'], n_src=n_src)
            user = O.kq_attention(sd, ue, g, c, cfg.attention_dim)
            out[b0:b1] = O.remaining_lifetime_weighting(cfg, user, c, b['remaining_lifetime'][b0:b1].reshape(n, 1).float()).view(b1 - b0, K)
    return out
