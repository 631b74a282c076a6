"""Definitions of the golden-vector cases shared by tools/make_goldens.py (which runs the imported
reference on them in the build container) and by the tests (which run the oracle / the HIP path on
the very same regenerated inputs and weights).  Only outputs are stored in tests/golden/*.npz.
"""
import numpy as np
import torch

from lime_cikm25_amd.config import make_config
from lime_cikm25_amd import synth

# oracle.BUCKET_THRESHOLD_BITS restated here on purpose: the golden inputs must not depend on the
# thing they are pinning
_CUTS = (0x45326B18, 0x4AF8B232, 0x50AD53E8, 0x567199BD, 0x5C2861F4, 0x61EAB505, 0x67A39429, 0x6D6402D2, 0x731EE960)


def _edit_none(cfg, batch):
    return batch


def _edit_empty_history(cfg, batch):
    """Rows 0 and 3: no clicked news at all (mask all False, every slot the pad news)."""
    for r in (0, 3):
        batch['user_history_mask'][r] = False
        for k in ('user_category', 'user_subCategory', 'user_title_text', 'user_content_text'):
            batch[k][r] = 0
        batch['user_title_mask'][r] = False
        batch['user_title_mask'][r, :, 0] = True
        batch['user_content_mask'][r] = False
        batch['user_content_mask'][r, :, 0] = True
        batch['user_freshness'][r] = 0.0
        batch['user_user_topic_lifetime'][r] = 0.0
    # row 1: completely full history
    batch['user_history_mask'][1] = True
    return batch


def _edit_bucket_edges(cfg, batch):
    """Candidate freshness / lifetime placed on both sides of every fp32 bucket cut point
    (SURVEY Q11), plus 0, 1 and values below 1; remaining lifetime both saturated and not."""
    cuts = np.array(_CUTS, dtype=np.uint32)
    vals = np.concatenate([(cuts - 1).view(np.float32), cuts.view(np.float32), (cuts + 1).view(np.float32),
                           np.array([0.0, 0.5, 1.0, 2854.0, 86400.0], dtype=np.float32)])
    B, N = batch['news_freshness'].shape
    assert B * N >= vals.size
    f = batch['news_freshness'].reshape(-1).clone()
    l = batch['news_user_topic_lifetime'].reshape(-1).clone()
    f[:vals.size] = torch.from_numpy(vals)
    l[:vals.size] = torch.from_numpy(vals[::-1].copy())
    batch['news_freshness'] = f.view(B, N)
    batch['news_user_topic_lifetime'] = l.view(B, N)
    # the same values feed the history side of rows 0..
    H = batch['user_freshness'].shape[1]
    uf = batch['user_freshness'].reshape(-1).clone()
    ul = batch['user_user_topic_lifetime'].reshape(-1).clone()
    n = min(vals.size, uf.numel())
    uf[:n] = torch.from_numpy(vals[:n])
    ul[:n] = torch.from_numpy(vals[::-1].copy()[:n])
    batch['user_freshness'] = uf.view(-1, H)
    batch['user_user_topic_lifetime'] = ul.view(-1, H)
    # remaining lifetime: mix of huge (saturated), tiny positive / negative and exact zero
    r = (batch['news_user_topic_lifetime'] - batch['news_freshness']).reshape(-1).clone()
    small = torch.tensor([0.0, -0.0, 0.5, -0.5, 3.0, -3.0, 40.0, -40.0, 120.0, -120.0, 400.0, -400.0])
    r[-small.numel():] = small
    batch['remaining_lifetime'] = r.view(B, N)
    return batch


EDITS = {
    'none': _edit_none,
    'empty_history': _edit_empty_history,
    'bucket_edges': _edit_bucket_edges,
}

_SMALL = dict(vocabulary_size=5000, category_num=18, subCategory_num=270)

# name -> dict(config overrides, B, N, seed, eval_shape, edit)
CASES = {
    # BASELINE.json configs[0]: batch 8, history 10, title 16, K = 1+1 (body 32)
    'cfg1_crown': dict(cfg=dict(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8, **_SMALL),
                       B=8, N=2, seed=11, eval_shape=False, edit='none'),
    'cfg1_mhsa': dict(cfg=dict(content_encoder='MHSA', max_history_num=10, max_title_length=16, max_abstract_length=32,
                               batch_size=8, **_SMALL),
                      B=8, N=2, seed=12, eval_shape=False, edit='none'),
    # the reference's eval path: one candidate per row, no N axis on the inputs (model.py:158-169)
    'cfg1_crown_eval': dict(cfg=dict(max_history_num=10, max_title_length=16, max_abstract_length=32, batch_size=8, **_SMALL),
                            B=8, N=1, seed=13, eval_shape=True, edit='none'),
    # rows per forward (6) > history slots (4): the GraphSAGE mean spills into user-node slots
    'spill': dict(cfg=dict(max_history_num=4, max_title_length=8, max_abstract_length=16, batch_size=6, **_SMALL),
                  B=6, N=3, seed=14, eval_shape=False, edit='none'),
    # rows per forward (3) < config.batch_size (8): user_node_embedding is larger than the batch
    'short_batch': dict(cfg=dict(max_history_num=6, max_title_length=8, max_abstract_length=16, batch_size=8, **_SMALL),
                        B=3, N=5, seed=15, eval_shape=False, edit='none'),
    'empty_history': dict(cfg=dict(max_history_num=6, max_title_length=8, max_abstract_length=16, batch_size=4, **_SMALL),
                          B=4, N=2, seed=16, eval_shape=False, edit='empty_history'),
    'bucket_edges': dict(cfg=dict(max_history_num=8, max_title_length=8, max_abstract_length=16, batch_size=8, **_SMALL),
                         B=8, N=4, seed=17, eval_shape=False, edit='bucket_edges'),
    # real sequence lengths (title 32, body 128) at a tiny batch: pins the S=32 / S=128 token encoder
    'full_len': dict(cfg=dict(max_history_num=3, max_title_length=32, max_abstract_length=128, batch_size=2, **_SMALL),
                     B=2, N=2, seed=18, eval_shape=False, edit='none'),
    # BASELINE configs[3] body length (Adressa shape, 512 tokens) at a tiny batch: the S = 512 token encoder and, in the
    # gradient goldens, the blocked attention backward
    'long_body': dict(cfg=dict(max_history_num=2, max_title_length=32, max_abstract_length=512, batch_size=2, **_SMALL),
                      B=2, N=2, seed=19, eval_shape=False, edit='none'),
    # the other two fusion methods of LIME (newsEncoders.py:154-159): no `project`, the news representation keeps the content
    # encoder's 900 columns and the whole user encoder runs 900 wide (10 heads x 90)
    'fusion_add': dict(cfg=dict(fusion_method='add', max_history_num=6, max_title_length=8, max_abstract_length=16, batch_size=4, **_SMALL),
                       B=4, N=3, seed=20, eval_shape=False, edit='none'),
    'fusion_gated': dict(cfg=dict(fusion_method='gated', max_history_num=6, max_title_length=8, max_abstract_length=16, batch_size=4, **_SMALL),
                         B=4, N=3, seed=21, eval_shape=False, edit='none'),
    # config.py:70 allows num_layers = 2: two post-LN encoder layers per token encoder (newsEncoders.py:244-247)
    'two_layers': dict(cfg=dict(num_layers=2, max_history_num=5, max_title_length=32, max_abstract_length=64, batch_size=3, **_SMALL),
                       B=3, N=2, seed=22, eval_shape=False, edit='none'),
}

WEIGHT_SEED = 7


def build_case(name):
    """-> (config, OrderedDict of the 26 inputs, case dict)."""
    c = CASES[name]
    cfg = make_config(**c['cfg'])
    batch = synth.make_batch(cfg, c['B'], c['N'], seed=c['seed'], eval_shape=c['eval_shape'])
    batch = EDITS[c['edit']](cfg, batch)
    return cfg, batch, c
