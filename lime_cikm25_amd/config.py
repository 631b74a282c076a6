"""Plain-namespace stand-in for the reference's ``Config`` attribute bag.

The reference's ``Config`` (config.py:20-301) is argparse + dataset directory
creation + a CUDA assert (config.py:211-212); the candidate-scoring path only
reads ~45 attributes from it.  ``make_config`` returns a
``SimpleNamespace`` carrying exactly those attribute names with the reference's
argparse defaults (config.py:24-107), so ``Model(config)`` here and
``Model(config)`` in the reference can be built from the same object.
"""
from types import SimpleNamespace

# attribute -> default, names and defaults from the reference's argparse block
_DEFAULTS = dict(
    news_encoder='LIME',                  # config.py:25
    user_encoder='CROWN',                 # config.py:26
    content_encoder='CROWN',              # config.py:27
    dataset='mind',                       # config.py:36
    tokenizer='MIND',                     # config.py:37
    word_threshold=3,                     # config.py:38
    max_title_length=32,                  # config.py:39
    max_abstract_length=128,              # config.py:40
    negative_sample_num=4,                # config.py:42
    max_history_num=50,                   # config.py:43
    batch_size=32,                        # config.py:45
    dropout_rate=0.0,                     # config.py:78 (0.2 there; scoring path runs eval-mode)
    fusion_method='concat',               # config.py:55
    freshness_embedding_dim=500,          # config.py:56
    lime_hidden_dim=200,                  # config.py:57
    lime_output_dim=400,                  # config.py:58
    num_buckets=10,                       # config.py:59
    use_candidate_ware_clicked_news_attention=True,   # config.py:60
    use_residual_connection=True,         # config.py:61
    lifetime_type='user_topic',           # config.py:62
    use_remaining_lifetime_weighting=True,  # config.py:63
    sigmoid_scaling_alpha=0.3,            # config.py:64
    penalty_scaling_beta=0.3,             # config.py:65
    use_expired_penalty=True,             # config.py:66
    fixed_lifetime=36 * 3600,             # config.py:67
    num_layers=1,                         # config.py:70
    feedforward_dim=512,                  # config.py:71
    head_num=10,                          # config.py:72
    head_dim=20,                          # config.py:73
    intent_embedding_dim=400,             # config.py:74
    intent_num=3,                         # config.py:75
    attention_dim=400,                    # config.py:77
    word_embedding_dim=300,               # config.py:79
    isab_num_inds=4,                      # config.py:80
    isab_num_heads=4,                     # config.py:81
    alpha=0.0,                            # config.py:82
    beta=0.0,                             # config.py:83
    category_embedding_dim=50,            # config.py:91
    subCategory_embedding_dim=50,         # config.py:92
    user_embedding_dim=50,                # config.py:90
    click_predictor='dot_product',        # config.py:109
    # training loop (trainer.py:17-69)
    seed=0,                               # config.py:33
    epoch=16,                             # config.py:44
    lr=1e-4,                              # config.py:46
    weight_decay=0.0,                     # config.py:47
    gradient_clip_norm=4.0,               # config.py:48
    dev_criterion='auc',                  # config.py:51
    early_stopping_epoch=5,               # config.py:52
    model_dir='models',                   # config.py:249-253 derive these four from the dataset and model name
    best_model_dir='best_model',
    dev_res_dir='dev/res',
    result_dir='results',
    # normally filled in by corpus.py:309-326
    vocabulary_size=50000,
    category_num=18,
    subCategory_num=270,
    user_num=1000,
    # not in the reference: arithmetic of the token encoders on the MI355X path.  'fp32' (exact-fp32 MFMA, the parity
    # configuration) or 'bf16' (BASELINE config 3: bf16 MFMA operands / activations, fp32 accumulate, softmax, LayerNorm)
    compute_dtype='fp32',
)


def make_config(**overrides):
    """Build the attribute bag; unknown names are rejected to catch typos."""
    unknown = set(overrides) - set(_DEFAULTS)
    if unknown:
        raise TypeError('unknown config attribute(s): %s' % sorted(unknown))
    d = dict(_DEFAULTS)
    d.update(overrides)
    return SimpleNamespace(**d)
