"""ctypes binding of liblime_hip.so (the C ABI declared in include/lime_hip.h).

There is deliberately no fallback: if the shared library is missing or does not export a symbol,
loading raises -- the product path never routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblime_hip.so')

ABI_VERSION = 7          # LIME_ABI_VERSION of include/lime_hip.h this binding was written against
LIME_ACT = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'relu_grad': 4}


class LinearArgs(Structure):
    """Mirror of ``lime_linear_args`` (include/lime_hip.h)."""
    _fields_ = [
        ('a', c_void_p), ('lda', c_int64),
        ('a_ids', c_void_p), ('a_pe', c_void_p), ('lda_pe', c_int64), ('a_period', c_int32),
        ('w', c_void_p), ('ldw', c_int64),
        ('bias', c_void_p),
        ('res', c_void_p), ('ldr', c_int64), ('res_div', c_int32),
        ('res_ids', c_void_p), ('res_pe', c_void_p), ('ldr_pe', c_int64), ('res_period', c_int32),
        ('ln_gamma', c_void_p), ('ln_beta', c_void_p), ('ln_eps', c_float),
        ('c', c_void_p), ('ldc', c_int64),
        ('M', c_int32), ('N', c_int32), ('K', c_int32),
        ('act', c_int32),
        ('res_mod', c_int32), ('pool32', c_int32),
        ('ln_rstd', c_void_p),
        ('m_dev', c_void_p), ('c_ids', c_void_p),
        ('act_scale', c_float),
        ('dropout_p', c_float), ('dropout_seed', c_uint64), ('dropout_site', c_uint32),
    ]


class LinearBf16Args(ctypes.Structure):
    """lime_linear_bf16_args of include/lime_hip.h (same field order)."""
    _fields_ = [
        ('a', c_void_p), ('lda', c_int64),
        ('a_ids', c_void_p),
        ('w', c_void_p), ('ldw', c_int64),
        ('bias', c_void_p),
        ('res', c_void_p), ('ldr', c_int64), ('res_kind', c_int32), ('res_mod', c_int32),
        ('res_ids', c_void_p), ('res_pe', c_void_p), ('ldr_pe', c_int64), ('res_period', c_int32),
        ('ln_gamma', c_void_p), ('ln_beta', c_void_p), ('ln_eps', c_float), ('ln_count', c_int32),
        ('c', c_void_p), ('ldc', c_int64),
        ('M', c_int32), ('N', c_int32), ('K', c_int32),
        ('act', c_int32),
        ('pool32', c_int32), ('reserved', c_int32),
        ('m_dev', c_void_p), ('c_ids', c_void_p),
    ]


class FfnBf16Args(ctypes.Structure):
    """lime_ffn_bf16_args of include/lime_hip.h (same field order)."""
    _fields_ = [
        ('x', c_void_p), ('ldx', c_int64),
        ('w1p', c_void_p),
        ('w2p', c_void_p),
        ('b2', c_void_p),
        ('ln_gamma', c_void_p), ('ln_beta', c_void_p), ('ln_eps', c_float),
        ('pool32', c_int32),
        ('out', c_void_p), ('ldo', c_int64),
        ('M', c_int32), ('E', c_int32), ('F', c_int32),
        ('reserved', c_int32),
        ('m_dev', c_void_p),
    ]


class EncoderBlockBf16Args(ctypes.Structure):
    """lime_encoder_block_bf16_args of include/lime_hip.h (same field order)."""
    _fields_ = [
        ('attn', c_void_p), ('lda', c_int64),
        ('w0p', c_void_p),
        ('add_rows', c_void_p), ('ld_add', c_int64), ('add_period', c_int32),
        ('res_kind', c_int32),
        ('res', c_void_p), ('ldr', c_int64), ('res_rows', c_int64),
        ('res_ids', c_void_p),
        ('ln1_gamma', c_void_p), ('ln1_beta', c_void_p), ('ln1_eps', c_float),
        ('pool32', c_int32),
        ('w1p', c_void_p), ('w2p', c_void_p), ('b2', c_void_p),
        ('ln2_gamma', c_void_p), ('ln2_beta', c_void_p), ('ln2_eps', c_float),
        ('M', c_int32), ('E', c_int32), ('F', c_int32),
        ('out', c_void_p), ('ldo', c_int64),
        ('m_dev', c_void_p),
    ]


class InprojBf16Args(ctypes.Structure):
    """lime_inproj_bf16_args of include/lime_hip.h (same field order)."""
    _fields_ = [
        ('a', c_void_p), ('lda', c_int64), ('a_rows', c_int64), ('a_ids', c_void_p),
        ('wp', c_void_p),
        ('add_rows', c_void_p), ('ld_add', c_int64), ('add_period', c_int32),
        ('M', c_int32), ('N', c_int32), ('K', c_int32),
        ('reserved', c_int32),
        ('c_ids', c_void_p),
        ('out', c_void_p), ('ldo', c_int64), ('out_rows', c_int64),
        ('m_dev', c_void_p),
    ]


class CopyDesc(ctypes.Structure):
    """lime_copy_desc of include/lime_hip.h."""
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('bytes', c_int64)]


MAX_COPIES = 32


class GatherDesc(ctypes.Structure):
    """lime_gather_desc of include/lime_hip.h."""
    _fields_ = [('table', c_void_p), ('table_stride', c_int64), ('out', c_void_p), ('out_stride', c_int64),
                ('row_bytes', c_int32), ('reserved', c_int32)]


MAX_GATHERS = 16

# name -> (restype, argtypes); every symbol include/lime_hip.h declares
SIGNATURES = {
    'lime_abi_version': (c_int32, []),
    'lime_last_error_string': (c_char_p, []),
    'lime_last_linear_kernel': (c_char_p, []),
    'lime_set_split_gemm': (c_int32, [c_int32]),
    'lime_linear_f32': (c_int32, [POINTER(LinearArgs), c_void_p]),
    'lime_linear_group_f32': (c_int32, [POINTER(LinearArgs), c_int32, c_void_p]),
    'lime_embed_pe_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int64,
                                    c_int32, c_void_p]),
    'lime_token_attention_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                           c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    'lime_token_attention_count_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                 c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    'lime_token_attention_rows_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                c_int32, c_int32, c_int32, c_float, c_void_p]),
    'lime_token_attention_rows_bf16': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                 c_int32, c_int32, c_int32, c_float, c_int32, c_void_p]),
    'lime_compact_sequences': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    'lime_compact_sequences_workspace': (c_int64, [c_int32]),
    'lime_pad_heads_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'lime_mean_pool_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    'lime_mean_pool_count_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'lime_bucketize_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    'lime_mhsa_live_ids': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'lime_mhsa_compact_mask': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'lime_fuse_rows_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'lime_bucketize_cuts_f32': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    'lime_topic_rep_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                     c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    'lime_intent_fuse_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
                                       c_int32, c_void_p]),
    'lime_additive_pool_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p,
                                         c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    'lime_additive_pool_count_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                               c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    'lime_cand_attn_weights_workspace': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'lime_cand_attn_weights_ws_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                                c_int64, c_void_p]),
    'lime_cand_attn_weights_shared_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                    c_int32, c_void_p, c_int64, c_void_p]),
    'lime_cand_attn_weights_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                             c_int32, c_void_p]),
    'lime_gate_ln_sage_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'lime_gate_ln_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64,
                                   c_int32, c_void_p]),
    'lime_sage_mean_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'lime_interest_match_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                          c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_int32, c_int32,
                                          c_void_p]),
    'lime_lifetime_score_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float,
                                          c_int32, c_int32, c_void_p]),
    'lime_row_scale_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    'lime_gather_rows_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'lime_multi_copy': (c_int32, [ctypes.POINTER(CopyDesc), c_int32, c_void_p]),
    'lime_gather_rows_multi': (c_int32, [c_void_p, c_int64, ctypes.POINTER(GatherDesc), c_int32, c_void_p]),
    'lime_linear_bf16': (c_int32, [ctypes.POINTER(LinearBf16Args), c_void_p]),
    'lime_token_attention_bf16': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                            c_int32, c_float, c_int32, c_void_p]),
    'lime_encoder_ffn_bf16': (c_int32, [ctypes.POINTER(FfnBf16Args), c_void_p]),
    'lime_ffn_pack_bf16': (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'lime_encoder_block_bf16': (c_int32, [ctypes.POINTER(EncoderBlockBf16Args), c_void_p]),
    'lime_oproj_pack_bf16_size': (c_int64, []),
    'lime_oproj_pack_bf16': (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    'lime_inproj_bf16': (c_int32, [ctypes.POINTER(InprojBf16Args), c_void_p]),
    'lime_inproj_pack_bf16_size': (c_int64, [c_int32]),
    'lime_inproj_pack_bf16': (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    'lime_ffn_bf16_model_columns': (c_int32, []),
    'lime_ffn_pack_bf16_size': (c_int64, [c_int32, c_int32]),
    'lime_to_bf16': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'lime_mean_pool_bf16': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    # training step
    'lime_linear_wgrad_workspace': (c_int64, [c_int32, c_int32, c_int32]),
    'lime_linear_wgrad_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32,
                                        c_int32, c_void_p, c_int64, c_void_p]),
    'lime_colsum_workspace': (c_int64, [c_int32, c_int32]),
    'lime_colsum_f32': (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    'lime_layernorm_bwd_workspace': (c_int64, [c_int32, c_int32]),
    'lime_layernorm_bwd_f32': (c_int32, [c_void_p, c_int64, c_int32, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                         c_int64, c_void_p]),
    'lime_layernorm_bwd_dropout_f32': (c_int32, [c_void_p, c_int64, c_int32, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                                 c_int64, c_void_p, c_int64, c_float, c_uint64, c_uint32, c_void_p]),
    'lime_dropout2_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_float, c_uint64, c_uint32, c_uint32, c_void_p]),
    'lime_relu_bwd_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_float, c_void_p]),
    'lime_token_attention_bwd_workspace': (c_int64, [c_int32, c_int32, c_int32]),
    'lime_token_attention_stats_workspace': (c_int64, [c_int32, c_int32, c_int32]),
    'lime_token_attention_bwd_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                               c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_float,
                                               c_void_p, c_int64, c_float, c_uint64, c_uint32, c_void_p, c_void_p]),
    'lime_fill_pad_rows_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p]),
    'lime_additive_pool_bwd_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64,
                                             c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p]),
    'lime_token_attention_lse_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                               c_int32, c_int32, c_int32, c_float, c_void_p]),
    'lime_token_attention_bwd_lse_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                                   c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                   c_float, c_void_p, c_int64, c_void_p]),
    'lime_dropout_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_float, c_uint64, c_uint32, c_void_p]),
    'lime_embed_pe_dropout_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int32,
                                            c_float, c_uint64, c_uint32, c_uint32, c_void_p]),
    'lime_dropout_add_layernorm_f32': (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64,
                                                 c_void_p, c_int64, c_int32, c_float, c_uint64, c_uint32, c_void_p]),
    'lime_token_attention_dropout_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                                   c_int32, c_int32, c_float, c_float, c_uint64, c_uint32, c_void_p, c_int64, c_void_p]),
    'lime_embed_bwd_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p]),
    'lime_embed_bwd_sorted_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64,
                                            c_void_p]),
    'lime_embed_bwd_sorted_workspace': (c_int64, [c_int64, c_int32]),
    'lime_embed_bwd_small_f32': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p]),
    'lime_grad_clip_coef_f32': (c_int32, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    'lime_adam_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                c_int32, c_void_p, c_void_p]),
    'lime_intent_fuse_bwd_workspace': (c_int64, [c_int64, c_int32]),
    'lime_intent_fuse_bwd_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    'lime_gate_ln_bwd_workspace': (c_int64, [c_int64, c_int32]),
    'lime_gate_ln_bwd_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    'lime_interest_match_bwd_workspace': (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    'lime_interest_match_bwd_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_int32,
                                              c_int32, c_void_p, c_int64, c_void_p]),
    'lime_cand_attn_weights_train_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                   c_float, c_uint64, c_uint32, c_void_p]),
    'lime_cand_attn_weights_bwd_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                                 c_int32, c_int32, c_float, c_uint64, c_uint32, c_void_p]),
    'lime_nll_softmax_f32': (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None


class LimeHipError(RuntimeError):
    pass


def load():
    """Load liblime_hip.so once; raises if it is missing (build it with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LimeHipError('%s is missing: the HIP extension has not been built (run `python -c "import '
                           '__graft_entry__ as g; g.build()"` at the repo root); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.lime_abi_version()
    if got != ABI_VERSION:
        raise LimeHipError('liblime_hip.so has ABI version %d, this binding expects %d' % (got, ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().lime_last_error_string()
        raise LimeHipError('%s failed with status %d: %s' % (what, status, msg.decode() if msg else ''))
