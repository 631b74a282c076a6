// Internal interface between lime_linear_f32's dispatcher (gemm_f32.hip) and the two-workgroups-per-CU LDS-DMA
// kernel (gemm_pp_f32.hip).
#pragma once
#include "common.h"
#include "dropout.h"

#define LIME_PP_NOT_APPLICABLE 1

struct PPParams {
    const float* a; long lda; const int* a_ids;
    const float* w; long ldw; const float* bias;
    const float* res; long ldr; int res_mod; const int* res_ids; const float* res_pe; long ldr_pe; int res_period;
    const float* ln_g; const float* ln_b; float ln_eps; float* ln_rstd;
    float* c; long ldc; int M, N, K;
    int ln_count;        // LayerNorm divides by this many columns (N unless the caller zero-padded N)
    int n_row_blocks, n_col_blocks;
    const int* m_dev;    // optional device-side row count: the kernel runs min(*m_dev, M) rows (M is the capacity)
    const int* c_ids;    // optional (CID instantiations): output row of A row r is c_ids[r]; a periodic residual is indexed by it
    int act = 0;         // gemm_sp_kernel only, instantiations without the ReLU template flag: LIME_ACT_TANH / LIME_ACT_SIGMOID at run time
    int res_div = 1;     // gemm_sp_kernel only (RES == 1 without res_mod): residual row = r / res_div (one row broadcast to res_div rows)
    float act_scale = 1.f;   // gemm_sp_kernel, RES == 3 (LIME_ACT_RELU_GRAD): v = res > 0 ? v * act_scale : 0
    LimeDropout drop = {0, 0, 1.f};   // gemm_sp_kernel, ReLU instantiations without residual: thresh != 0 -> the dropout mask behind the ReLU
#ifdef LIME_STAMPS
    unsigned long long* stamps;
#endif
};

int lime_linear_pp(const lime_linear_args* a, hipStream_t stream);
int lime_linear_sp(const lime_linear_args* a, hipStream_t stream);       // gemm_sp_f32.hip (split product on the bf16 cores), same convention
int lime_linear_mid(const lime_linear_args* a, hipStream_t stream);      // gemm_mid_f32.hip, same return convention

// wgrad_sp_f32.hip: the weight gradient on the split product (lime_linear_wgrad_f32's big-M path, backward_f32.hip dispatches)
struct LimeWgradSpPlan { bool swap; int n_tiles, k_tiles, splits, rows_per_split; long np, kp; double fill; };
LimeWgradSpPlan lime_wgrad_sp_plan(int M, int N, int K);
int lime_wgrad_sp_launch(const LimeWgradSpPlan& w, const float* dy, long ldy, const float* x, long ldx, float* ws, int M, int N, int K,
                         int ones_col, hipStream_t s);
int lime_wgrad_sp_reduce_t(const LimeWgradSpPlan& w, const float* ws, float* dw, long lddw, int N, int K, int accumulate, hipStream_t s);
int lime_split_mode();                                                   // gemm_sp_f32.hip: the lime_set_split_gemm() setting
// token_attn_sp_f32.hip: unmasked S = 32 / 64 / 128 attention on the split product (LIME_PP_NOT_APPLICABLE: not taken)
int lime_token_attention_sp(const float* q, const float* k, const float* v, long ld, const int* row_map, const int* n_seq_dev,
                            float* out, long ldo, int n_seq, int S, int n_head, int hd, float scale, float* lse, hipStream_t s,
                            const LimeDropout* drop = nullptr);
// token_attn_bwd_sp_f32.hip: the one-pass attention backward (64 < S <= 128, no key mask) with all its products on the split product
int lime_token_attention_bwd_sp(const float* q, const float* k, const float* v, long ld, const float* dout, long ldo, float* dq, float* dk,
                                float* dv, long ldd, int n_seq, int S, int n_head, int head_dim, int head_stride, float scale,
                                const LimeDropout& drop, hipStream_t s);
