/*
 * lime_hip.h -- C ABI of liblime_hip.so: hand-written gfx950 (MI355X) kernels for LIME's
 * candidate-scoring path.
 *
 * The reference (seongeunryu/lime-cikm25) has no FFI layer: its boundary is the Python nn.Module
 * surface.  Every entry point below replaces a PyTorch op sequence of the
 * reference and cites it (file:line into the reference repository).  The drop-in nn.Modules in
 * lime_cikm25_amd/ bind these through ctypes; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (row-major, fp32 unless stated);
 *     the library never allocates, frees, copies to the host or synchronises;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns at once;
 *   - return value: LIME_OK (0) or a negative lime_status; lime_last_error_string() describes the
 *     last failure of the calling thread;
 *   - calls are stateless and re-entrant; every entry point of the scoring path is bitwise reproducible run to run (no
 *     atomics in any reduction).  Two training-step kernels add with float atomics and are reproducible only up to the
 *     order of those fp32 additions: lime_embed_bwd_f32 (word rows repeat across tokens; lime_embed_bwd_sorted_f32 is the atomic-free
 *     replacement the training step uses) and the S > 128 path of
 *     lime_token_attention_bwd_f32 (dq from the key blocks).
 */
#ifndef LIME_HIP_H
#define LIME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIME_ABI_VERSION 7

typedef enum {
    LIME_OK = 0,
    LIME_ERR_BAD_ARG = -1,      /* null pointer, non-positive dimension, misaligned leading dimension */
    LIME_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels are built for */
    LIME_ERR_LAUNCH = -3        /* hipGetLastError() after the launch was not hipSuccess */
} lime_status;

int lime_abi_version(void);
const char* lime_last_error_string(void);

/* activation applied after bias (+ residual) */
enum { LIME_ACT_NONE = 0, LIME_ACT_RELU = 1, LIME_ACT_TANH = 2, LIME_ACT_SIGMOID = 3,
       LIME_ACT_RELU_GRAD = 4 /* backward of a ReLU: v = res(r, n) > 0 ? v * act_scale : 0, res = the forward activation (no residual add) */ };

/*
 * lime_linear_f32: C = epilogue(A . W^T + bias), exact-fp32 MFMA: v_mfma_f32_16x16x4_f32 in the LDS-DMA kernel that takes the
 * large 16-byte-aligned problems (gemm_pp_kernel: the four GEMMs of an encoder layer), v_mfma_f32_32x32x2_f32 in the general
 * kernel behind every other shape (gemm_f32_kernel).
 *
 * Replaces every nn.Linear on the path, with the surrounding element-wise work fused:
 *   in_proj / out_proj / linear1 / linear2 of the two TransformerEncoderLayers (newsEncoders.py:244-247,316,320),
 *   intent_layers (newsEncoders.py:284-295), Attention.affine1 (layers.py:288), FreshnessEncoder.dense
 *   (newsEncoders.py:82), LIME.project (newsEncoders.py:152-153), gate_proj (layers.py:87), SAGEConv lin_l / lin_r (userEncoders.py:153), K / Q (userEncoders.py:161-162),
 *   MultiHeadAttention W_Q/W_K/W_V (layers.py:224-226).
 *
 * A operand, row r (0 <= r < M), K columns:
 *   a_ids == NULL : a[r * lda + k]
 *   a_ids != NULL : a[a_ids[r] * lda + k] (+ a_pe[(r % a_period) * lda_pe + k] when a_pe != NULL)
 *                   -- the word-embedding gather and the positional table of newsEncoders.py:311-312,827
 *                      fused into the operand fetch; `a` is then the [V, K] table.
 * W is [N, K] row-major with leading dimension ldw (the nn.Linear weight as stored).
 * Epilogue, in this order, per output element (r, n):
 *   v  = acc + bias[n]                                   (bias may be NULL)
 *   v  = act(v)                                          (LIME_ACT_*)
 *   v += residual(r, n)                                  (res may be NULL)
 *        res_ids == NULL : res[(r / res_div) * ldr + n]   (res_div >= 1; > 1 broadcasts one row to res_div rows);
 *                          with res_mod > 0 the row is ((r / res_div) % res_mod): a [res_mod, N] table repeated down
 *                          the rows -- in_proj of the encoder layers uses it for the positional term, which is linear:
 *                          (E[ids] + PE) W^T + b = E[ids] W^T + (PE W^T + b)[r % S]
 *        res_ids != NULL : res[res_ids[r] * ldr + n] (+ res_pe[(r % res_period) * ldr_pe + n])
 *   ln_gamma != NULL : v = LayerNorm over the N columns of row r (eps = ln_eps); requires N <= 320 and either no
 *        activation (residual allowed) or ReLU without residual
 *   c[r * ldc + n] = v                                   (pool32 == 0)
 *   pool32 == 1: c[(r / 32) * ldc + n] = mean over the 32 rows of block r / 32 of v -- the token mean pooling of
 *        newsEncoders.py:317,321 taken in the epilogue (a 32-token title is finished; a longer sequence is the mean of its
 *        S / 32 block rows).  Needs M % 32 == 0, M >= 4096, the LayerNorm epilogue with a dense residual, 16-byte operands.
 * lda/ldw/ldr/ldc are in elements.
 */
typedef struct {
    const float* a;       int64_t lda;
    const int32_t* a_ids; const float* a_pe; int64_t lda_pe; int32_t a_period;
    const float* w;       int64_t ldw;
    const float* bias;
    const float* res;     int64_t ldr;   int32_t res_div;
    const int32_t* res_ids; const float* res_pe; int64_t ldr_pe; int32_t res_period;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    float* c;             int64_t ldc;
    int32_t M, N, K;
    int32_t act;
    int32_t res_mod;      /* > 0 (res_ids == NULL): residual row = (r / res_div) % res_mod -- a periodic table */
    int32_t pool32;       /* 1: (LayerNorm epilogue) c is [M / 32, N]: row r = mean of result rows 32 r .. 32 r + 31 */
    float* ln_rstd;       /* optional [M]: 1 / sqrt(var + eps) of every row's LayerNorm, kept for lime_layernorm_bwd_f32 */
    const int32_t* m_dev; /* optional DEVICE int: the launch computes min(*m_dev, M) rows -- M is then the capacity the buffers were
                             sized for.  Lets a launch captured in a HIP graph follow a per-batch row count (the live rows that
                             lime_compact_sequences counted) without a host round trip.  Needs 16-byte friendly operands (the LDS-DMA kernels). */
    const int32_t* c_ids; /* optional int32 [M]: the A rows are a compacted row list -- result row r is stored at c[c_ids[r] * ldc] and
                             the periodic residual is res[(c_ids[r] % res_mod) * ldr].  Needs res with res_mod > 0, no LayerNorm, act none
                             (the in_proj GEMM over the non-padding tokens of a batch). */
    float act_scale;      /* LIME_ACT_RELU_GRAD only: the factor on the passed gradient (1 / (1 - p) when the forward ReLU output went
                             through dropout in place: h > 0 <=> ReLU passed AND the mask kept); dH = dY W2 with the ReLU gradient of
                             linear1 applied in the epilogue (trainer.py:145 through newsEncoders.py:244-247).  res = h, dense rows. */
    float dropout_p;      /* > 0 (act none or ReLU, no residual / LayerNorm / pool32): v = keep(r * N + n) ? v / (1 - p) : 0 behind the activation,
                             the counter-based mask of (dropout_seed, dropout_site) as lime_dropout_f32 draws it over the [M, N] result --
                             nn.TransformerEncoderLayer's dropout behind linear1's ReLU in training mode (newsEncoders.py:244-247) */
    uint64_t dropout_seed;
    uint32_t dropout_site;
} lime_linear_args;

int lime_linear_f32(const lime_linear_args* args, void* stream);

/* n (1 .. 8) INDEPENDENT problems in one launch of the small / mid-M kernel (csrc/gemm_mid_f32.hip): the GEMMs around the token
 * encoders are latency bound -- a launch costs one tile's k loop, >= 10 us however small --, so two that do not depend on each other
 * (the two intent-attention affine1 layers, layers.py:288; gate_proj and Q, layers.py:87 / userEncoders.py:162; the positional tables
 * of the two encoders through in_proj) take one launch's time side by side.  Every problem must be one that kernel takes: 16-byte
 * friendly operands (K, N multiples of 4, aligned rows), K >= 16, no LayerNorm / pool32 / a_pe / c_ids / res_pe; m_dev is honoured.
 * Any M (it is meant for M < 4096); LIME_ERR_UNSUPPORTED names the first problem outside the kernel. */
int lime_linear_group_f32(const lime_linear_args* args, int32_t n, void* stream);

/* The kernel instantiation the calling thread's last lime_linear_f32 launched (e.g. "gemm_pp_kernel<10, true, false, 1>"),
 * as rocprofv3 names it: lets a profiler harness match its own event timings to the kernel trace. */
const char* lime_last_linear_kernel(void);

/* Which kernels take the large 16-byte-aligned problems of lime_linear_f32 (M x N tiles >= 96 of 256 x 320) -- and, with the same bit 0, of
 * lime_linear_wgrad_f32 (M >= 4096), the unmasked padded-head lime_token_attention*_f32 (S = 32 ... 512) and lime_token_attention_bwd*_f32
 * (S > 64, no key mask):
 *   1 (default): csrc/gemm_sp_f32.hip -- fp32 operands split in registers into three bf16 terms each, six bf16 MFMAs per product
 *                block with fp32 accumulation: the error of one fp32 rounding per product (the same bound as the fp32 MFMA), fp32's
 *                exponent range, 2.7x the fp32 matrix rate;
 *   0          : csrc/gemm_pp_f32.hip -- v_mfma_f32_16x16x4_f32 (an fp32 fma chain).
 *   1 | 2      : as 1, and also the gathered-residual LayerNorm GEMM (out_proj), which is slower there and stays on the fp32 kernel by default;
 *   1 | 4      : as 1, without the rules that leave badly filling launches (few 256-row tiles) to the 128- / 64-row tile kernels (tests, A/B runs);
 *   4          : as 0, without the rule that hands big-M problems with few 128-row tiles to the 64-row-tile kernel (tests).
 * Process-wide; returns the previous setting; any other argument only queries.  LIME_SPLIT_GEMM=0 in the environment sets the start value. */
int lime_set_split_gemm(int on);

/*
 * lime_linear_bf16: the same operation on the bf16 matrix cores (BASELINE config 3: "bf16 MFMA with fp32
 * accumulate / softmax / LayerNorm").  A (or the gathered table), W and C are bf16 (uint16 storage, row-major);
 * accumulation, bias, activation, residual add and LayerNorm are fp32; the result is rounded to bf16 (nearest even).
 *   K % 8 == 0, K >= 64, lda % 8 == 0, ldw % 8 == 0 (16-byte rows); N % 4 == 0, ldc % 4 == 0.  K = 300 / N = 300 of the
 *   encoder layers are passed as 304 with zero columns / zero weight rows (lime_to_bf16 pads while converting).
 *   res_kind: 0 none; 1 fp32 rows res[(r or r % res_mod) * ldr + n]; 2 bf16 rows gathered by res_ids (+ fp32
 *   res_pe[(r % res_period)]); 3 bf16 rows res[r * ldr + n].  Kinds 2 and 3 are built with the LayerNorm epilogue only,
 *   kind 1 without it; act is none, or ReLU without residual.  ln_count: the number of real columns LayerNorm divides
 *   by (zero-padded columns must have zero weights, bias, residual, gamma and beta: they come out as zeros).
 *   Any M works; the kernel is built for M >= 4096 (two 128-row workgroups per CU).
 */
typedef struct {
    const uint16_t* a;    int64_t lda;
    const int32_t* a_ids;
    const uint16_t* w;    int64_t ldw;
    const float* bias;
    const void* res;      int64_t ldr;   int32_t res_kind; int32_t res_mod;
    const int32_t* res_ids; const float* res_pe; int64_t ldr_pe; int32_t res_period;
    const float* ln_gamma; const float* ln_beta; float ln_eps; int32_t ln_count;
    uint16_t* c;          int64_t ldc;      /* pool32: float* instead, [M / 32, N] (see lime_linear_args.pool32) */
    int32_t M, N, K;
    int32_t act;
    int32_t pool32;       /* 1: LayerNorm epilogue with a bf16 residual (res_kind 3) only; M % 32 == 0; fp32 block means */
    int32_t reserved;     /* must be 0 */
    const int32_t* m_dev; /* optional device int: min(*m_dev, M) rows are computed (see lime_linear_args.m_dev) */
    const int32_t* c_ids; /* optional int32 [M]: result row r goes to c[c_ids[r] * ldc], the fp32 periodic residual (res_kind 1 with
                             res_mod > 0) is indexed by c_ids[r] % res_mod; no LayerNorm, act none (see lime_linear_args.c_ids) */
} lime_linear_bf16_args;

int lime_linear_bf16(const lime_linear_bf16_args* args, void* stream);

/*
 * lime_encoder_ffn_bf16: the feed-forward half of an encoder layer in one launch on the bf16 matrix cores
 * (newsEncoders.py:244-247 as nn.TransformerEncoderLayer runs it, :316-321 with the token mean pooling):
 *     y = LayerNorm(x + W2 relu(W1 x + b1) + b2)          out = y (bf16 rows), or with pool32 the fp32 means of 32-row blocks
 * x: bf16 [M, ldx], the E real columns followed by zero columns up to lime_ffn_bf16_model_columns() (304; E = 300 is carried as
 * 304, as lime_linear_bf16 produces it).  w1p / w2p: the weights as lime_ffn_pack_bf16 lays them out.  The hidden state stays
 * in registers and the layer input is read once: against lime_linear_bf16 x 2 the launch moves 0.6 KB instead of 3.8 KB per token
 * through HBM.  Accumulation, residual and LayerNorm are fp32; b1 is applied in bf16 (it rides in the GEMM).
 * Built for 289 <= E < 304 and F % 128 == 0; anything else returns LIME_ERR_UNSUPPORTED (use lime_linear_bf16).
 * m_dev: optional device row count, as in lime_linear_args.
 */
typedef struct {
    const uint16_t* x;    int64_t ldx;
    const uint16_t* w1p;                    /* lime_ffn_pack_bf16's two outputs (opaque: the kernel's weight-ring slots, */
    const uint16_t* w2p;                    /* each one contiguous block in the order of its LDS image); 16-byte aligned   */
    const float* b2;                        /* fp32 [E] */
    const float* ln_gamma; const float* ln_beta; float ln_eps;     /* fp32 [E] */
    int32_t pool32;                         /* 1: out is float [M / 32, ldo] (M % 32 == 0); 0: out is bf16 [M, ldo] */
    void* out;            int64_t ldo;      /* >= 304 columns; the columns behind E come out as zeros */
    int32_t M, E, F;
    int32_t reserved;                       /* must be 0 */
    const int32_t* m_dev;
} lime_ffn_bf16_args;

int lime_encoder_ffn_bf16(const lime_ffn_bf16_args* args, void* stream);

/* w1 fp32 [F, E] (ld ldw1), b1 fp32 [F], w2 fp32 [E, F] (ld ldw2) -> w1p, w2p: bf16 buffers of lime_ffn_pack_bf16_size(F, 0) and
 * (F, 1) elements (F * 320 and F * 304). */
int64_t lime_ffn_pack_bf16_size(int32_t F, int32_t which);
int lime_ffn_pack_bf16(const float* w1, int64_t ldw1, const float* b1, const float* w2, int64_t ldw2, int32_t E, int32_t F,
                       uint16_t* w1p, uint16_t* w2p, void* stream);

/*
 * lime_encoder_block_bf16: everything of an encoder layer behind the attention core in one launch
 * (nn.TransformerEncoderLayer as newsEncoders.py:244-247, 316-321 run it):
 *     x1 = LayerNorm1(res + add_rows + attn Wo^T)          y = LayerNorm2(x1 + W2 relu(W1 x1 + b1) + b2)
 * attn: bf16 [M, lda] (the E real columns + zero columns up to 304, as lime_token_attention_bf16 writes it with out_cols = 304).
 * res (bf16 rows of 304 columns): res_kind 2 = the word table [res_rows, ldr] gathered by res_ids[M] (layer 0), res_kind 3 = the layer
 * input [M, ldr].  add_rows: fp32 [add_period, ld_add], row r % add_period is added to token r -- out_proj's bias, with the positional
 * rows added to it where the residual is the bare word rows (add_period = S), or the bias alone (add_period = 1).
 * x1 is rounded to bf16 (it feeds the bf16 GEMM and is the second residual, exactly as when lime_linear_bf16 stores it) and never
 * leaves the CU: the attention tile, the residual rows and x1 share one stationary LDS image.  out / pool32 / m_dev as in
 * lime_encoder_ffn_bf16.  w0p: lime_oproj_pack_bf16's output; ln1_* fp32 [E], 16-byte aligned.
 */
typedef struct {
    const uint16_t* attn; int64_t lda;
    const uint16_t* w0p;
    const float* add_rows; int64_t ld_add; int32_t add_period;
    int32_t res_kind;
    const uint16_t* res;  int64_t ldr;  int64_t res_rows;
    const int32_t* res_ids;
    const float* ln1_gamma; const float* ln1_beta; float ln1_eps;
    int32_t pool32;
    const uint16_t* w1p;  const uint16_t* w2p;  const float* b2;
    const float* ln2_gamma; const float* ln2_beta; float ln2_eps;
    int32_t M, E, F;
    void* out;            int64_t ldo;
    const int32_t* m_dev;
} lime_encoder_block_bf16_args;

int lime_encoder_block_bf16(const lime_encoder_block_bf16_args* args, void* stream);

/* out_proj weight fp32 [E, E] (ld ldw) -> wp: bf16 buffer of lime_oproj_pack_bf16_size() elements (the kernel's ring slots) */
int64_t lime_oproj_pack_bf16_size(void);
int lime_oproj_pack_bf16(const float* w, int64_t ldw, int32_t E, uint16_t* wp, void* stream);

/*
 * lime_inproj_bf16: the q / k / v projection in front of lime_token_attention_bf16, activation-stationary (csrc/inproj_bf16.hip):
 *     out[c_ids[r] or r, n] = bf16( a[a_ids[r] or r, :] . w[n, :] + add_rows[(c_ids[r] or r) % add_period, n] ),   n < N
 * a: bf16 rows of K valid columns (the word table gathered by a_ids, or the layer input); wp: lime_inproj_pack_bf16's output for the
 * fp32 weight [N, K] with the heads already padded to 32 columns (lime_pad_heads_f32), N a multiple of 320, K <= 320, K % 8 == 0;
 * add_rows fp32 [add_period, >= N]: bias, or positional rows x weight + bias (the linear identity of SURVEY section 7).
 * The same operation as lime_linear_bf16 with c_ids (that kernel remains for other shapes); this one reads the tile once for all
 * N columns.  a_rows / out_rows: the row counts of a and out (32-bit offset checks).  m_dev: optional device row count.
 */
typedef struct {
    const uint16_t* a;   int64_t lda;  int64_t a_rows;  const int32_t* a_ids;
    const uint16_t* wp;
    const float* add_rows; int64_t ld_add; int32_t add_period;
    int32_t M, N, K;
    int32_t reserved;
    const int32_t* c_ids;
    uint16_t* out;       int64_t ldo;  int64_t out_rows;
    const int32_t* m_dev;
} lime_inproj_bf16_args;

int lime_inproj_bf16(const lime_inproj_bf16_args* args, void* stream);
int64_t lime_inproj_pack_bf16_size(int32_t N);
int lime_inproj_pack_bf16(const float* w, int64_t ldw, int32_t N, int32_t K, uint16_t* wp, void* stream);

/* the column count (304) the bf16 encoder-block kernels carry the model dimension in */
int32_t lime_ffn_bf16_model_columns(void);

/*
 * lime_to_bf16: dst[r, c] = bf16(src[r, c]) for r < rows, c < cols, zero for the padding up to [rows_out, cols_out]
 * (src fp32 [rows, cols] with leading dimension lds; dst bf16 [rows_out, ldd]).  Converts the word table, the weights
 * and pads K = 300 -> 304 / N = 300 -> 304 on the way.
 */
int lime_to_bf16(const float* src, int64_t lds, int64_t rows, int32_t cols, uint16_t* dst, int64_t ldd, int64_t rows_out,
                 int32_t cols_out, void* stream);

/* lime_mean_pool_bf16: out[s, 0:dim) = mean_t float(x[(s * S + t), 0:dim))  -- bf16 input, fp32 output */
int lime_mean_pool_bf16(const uint16_t* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                        void* stream);

/*
 * lime_embed_pe_f32: out[r, :] = table[ids[r], :] + pe[(r % period), :]   (pe may be NULL)
 * The stand-alone word-embedding gather (newsEncoders.py:311-312 + :827); the HBM-bound kernel of the
 * path.  ids int32 [rows]; table [V, dim]; out [rows, ldo].
 */
int lime_embed_pe_f32(const int32_t* ids, const float* table, int64_t ld_table, const float* pe, int64_t ld_pe,
                      int32_t period, float* out, int64_t ldo, int64_t rows, int32_t dim, void* stream);

/*
 * lime_token_attention_f32: softmax(Q K^T * scale [+ key mask]) V per (sequence, head), exact-fp32 MFMA.
 * Replaces the attention core of nn.MultiheadAttention inside the TransformerEncoderLayers
 * (newsEncoders.py:316,320; unmasked) and layers.MultiHeadAttention.forward (layers.py:227-237; key mask
 * filled with -1e9).  q/k/v: row (seq * S + t), column (head * head_stride + d), d < head_dim, leading dimension
 * ld_qkv (the three may alias one packed buffer).  head_stride == head_dim is the packed layout of the reference;
 * head_stride = 32 (heads padded with zero columns, see lime_pad_heads_f32) lets the kernel use aligned 16-byte loads.
 * key_mask: uint8 [n_seq, S], 0 = masked, or NULL.  out[(seq * S + t) * ldo + head * head_dim + d] (always packed).
 * Requires S <= 512, head_dim <= 32, head_stride >= head_dim.
 */
int lime_token_attention_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const uint8_t* key_mask,
                             float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim,
                             int32_t head_stride, float scale, void* stream);

/* lime_token_attention_count_f32: lime_token_attention_f32 with an optional device-side sequence count (min(*n_seq_dev, n_seq)
 * sequences are computed; n_seq is the capacity): the masked attention of a compacted batch inside a HIP graph. */
int lime_token_attention_count_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const uint8_t* key_mask,
                                   const int32_t* n_seq_dev, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head,
                                   int32_t head_dim, int32_t head_stride, float scale, void* stream);

/*
 * lime_token_attention_rows_f32: lime_token_attention_f32 (unmasked, heads padded to 32 columns) over COMPACTED sequences: the
 * q / k / v row of token (seq * S + t) is row_map[seq * S + t] -- a live token's own row, or one of the S rows that all padding
 * tokens at position t share (lime_compact_sequences) -- and min(*n_seq_dev, n_seq) sequences are computed when n_seq_dev is
 * given (a device int: the launch sits in a HIP graph, the count changes per batch).  out rows stay dense: (seq * S + t).
 * S in {32, 64, 96, 128, 256, 512}; q / k / v 16-byte aligned, ld_qkv % 4 == 0.
 */
int lime_token_attention_rows_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const int32_t* row_map,
                                  const int32_t* n_seq_dev, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head,
                                  int32_t head_dim, float scale, void* stream);

/*
 * lime_compact_sequences: index lists for encoding only what differs in a batch of token sequences (ids int32 [n_seq, S], 0 = the
 * padding word).  The reference encodes every slot (newsEncoders.py:311-321), including history slots padded with the all-zero
 * <PAD> news (corpus.py:476-477) and the padding behind every text; those are exact repetitions:
 *   seq_inv  [n_seq]          compact index of every sequence; all-padding sequences share ONE representative (index n_live)
 *   ids_c    [(n_seq + 1) S]  ids in compact order (representative: zeros)
 *   row_map  [(n_seq + 1) S]  compact token row -> its q/k/v row: itself when live, pad_base + t for a padding token
 *   tok_ids / tok_rows [(n_seq + 1) S]  ids and compact rows of the live tokens, in (compact sequence, position) order; behind them S
 *                             more entries (id 0 -> row pad_base + t): the padding rows themselves, for callers that produce them
 *                             in the same in_proj launch (counts[4] rows)
 *   counts   [5]              n_live + 1, (n_live + 1) S, live tokens, n_live, live tokens + S   (device memory: the m_dev /
 *                             n_seq_dev arguments of the other entry points)
 * Ordered and deterministic (no atomics).  work: lime_compact_sequences_workspace(n_seq) int32 words.
 */
int lime_compact_sequences(const int32_t* ids, int32_t n_seq, int32_t S, int32_t pad_base, int32_t* seq_inv, int32_t* ids_c,
                           int32_t* row_map, int32_t* tok_ids, int32_t* tok_rows, int32_t* counts, int32_t* work, void* stream);

/* dst[r, 0 .. cols) = src[r % S, 0 .. cols) for every row r < rows with ids[r] == 0 (the padding word); other rows are left alone.  The q / k / v
 * rows of a padding token depend on its position only (table[0] W^T + (PE W^T + b)[t]): the training forward runs in_proj over the live
 * tokens (lime_linear_f32 with c_ids) and copies the S padding rows into the rest with this.  cols % 4 == 0, 16-byte aligned rows. */
int lime_fill_pad_rows_f32(const int32_t* ids, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t S, int32_t cols,
                           void* stream);
int64_t lime_compact_sequences_workspace(int32_t n_seq);

/* lime_token_attention_rows_bf16: lime_token_attention_bf16 over compacted sequences (row_map / n_seq_dev as in
 * lime_token_attention_rows_f32); S in {32, 64, 128}, even head_dim. */
int lime_token_attention_rows_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, const int32_t* row_map,
                                   const int32_t* n_seq_dev, uint16_t* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head,
                                   int32_t head_dim, float scale, int32_t out_cols, void* stream);

/*
 * lime_token_attention_bf16: the unmasked encoder-layer attention on bf16 storage (config 3): q / k / v bf16 with every
 * head at 32 columns (64-byte head rows), out bf16 packed [.., n_head * head_dim] plus zero columns up to out_cols (the
 * K padding the next GEMM reads).  S in {32, 64, 128}: Q.K^T and P.V on v_mfma_f32_32x32x16_bf16 (fp32 scores, softmax and
 * accumulation; P rounded to bf16); S in {256, 512}: operands widened to fp32, fp32 MFMA core.
 */
int lime_token_attention_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, uint16_t* out,
                              int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, float scale,
                              int32_t out_cols, void* stream);

/*
 * lime_pad_heads_f32: dst[(blk * head_stride + d), :] = d < head_dim ? src[(blk * head_dim + d), :] : 0 for blk < n_blk.
 * Pads the rows of in_proj_weight / in_proj_bias (n_blk = 3 * n_head blocks of head_dim rows, `cols` columns) so that the
 * in_proj GEMM writes every head at a 128-byte aligned column offset.  src [n_blk * head_dim, cols] (ld lds),
 * dst [n_blk * head_stride, cols] (ld ldd).
 */
int lime_pad_heads_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t n_blk, int32_t head_dim,
                       int32_t head_stride, int32_t cols, void* stream);

/* lime_mean_pool_f32: out[s, :] = mean_t x[(s * S + t), :]   (newsEncoders.py:317,321; padding included) */
int lime_mean_pool_f32(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                       void* stream);
/* the same over the first min(n_seq, *n_seq_dev) sequences (a compacted batch: device-side count, the rows behind it are not
 * read and their outputs not written); S <= 16 rows per sequence (the block means `pool32` leaves), dim % 4 == 0 */
int lime_mean_pool_count_f32(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                             const int32_t* n_seq_dev, void* stream);

/*
 * lime_bucketize_f32: b = min(trunc(log(max(x,1)) / log(86400) * (10/7)), 9)  (newsEncoders.py:53-58),
 * evaluated by comparison against the nine fp32 cut points of the reference's own fp32 evaluation, so the
 * indices are bit-exact without a device logf.  out int32 [n].  NaN -> 0, +inf -> 9 (the reference raises).
 */
int lime_bucketize_f32(const float* x, int32_t* out, int64_t n, void* stream);
/* The same rule for another num_buckets (config.py:59): out = #{k : x >= cuts[k]} against an ascending device table of the
 * num_buckets - 1 fp32 cut points of b = min(trunc(log(max(x,1)) / log(86400) * (num_buckets/7)), num_buckets - 1), which the
 * host derives from the reference formula's own fp32 evaluation (lime_cikm25_amd.newsEncoders.bucket_cut_points). */
int lime_bucketize_cuts_f32(const float* x, const float* cuts, int32_t n_cuts, int32_t* out, int64_t n, void* stream);

/*
 * lime_topic_rep_f32: out[r, :] = category_affine(cat[cat_table[cat[r]], sub_table[sub[r]]])
 * (newsEncoders.py:340-342 and userEncoders.py:103-105,115-117).  cat/sub int32 [rows]; tables [*, dc]/[*, ds];
 * w [dout, dc + ds]; also writes the two raw embedding rows to emb_out[r, 0:dc+ds] when emb_out != NULL
 * (feature_fusion, newsEncoders.py:221-225).
 */
int lime_topic_rep_f32(const int32_t* cat, const int32_t* sub, const float* cat_table, const float* sub_table,
                       int32_t dc, int32_t ds, const float* w, const float* bias, int32_t dout, float* out, int64_t ldo,
                       float* emb_out, int64_t ld_emb, int64_t rows, void* stream);

/*
 * lime_intent_fuse_f32: the tail of newsEncoders.CROWN.forward (:355-371).
 *   intents   [2, M, k, D]  title rows then body rows (the relu'd intent embeddings, :288)
 *   att_hidden[2, M, k, A]  tanh(affine1(intents)) (layers.py:288)
 *   affine2_t / affine2_b [A]  (layers.py:290, no bias)
 *   per news: alpha = softmax_k(att_hidden . affine2); x = sum_k alpha_k intents_k (layers.py:296-299), for title
 *   and body; s = (cos(title, body) + 1) / 2 (:297-300); content[m] = [title(D), s * body(D)] written at
 *   content[m * ldc + 0 .. 2D).
 */
int lime_intent_fuse_f32(const float* intents, const float* att_hidden, const float* affine2_t, const float* affine2_b,
                         float* content, int64_t ldc, int64_t M, int32_t k, int32_t D, int32_t A, void* stream);

/*
 * lime_additive_pool_f32: layers.Attention.forward (layers.py:285-300) after affine1:
 *   a[t] = hidden[s, t, :] . affine2;  masked_fill(mask == 0, -1e9);  alpha = softmax_t;  out[s] = sum_t alpha_t x[s, t, :]
 * hidden [n_seq * S, A] (ld ldh), x [n_seq * S, D] (ldx), mask uint8 [n_seq, S] or NULL, out [n_seq, ldo].
 */
int lime_additive_pool_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x, int64_t ldx,
                           int32_t D, const uint8_t* mask, float* out, int64_t ldo, int32_t n_seq, int32_t S, void* stream);
/* the same with an optional device-side sequence count: sequences >= min(*n_seq_dev, n_seq) are skipped, their output rows untouched (the
 * compacted batch of newsEncoders.MHSA: the rows behind the live sequences hold stale data) */
int lime_additive_pool_count_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x, int64_t ldx, int32_t D,
                                 const uint8_t* mask, const int32_t* n_seq_dev, float* out, int64_t ldo, int32_t n_seq, int32_t S, void* stream);

/*
 * lime_cand_attn_weights_f32: the attention-weight part of CandidateAware_ClickedNewsAttention.forward
 * (layers.py:70-81), one workgroup per row b.  qp = query_proj(cand_topic) [B, N, D] and kp = key_proj(hist_topic)
 * [B, H, D] come from lime_linear_f32 (layers.py:66-67).  Per head (D / n_head columns, staged in LDS):
 * scores = Q_h K_h^T / sqrt(D); masked_fill(mask == 0, -1e9); softmax over H (row max / sum by wave64 shuffles);
 * then query weights qw = softmax_N(||Q_n||_2) (:79) and agg = softmax_H(sum_n qw_n sum_heads A) (:80-81).
 * mask uint8 [B, H] (0 = padded history slot); agg out [B, H].  Requires N <= 128, H <= 512.
 */
int lime_cand_attn_weights_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N,
                               int32_t H, int32_t D, int32_t n_head, void* stream);
/* The same result with (row, head) parallelism -- one wave per (b, head) writes that head's softmaxed weights and its share of
 * ||Q_n||^2 to `workspace` (lime_cand_attn_weights_workspace(B, N, H, n_head) floats), one workgroup per row sums the heads in head
 * order and finishes: two short launches that fill the chip at B = 32 (the one-workgroup-per-row kernel above: 32 workgroups, 64 us). */
int64_t lime_cand_attn_weights_workspace(int32_t B, int32_t N, int32_t H, int32_t n_head);
int lime_cand_attn_weights_ws_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N, int32_t H,
                                  int32_t D, int32_t n_head, float* workspace, int64_t workspace_floats, void* stream);
/* the same with ONE history (kp [B / hist_div, H, D], mask [B / hist_div, H]) shared by hist_div consecutive rows: the K candidate rows of an
 * impression in Model.score_impressions -- their key projections and topic representations are computed once per impression, not per row */
int lime_cand_attn_weights_shared_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N, int32_t H,
                                      int32_t D, int32_t n_head, int32_t hist_div, float* workspace, int64_t workspace_floats, void* stream);

/*
 * lime_gate_ln_f32: the gated residual + LayerNorm of CandidateAware_ClickedNewsAttention (layers.py:84-89), one
 * workgroup per history row.  y = gate_proj.weight . x (no bias, from lime_linear_f32), s = agg[row]:
 *   g = sigmoid(s * y + bias);  v = g * (s * x) + (1 - g) * x;  out = LayerNorm(v) * gamma + beta
 * (gate_proj(s * x) = s * (W x) + b: the row scale commutes with the projection).  x, y, out: [rows, D] contiguous.
 */
int lime_gate_ln_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma,
                     const float* beta, float eps, float* out, int64_t rows, int32_t D, void* stream);

/* lime_gate_ln_f32 over the H history rows of each user row and the GraphSAGE aggregate of the result in one pass (layers.py:83-91 +
 * userEncoders.py:121,151-157).  Group g (one impression row of the user encoder; `groups` of them) reads its H history rows from
 * impression g / row_div -- x and y = gate_proj(x) are [groups / row_div, H, D]: Model.score_impressions keeps ONE copy of a history for
 * its row_div candidates -- and scale[g * H + h]; writes out[g, h, :] as lime_gate_ln_f32 does and
 *     mean_out[g, :] = (sum_{h < min(H, n_src)} out[g, h, :] + node_const) / n_src,
 * node_const [D] = the sum of the first n_src - H user-node rows (the part of the SAGEConv mean that is the same for every row; may be
 * NULL when n_src <= H).  D <= 512. */
int lime_gate_ln_sage_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma, const float* beta,
                          float eps, float* out, const float* node_const, float* mean_out, int64_t groups, int32_t H, int32_t D,
                          int32_t row_div, int32_t n_src, void* stream);

/*
 * lime_sage_mean_f32: m[b, :] = mean over the first n_src node slots of row b of cat[hist[b] (H rows),
 * user_nodes (n_user rows)] -- the aggregation PyG's SAGEConv performs for the edge list of
 * create_bipartite_graph (userEncoders.py:91-98,121,153).  Requires n_src <= H + n_user.
 */
int lime_sage_mean_f32(const float* hist, const float* user_nodes, float* out, int32_t B, int32_t H, int32_t n_user,
                       int32_t n_src, int32_t D, void* stream);

/*
 * lime_interest_match_f32: userEncoders.py:163-169 + util.py:23-49 fused, one workgroup per (row b, candidate n):
 *   a[n, h] = kp[b, h, :] . qp[b, n, :] * scale;  alpha = softmax_h (unmasked);  u[b, n, :] = sum_h alpha g[b, h, :]
 *   base = u . cand[b, n, :];  w = sigmoid(alpha_s * r) (x beta_s where r < 0 when use_penalty; |r| when not)
 *   logits[b, n] = use_weight ? base * w : base;   user_rep receives u.  Either of user_rep / logits may be NULL.
 * kp [B, H, A], qp [B, N, A], g [B, H, D], cand [B, N, D], remaining [B, N].
 */
int lime_interest_match_f32(const float* kp, const float* qp, const float* g, const float* cand, const float* remaining,
                            float* user_rep, float* logits, int32_t B, int32_t N, int32_t H, int32_t A, int32_t D,
                            float scale, float alpha_s, float beta_s, int32_t use_weight, int32_t use_penalty,
                            void* stream);

/*
 * lime_lifetime_score_f32: RemainingLifetimeWeighting.forward stand-alone (util.py:23-49):
 * logits[r] = (user[r, :] . news[r, :]) * w(remaining[r]); rows = B * N.  Same weight rule as above.
 */
int lime_lifetime_score_f32(const float* user, const float* news, const float* remaining, float* logits, int64_t rows,
                            int32_t D, float alpha_s, float beta_s, int32_t use_weight, int32_t use_penalty, void* stream);

/* lime_row_scale_f32: out[r, :] = scale[r] * x[r, :]   (layers.py:84 when use_residual_connection is off) */
int lime_row_scale_f32(const float* x, const float* scale, float* out, int64_t rows, int32_t D, void* stream);

/*
 * The masked title encoder (newsEncoders.py:566-595, LIME-MHSA-CROWN) on a compacted batch.  lime_mhsa_live_ids: ids_eff = ids with
 * -1 in the first position of every all-zero sequence whose key mask is NOT the padding news' mask (first position set,
 * corpus.py:476-477): such a sequence must be encoded, and the sentinel makes lime_compact_sequences count it as live.
 * lime_mhsa_compact_mask: behind lime_compact_sequences -- ids_c[e] = max(ids_c[e], 0) (the sentinel back to the padding word) and
 * mask_c[cs, t] = mask[seq_src[cs], t] (a compact slot without a source sequence gets the padding news' mask); n_compact_slots =
 * n_seq + 1.  mask / mask_c: uint8, non-zero = attend.
 */
int lime_mhsa_live_ids(const int32_t* ids, const uint8_t* mask, int32_t n_seq, int32_t T, int32_t* ids_eff, void* stream);
int lime_mhsa_compact_mask(int32_t* ids_c, const int32_t* seq_src, const uint8_t* mask, int32_t n_compact_slots, int32_t T,
                           uint8_t* mask_c, void* stream);

/* lime_fuse_rows_f32: LIME's fusion_method 'add' (gate == NULL: out = a + b) and 'gated' (out = gate * a + (1 - gate) * b),
 * newsEncoders.py:154-159; [rows, cols] matrices with leading dimensions. */
int lime_fuse_rows_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gate, int64_t ldg, float* out, int64_t ldo,
                       int64_t rows, int32_t cols, void* stream);

/* lime_gather_rows_f32: out[r, 0:dim) = table[idx[r], 0:dim)   (nn.Embedding lookups of small tables) */
int lime_gather_rows_f32(const int32_t* idx, const float* table, int64_t ld_table, float* out, int64_t ldo, int64_t rows,
                         int32_t dim, void* stream);

/*
 * lime_multi_copy: dst_i[0:bytes_i) = src_i[0:bytes_i) for up to LIME_MAX_COPIES device buffers in ONE launch.
 * The drop-in Model replays a HIP graph captured on its own input buffers; the caller's 17 input tensors
 * (model.py:151-154) are moved there with this instead of 17 separate copies (5 us of launch floor each).
 * `descs` is a HOST array (the only host pointer of this ABI; it is copied into the kernel arguments before the call
 * returns).  Buffers must not overlap.
 */
#define LIME_MAX_COPIES 32
typedef struct {
    const void* src;
    void* dst;
    int64_t bytes;
} lime_copy_desc;
int lime_multi_copy(const lime_copy_desc* descs, int32_t n, void* stream);

/*
 * lime_gather_rows_multi: for up to LIME_MAX_GATHERS tables at once, out_f[r, 0:row_bytes_f) = table_f[idx[r], 0:row_bytes_f)
 * (byte rows; strides in bytes).  The device-side batch assembly of dataset.py:105-141 / :192-227: one launch gathers the
 * eight per-news arrays (category, subCategory, title / abstract text, mask, entity: corpus.py:360-367) of every history
 * slot and candidate of a batch by news index, another the per-behaviour rows by behaviour index.  idx int32 [n_rows];
 * `descs` is a HOST array (copied into the kernel arguments).  Indices are not range checked (as nn.Embedding's device path).
 */
#define LIME_MAX_GATHERS 16
typedef struct {
    const void* table;  int64_t table_stride;   /* bytes between table rows */
    void* out;          int64_t out_stride;     /* bytes between output rows */
    int32_t row_bytes;  int32_t reserved;
} lime_gather_desc;
int lime_gather_rows_multi(const int32_t* idx, int64_t n_rows, const lime_gather_desc* descs, int32_t n, void* stream);

/* =====================================================================================================
 * Training step: the backward of the token encoder layers and the optimizer of
 * trainer.py:33,71-73,131-148.  Gradients of matrix products reuse lime_linear_f32 (dX = dY . W with the transposed
 * weight as its W operand); the entry points below are the pieces lime_linear_f32 cannot express.  Dense reductions
 * over the rows go through caller-provided workspaces and are fixed-order; lime_embed_bwd_f32 alone uses float atomics
 * (a word row receives contributions from many tokens), so the word-table gradient is reproducible only up to the
 * order of its fp32 additions.
 * ===================================================================================================== */

/* dW[n, k] (+)= sum_m dy[m, n] * x[m, k]: the weight gradient of y = x W^T (nn.Linear backward; loss.backward() at
 * trainer.py:145).  dy [M, N], x [M, K], dw [N, K]; exact-fp32 MFMA, M split over workgroups, partial tiles in
 * `workspace` (lime_linear_wgrad_workspace(M, N, K) floats), summed in split order.  accumulate != 0: dw += ...
 * db (optional, [N]) (+)= sum_m dy[m, n], the bias gradient: taken in the same pass as an extra all-ones column of x when
 * the padded tile grid has room for one (K = 300 of the encoder layers does), else by a column-sum pass. */
int64_t lime_linear_wgrad_workspace(int32_t M, int32_t N, int32_t K);
int lime_linear_wgrad_f32(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw, int64_t lddw, float* db,
                          int32_t M, int32_t N, int32_t K, int32_t accumulate, float* workspace, int64_t workspace_floats,
                          void* stream);

/* out[n] (+)= sum_m x[m, n]: the bias gradient.  workspace: lime_colsum_workspace(M, N) floats. */
int64_t lime_colsum_workspace(int32_t M, int32_t N);
int lime_colsum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, float* workspace,
                    int64_t workspace_floats, void* stream);

/* Backward of y = LayerNorm(z) (norm1 / norm2 of the encoder layers, newsEncoders.py:244-247) from what the forward keeps:
 * y itself and rstd (lime_linear_args.ln_rstd); xhat is recovered as (y - beta) / gamma (gamma must be non-zero).
 *   dY(r, :) = dy[(r / dy_div), :] * dy_scale      (dy_div = S, dy_scale = 1 / S: the mean pooling of :317,:321 folded in)
 *   dz[r, :] = rstd[r] * (g - mean(g) - xhat * mean(g * xhat)),  g = dY * gamma
 *   dgamma (+)= sum_r dY * xhat;  dbeta (+)= sum_r dY;  dzsum (+)= sum_r dz (the bias gradient of the linear layer whose
 *   output fed the residual sum); each of the three may be NULL.  E <= 512.
 * workspace: lime_layernorm_bwd_workspace(M, E) floats. */
int64_t lime_layernorm_bwd_workspace(int32_t M, int32_t E);
int lime_layernorm_bwd_f32(const float* dy, int64_t lddy, int32_t dy_div, float dy_scale, const float* y, int64_t ldy,
                           const float* gamma, const float* beta, const float* rstd, float* dz, int64_t lddz, int32_t M,
                           int32_t E, float* dgamma, float* dbeta, float* dzsum, int32_t accumulate, float* workspace,
                           int64_t workspace_floats, void* stream);
/* The same with a second result dz_drop[r, c] = keep(r * E + c) ? dz[r, c] / (1 - p) : 0 -- the gradient through the dropout that sits in
 * front of the residual add (dropout1 / dropout2 of the encoder layer, newsEncoders.py:244-247) with the forward's (p, seed, site) --
 * written in the same pass instead of a lime_dropout_f32 pass over dz; `dzsum` is then the column sums of dz_drop (the bias gradient of
 * the linear in front of that dropout), not of dz.  16-byte friendly operands only. */
int lime_layernorm_bwd_dropout_f32(const float* dy, int64_t lddy, int32_t dy_div, float dy_scale, const float* y, int64_t ldy,
                                   const float* gamma, const float* beta, const float* rstd, float* dz, int64_t lddz, int32_t M, int32_t E,
                                   float* dgamma, float* dbeta, float* dzsum, int32_t accumulate, float* workspace,
                                   int64_t workspace_floats, float* dz_drop, int64_t lddd, float dropout_p, uint64_t seed, uint32_t site,
                                   void* stream);

/* dh[r, c] = h[r, c] > 0 ? dh[r, c] * scale : 0, in place: ReLU backward on the saved activation of linear1 (scale = 1), or
 * ReLU + the dropout that follows it when h is the dropped-out activation (scale = 1 / (1 - p)) */
int lime_relu_bwd_f32(float* dh, int64_t lddh, const float* h, int64_t ldh, int64_t rows, int32_t cols, float scale, void* stream);

/* Backward of lime_token_attention_f32 (the encoder layers; optionally the key mask of MHSA): given q / k / v as the forward read them
 * and dout [tokens, n_head * head_dim] (packed), writes dq / dk / dv in the layout of q / k / v (row stride ld_dqkv, head
 * h at column h * head_stride; columns head_dim .. head_stride - 1 come out as zeros).  The probabilities are recomputed.
 * S <= 512, head_dim <= head_stride <= 32.  S <= 128: one pass per (sequence, head); `out` and `workspace` may be NULL.
 * 128 < S <= 512 (the 512-token bodies of BASELINE config 4): 128 x 128 blocks; needs the forward output `out` (packed like
 * dout) and lime_token_attention_bwd_workspace(n_seq, S, n_head) floats: the row statistics and one dq slab per key block behind the
 * first -- the key blocks' shares of dq are stored (no atomics) and summed in block order, so the result is bitwise reproducible.  dropout_p > 0: the forward was lime_token_attention_dropout_f32 with the same (dropout_p, seed, site).
 * key_mask (uint8 [n_seq, S], 0 = masked, or NULL; S <= 128): the masked attention of layers.MultiHeadAttention
 * (layers.py:227-232) -- masked scores are constants (-1e9) and receive no gradient. */
int64_t lime_token_attention_bwd_workspace(int32_t n_seq, int32_t S, int32_t n_head);
/* the row-statistics part of it (lse and delta per (token, head)): what lime_token_attention_dropout_f32 needs for S > 128 */
int64_t lime_token_attention_stats_workspace(int32_t n_seq, int32_t S, int32_t n_head);
int lime_token_attention_bwd_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, int64_t ld_out,
                                 const float* dout, int64_t ldo, float* dq, float* dk, float* dv, int64_t ld_dqkv, int32_t n_seq,
                                 int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride, float scale, float* workspace,
                                 int64_t workspace_floats, float dropout_p, uint64_t seed, uint32_t site, const uint8_t* key_mask,
                                 void* stream);

/* The pair for a training step whose forward keeps its softmax statistics (S > 128 pays a Q K^T pass of its own for them otherwise):
 * lime_token_attention_lse_f32 is lime_token_attention_f32 without a key mask that also writes lse [tokens, n_head] -- per (token, head)
 * the log2-domain log-sum-exp of the scaled scores, max + log2(sum 2^(s - max)) with s = scale * log2(e) * q . k --;
 * lime_token_attention_bwd_lse_f32 is lime_token_attention_bwd_f32 (no dropout, no key mask) reading it. */
int lime_token_attention_lse_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, int64_t ldo, float* lse,
                                 int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride, float scale, void* stream);
int lime_token_attention_bwd_lse_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, int64_t ld_out,
                                     const float* lse, const float* dout, int64_t ldo, float* dq, float* dk, float* dv, int64_t ld_dqkv,
                                     int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride, float scale,
                                     float* workspace, int64_t workspace_floats, void* stream);

/* Backward of lime_additive_pool_f32 (layers.Attention over a title's tokens, layers.py:285-300 / newsEncoders.py:591-592; the MHSA content
 * encoder in training): dhidden [n_seq * S, A], dx [n_seq * S, D] and da2_part [n_seq, A], the per-sequence partial rows of the
 * affine2 gradient (sum them with lime_colsum_f32).  Masked tokens receive no score gradient (their score is the constant -1e9). */
int lime_additive_pool_bwd_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x, int64_t ldx, int32_t D,
                               const uint8_t* mask, const float* dout, int64_t ldo, float* dhidden, int64_t lddh, float* dx, int64_t lddx,
                               float* da2_part, int32_t n_seq, int32_t S, void* stream);

/* ---- dropout inside the token encoders in training mode --------------------------------------------------------------
 * Masks are a pure function of (seed, site, element index) (csrc/dropout.h): element e of site `site` is kept iff
 * hash(seed, site, e) >= p * 2^32, kept values are scaled by 1 / (1 - p); the backward regenerates the mask from the same
 * triple.  torch's Philox stream is not reproduced (it differs between torch's own CPU and GPU generators as well). */

/* dst[r, c] = keep(r * cols + c) ? src[r, c] / (1 - p) : 0   (src == dst allowed).  Forward of nn.Dropout, and -- applied to a
 * gradient with the forward's (seed, site) -- its backward. */
int lime_dropout_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, float p, uint64_t seed,
                     uint32_t site, void* stream);
/* dst = dropout_site2(dropout_site1(src)) in one pass (the same values as two lime_dropout_f32 calls): the backward through the two
 * input dropouts of an encoder layer (positional, then embedding: newsEncoders.py:311-312, :827). */
int lime_dropout2_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, float p, uint64_t seed,
                      uint32_t site1, uint32_t site2, void* stream);

/* out[r, :] = drop_pe(drop_emb(table[ids[r], :]) + pe[r % period, :]): the inplace dropout on the word embeddings
 * (newsEncoders.py:311-312) and PositionalEncoding's dropout (:827) in one pass; element index r * dim + c for both sites. */
int lime_embed_pe_dropout_f32(const int32_t* ids, const float* table, int64_t ld_table, const float* pe, int64_t ld_pe,
                              int32_t period, float* out, int64_t ldo, int64_t rows, int32_t dim, float p, uint64_t seed,
                              uint32_t site_emb, uint32_t site_pe, void* stream);

/* y = LayerNorm(res + drop(t)) per row (norm1(x + dropout1(sa(x))) / norm2(x + dropout2(ff(x))) of nn.TransformerEncoderLayer);
 * rstd (optional, [M]) as lime_linear_args.ln_rstd.  E <= 512. */
int lime_dropout_add_layernorm_f32(const float* t, int64_t ldt, const float* res, int64_t ldr, const float* gamma, const float* beta,
                                   float eps, float* y, int64_t ldy, float* rstd, int64_t M, int32_t E, float p, uint64_t seed,
                                   uint32_t site, void* stream);

/* Encoder attention with dropout on the probabilities (nn.MultiheadAttention(dropout=p) in training mode):
 * out = (keep * softmax(scale q k^T) / (1 - p)) v; layouts as lime_token_attention_f32 without a key mask; mask element
 * ((seq * n_head + head) * S + i) * S + j.  S <= 512; S > 128 runs in 128 x 128 blocks and needs
 * lime_token_attention_bwd_workspace(n_seq, S, n_head) floats of workspace (row statistics), else workspace may be NULL. */
int lime_token_attention_dropout_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, int64_t ldo,
                                     int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride, float scale,
                                     float dropout_p, uint64_t seed, uint32_t site, float* workspace, int64_t workspace_floats,
                                     void* stream);

/* dtable[ids[r], :] += dx[r, :] (nn.Embedding backward, newsEncoders.py:311-312).  dtable must be initialised by the
 * caller (zeros, or a gradient to add to).  Rows with ids[r] == hot_id (the padding word, pass -1 for none) are summed
 * per wave before they touch memory.  dim <= 512. */
int lime_embed_bwd_f32(const int32_t* ids, const float* dx, int64_t lddx, float* dtable, int64_t ld_table, int64_t rows,
                       int32_t dim, int32_t hot_id, void* stream);

/* The same for a table of at most 32 rows (FreshnessEncoder's two 10-row bucket tables, newsEncoders.py:75-76): per-column
 * LDS accumulation, no atomics, fixed summation order.  Rows whose id is outside [0, table_rows) are ignored. */
/* lime_embed_bwd_sorted_f32: the same sum WITHOUT atomics -- bitwise reproducible.  order: the token positions sorted by id with a
 * stable sort, sorted_ids = ids[order] (both int32 [rows]); every row of dtable that receives a contribution is written exactly
 * once, in a fixed association (runs of equal ids summed in sorted order, in chunks of 256 positions; partials of a run that
 * crosses chunks added in chunk order); rows without a contribution keep the caller's value -- zero dtable first.  dim <= 320.
 * workspace: lime_embed_bwd_sorted_workspace(rows, dim) floats. */
int lime_embed_bwd_sorted_f32(const int32_t* order, const int32_t* sorted_ids, const float* dx, int64_t lddx, float* dtable,
                              int64_t ld_table, int64_t rows, int32_t dim, float* workspace, int64_t workspace_floats, void* stream);
int64_t lime_embed_bwd_sorted_workspace(int64_t rows, int32_t dim);

int lime_embed_bwd_small_f32(const int32_t* ids, const float* dx, int64_t lddx, float* dtable, int64_t ld_table, int64_t rows,
                             int32_t dim, int32_t table_rows, void* stream);

/* out2[0] = ||g||_2 over the flat gradient buffer, out2[1] = min(1, max_norm / (out2[0] + 1e-6))  -- the coefficient of
 * torch.nn.utils.clip_grad_norm_ (trainer.py:146-147); max_norm <= 0: out2[1] = 1.  workspace >= 1024 floats. */
int lime_grad_clip_coef_f32(const float* g, int64_t n, float max_norm, float* out2, float* workspace, int64_t workspace_floats,
                            void* stream);

/* One torch.optim.Adam step (trainer.py:33,148; amsgrad off) over flat buffers, the gradient scaled by *grad_scale (device
 * scalar, e.g. out2 + 1 above; NULL = 1): step is the 1-based step count used for the bias corrections. */
int lime_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, const float* grad_scale, void* stream);

/* loss[0] = mean_b(-log_softmax(logits[b, :])[0]) (trainer.py:71-73); dlogits (optional) = its gradient [B, K] */
int lime_nll_softmax_f32(const float* logits, int64_t ld, int32_t B, int32_t K, float* loss, float* dlogits, int64_t ldd,
                         void* stream);

/* ---- backward of the fused tail kernels (csrc/tail_backward_f32.hip) -------------------------------------------------- */

/* Backward of lime_intent_fuse_f32 (newsEncoders.py:355-371: intent attention over the k intents of title and body, cosine
 * similarity, concat): dcontent [M, ldc] holds the gradient of content[:, 0:2D]; writes d_intents [2 M k, D], d_hidden
 * [2 M k, A] and the two affine2 gradients [A] (summed over the news in a fixed order through `workspace`,
 * lime_intent_fuse_bwd_workspace(M, A) floats). */
int64_t lime_intent_fuse_bwd_workspace(int64_t M, int32_t A);
int lime_intent_fuse_bwd_f32(const float* intents, const float* hidden, const float* aff2_title, const float* aff2_body,
                             const float* dcontent, int64_t ldc, float* d_intents, float* d_hidden, float* d_aff2_title,
                             float* d_aff2_body, int64_t M, int32_t k, int32_t D, int32_t A, float* workspace,
                             int64_t workspace_floats, void* stream);

/* Backward of lime_gate_ln_f32 (layers.py:84-89): out = LayerNorm(g s x + (1 - g) x), g = sigmoid(s y + bias), s = scale[row].
 * Writes dy, dx [rows, D] (dx: the direct path only; y = W_g x is the caller's GEMM), dscale [rows], dbias / dgamma / dbeta
 * [D].  workspace: lime_gate_ln_bwd_workspace(rows, D) floats. */
int64_t lime_gate_ln_bwd_workspace(int64_t rows, int32_t D);
int lime_gate_ln_bwd_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma, const float* beta,
                         float eps, const float* dout, float* dy, float* dx, float* dscale, float* dbias, float* dgamma,
                         float* dbeta, int64_t rows, int32_t D, float* workspace, int64_t workspace_floats, void* stream);

/* Backward of lime_interest_match_f32 (userEncoders.py:158-169 + util.py:23-49) from dlogits [B, N]: dkp [B, H, A], dqp
 * [B, N, A], dg [B, H, D], dcand [B, N, D] (the lifetime weight is a constant of the parameters).  workspace:
 * lime_interest_match_bwd_workspace(B, N, H, A, D) floats (per-candidate shares of dkp / dg, summed over n in order). */
int64_t lime_interest_match_bwd_workspace(int32_t B, int32_t N, int32_t H, int32_t A, int32_t D);
int lime_interest_match_bwd_f32(const float* kp, const float* qp, const float* g, const float* cand, const float* remaining,
                                const float* dlogits, float* dkp, float* dqp, float* dg, float* dcand, int32_t B, int32_t N,
                                int32_t H, int32_t A, int32_t D, float scale, float alpha, float beta, int32_t use_weight,
                                int32_t use_penalty, float* workspace, int64_t workspace_floats, void* stream);

/* Candidate-aware attention weights (layers.py:66-81) in training mode: as lime_cand_attn_weights_f32 with the dropout of
 * layers.py:74 on the per-head probabilities (mask element ((b * n_head + h) * N + n) * H + j; dropout_p = 0: none), and its
 * backward from dagg [B, H] to dqp [B, N, D] / dkp [B, H, D] (the same dropout_p / seed / site regenerate the mask).
 * N <= 16, H <= 256; Q, K and the probabilities of one impression row live in LDS. */
int lime_cand_attn_weights_train_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N, int32_t H,
                                     int32_t D, int32_t n_head, float dropout_p, uint64_t seed, uint32_t site, void* stream);
int lime_cand_attn_weights_bwd_f32(const float* qp, const float* kp, const uint8_t* mask, const float* dagg, float* dqp, float* dkp,
                                   int32_t B, int32_t N, int32_t H, int32_t D, int32_t n_head, float dropout_p, uint64_t seed,
                                   uint32_t site, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIME_HIP_H */
