// The HBM/latency-bound kernels of the scoring path: embedding gathers, pooling, lifetime buckets,
// the intent tail of the CROWN news encoder and the three user-side fused kernels (candidate-aware
// attention weights, GraphSAGE mean, history-vs-candidate interest match + lifetime weighting).
// All reductions are fixed-order (wave64 shuffles, then LDS across the four waves): results are
// bitwise reproducible run to run.
#include "common.h"

namespace {

// sum over the 256 threads of a workgroup; `red` is >= 4 floats of LDS; every thread gets the total
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                       // protects `red` against the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// embedding gather (+ positional table)
// ---------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void embed_pe_kernel(const int* __restrict__ ids, const float* __restrict__ table,
                                                        long ld_table, const float* __restrict__ pe, long ld_pe, int period,
                                                        float* __restrict__ out, long ldo, long rows, int dim) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int vpr = dim / VEC;                                  // vectors per row
    const long total = rows * vpr;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / vpr;
        const int c = (int)(e - r * vpr) * VEC;
        vec_t v = *reinterpret_cast<const vec_t*>(table + (long)ids[r] * ld_table + c);
        if (pe) v += *reinterpret_cast<const vec_t*>(pe + (r % period) * ld_pe + c);
        *reinterpret_cast<vec_t*>(out + r * ldo + c) = v;
    }
}

__global__ __launch_bounds__(256) void embed_pe_scalar_kernel(const int* __restrict__ ids, const float* __restrict__ table,
                                                               long ld_table, const float* __restrict__ pe, long ld_pe,
                                                               int period, float* __restrict__ out, long ldo, long rows,
                                                               int dim) {
    const long total = rows * dim;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / dim;
        const int c = (int)(e - r * dim);
        float v = table[(long)ids[r] * ld_table + c];
        if (pe) v += pe[(r % period) * ld_pe + c];
        out[r * ldo + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// mean over the tokens of a sequence: one workgroup per sequence
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_pool_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out,
                                                         long ldo, int S, int dim) {
    const long s = blockIdx.x;
    const float inv = 1.0f / (float)S;
    for (int d = threadIdx.x; d < dim; d += 256) {
        const float* px = x + s * S * ldx + d;
        float acc = 0.f;
        for (int t = 0; t < S; ++t) acc += px[(long)t * ldx];
        out[s * ldo + d] = acc * inv;
    }
}

// 16-byte variant: dim / 4 column threads x RG row groups (each sums every RG-th token in order), then a fixed-order
// sum of the RG partials through LDS -- whole 1200-byte rows per wave instruction, ~2x the bytes in flight
__global__ __launch_bounds__(256) void mean_pool_vec4_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out,
                                                              long ldo, int S, int dim) {
    __shared__ f32x4 part[256];
    const long s = blockIdx.x;
    const int nc = dim >> 2;                       // float4 columns (<= 256)
    const int rg_n = 256 / nc;                     // row groups
    const int c = threadIdx.x % nc, rg = threadIdx.x / nc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rg < rg_n) {
        const float* px = x + s * S * ldx + c * 4;
        for (int t = rg; t < S; t += rg_n) acc += *reinterpret_cast<const f32x4*>(px + (long)t * ldx);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < rg_n; ++g) acc += part[g * nc + c];
        const float inv = 1.0f / (float)S;
        float* po = out + s * ldo + c * 4;
        po[0] = acc[0] * inv; po[1] = acc[1] * inv; po[2] = acc[2] * inv; po[3] = acc[3] * inv;
    }
}

// short "sequences" (the S / 32 block rows the pooled GEMM epilogue leaves: 4 rows at S = 128, 16 at S = 512): one thread per
// (sequence, four columns), rows summed in order -- a workgroup per sequence would be 8,000+ workgroups of almost no work
__global__ __launch_bounds__(256) void mean_pool_flat_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out,
                                                              long ldo, long n_seq, int S, int dim, const int* __restrict__ n_seq_dev) {
    const int nc = dim >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (n_seq_dev) n_seq = min(n_seq, (long)*n_seq_dev);               // compacted batch: the sequences behind the count hold nothing
    if (e >= n_seq * nc) return;
    const long s = e / nc;
    const int c = (int)(e - s * nc) * 4;
    const float* px = x + s * S * ldx + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (; t + 4 <= S; t += 4) {                                        // four rows in flight, added in row order
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(px + (long)t * ldx);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(px + (long)(t + 1) * ldx);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(px + (long)(t + 2) * ldx);
        const f32x4 r3 = *reinterpret_cast<const f32x4*>(px + (long)(t + 3) * ldx);
        acc += r0; acc += r1; acc += r2; acc += r3;
    }
    for (; t < S; ++t) acc += *reinterpret_cast<const f32x4*>(px + (long)t * ldx);
    const float inv = 1.0f / (float)S;
    float* po = out + s * ldo + c;
    po[0] = acc[0] * inv; po[1] = acc[1] * inv; po[2] = acc[2] * inv; po[3] = acc[3] * inv;
}

// ---------------------------------------------------------------------------------------------------
// lifetime / freshness buckets: comparison against the fp32 cut points (see oracle/lime_oracle.py)
// ---------------------------------------------------------------------------------------------------
__constant__ unsigned c_bucket_cuts[9] = {0x45326B18u, 0x4AF8B232u, 0x50AD53E8u, 0x567199BDu, 0x5C2861F4u,
                                          0x61EAB505u, 0x67A39429u, 0x6D6402D2u, 0x731EE960u};

__global__ __launch_bounds__(256) void bucketize_kernel(const float* __restrict__ x, int* __restrict__ out, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float v = x[e];
        int b = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) b += (v >= __uint_as_float(c_bucket_cuts[k])) ? 1 : 0;   // NaN compares false
        out[e] = b;
    }
}

// The masked title encoder (newsEncoders.py:566-595) on a compacted batch: a sequence repeats the all-padding representative when its
// ids are all zero AND its mask is the padding news' mask (first position set, corpus.py:476-477); an all-zero sequence under any
// other mask must be encoded -- it gets a sentinel (-1) in its first id so that lime_compact_sequences counts it as live.
// One wave per sequence.
__global__ __launch_bounds__(256) void mhsa_live_ids_kernel(const int* __restrict__ ids, const unsigned char* __restrict__ mask, int n, int T,
                                                             int* __restrict__ ids_eff) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    bool allz = true, padmask = true;
    for (int t = lane; t < T; t += 64) {
        allz = allz && ids[(long)s * T + t] == 0;
        padmask = padmask && ((mask[(long)s * T + t] != 0) == (t == 0));
    }
    const bool odd = (__ballot(!allz) == 0ull) && (__ballot(!padmask) != 0ull);
    for (int t = lane; t < T; t += 64) {
        const int id = ids[(long)s * T + t];
        ids_eff[(long)s * T + t] = (t == 0 && odd) ? -1 : id;
    }
}
// ... and behind the compaction: the sentinel back to the padding word, the key mask in compact order (a compact slot without a
// source sequence -- the representative, unused slots -- carries the padding news' mask)
__global__ __launch_bounds__(256) void mhsa_compact_mask_kernel(int* __restrict__ ids_c, const int* __restrict__ seq_src,
                                                                 const unsigned char* __restrict__ mask, long total, int T,
                                                                 unsigned char* __restrict__ mask_c) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long cs = e / T;
        const int t = (int)(e - cs * T);
        const int src = seq_src[cs];
        mask_c[e] = src < 0 ? (unsigned char)(t == 0) : (unsigned char)(mask[(long)src * T + t] != 0);
        const int id = ids_c[e];
        if (id < 0) ids_c[e] = 0;
    }
}

// LIME's 'add' / 'gated' fusion (newsEncoders.py:154-159): out = a + b, or gate * a + (1 - gate) * b
__global__ __launch_bounds__(256) void fuse_rows_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                         const float* __restrict__ gate, long ldg, float* __restrict__ out, long ldo,
                                                         long rows, int cols) {
    const long total = rows * cols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / cols;
        const int c = (int)(e - r * cols);
        const float x = a[r * lda + c], y = b[r * ldb + c];
        float v;
        if (gate) {
            const float g = gate[r * ldg + c];
            v = g * x + (1.0f - g) * y;
        } else {
            v = x + y;
        }
        out[r * ldo + c] = v;
    }
}

// the same against a caller-supplied ascending cut-point table (num_buckets != 10)
__global__ __launch_bounds__(256) void bucketize_cuts_kernel(const float* __restrict__ x, const float* __restrict__ cuts, int n_cuts,
                                                              int* __restrict__ out, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float v = x[e];
        int b = 0;
        for (int k = 0; k < n_cuts; ++k) b += (v >= cuts[k]) ? 1 : 0;                          // NaN compares false
        out[e] = b;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const int* __restrict__ idx, const float* __restrict__ table,
                                                           long ld_table, float* __restrict__ out, long ldo, long rows,
                                                           int dim) {
    const long total = rows * dim;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / dim;
        const int c = (int)(e - r * dim);
        out[r * ldo + c] = table[(long)idx[r] * ld_table + c];
    }
}

// ---------------------------------------------------------------------------------------------------
// topic representation: one 64-thread workgroup per news
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void topic_rep_kernel(const int* __restrict__ cat, const int* __restrict__ sub,
                                                        const float* __restrict__ cat_table, const float* __restrict__ sub_table,
                                                        int dc, int ds, const float* __restrict__ w, const float* __restrict__ bias,
                                                        int dout, float* __restrict__ out, long ldo, float* __restrict__ emb_out,
                                                        long ld_emb) {
    __shared__ float e[256];
    const long r = blockIdx.x;
    const int din = dc + ds;
    const float* crow = cat_table + (long)cat[r] * dc;
    const float* srow = sub_table + (long)sub[r] * ds;
    for (int i = threadIdx.x; i < din; i += 64) {
        const float v = (i < dc) ? crow[i] : srow[i - dc];
        e[i] = v;
        if (emb_out) emb_out[r * ld_emb + i] = v;
    }
    __syncthreads();
    if (out) {
        for (int o = threadIdx.x; o < dout; o += 64) {
            const float* wr = w + (long)o * din;
            float acc = 0.f;
            for (int i = 0; i < din; ++i) acc += wr[i] * e[i];
            out[r * ldo + o] = acc + (bias ? bias[o] : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// intent attention + cosine similarity + concat: one workgroup per news
// ---------------------------------------------------------------------------------------------------
constexpr int MAX_INTENT = 8;

__global__ __launch_bounds__(256) void intent_fuse_kernel(const float* __restrict__ intents, const float* __restrict__ hidden,
                                                           const float* __restrict__ aff2_t, const float* __restrict__ aff2_b,
                                                           float* __restrict__ content, long ldc, long M, int k, int D, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][D] pooled vectors, then 4 floats of scratch
    float* pooled = sm;
    float* red = sm + 2 * D;
    const long m = blockIdx.x;
    for (int tb = 0; tb < 2; ++tb) {
        const float* aff2 = tb == 0 ? aff2_t : aff2_b;
        const float* hid = hidden + ((long)tb * M + m) * k * A;
        const float* itn = intents + ((long)tb * M + m) * k * D;
        float a[MAX_INTENT];
        float mx = -INFINITY;
        for (int kk = 0; kk < k; ++kk) {
            float part = 0.f;
            for (int j = threadIdx.x; j < A; j += 256) part += hid[kk * A + j] * aff2[j];
            a[kk] = block_sum(part, red);
            mx = fmaxf(mx, a[kk]);
        }
        float den = 0.f;
        for (int kk = 0; kk < k; ++kk) {
            a[kk] = expf(a[kk] - mx);
            den += a[kk];
        }
        const float inv = 1.0f / den;
        for (int d = threadIdx.x; d < D; d += 256) {
            float x = 0.f;
            for (int kk = 0; kk < k; ++kk) x += (a[kk] * inv) * itn[kk * D + d];
            pooled[tb * D + d] = x;
        }
    }
    __syncthreads();
    float dot = 0.f, n1 = 0.f, n2 = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float t = pooled[d], b = pooled[D + d];
        dot += t * b;
        n1 += t * t;
        n2 += b * b;
    }
    dot = block_sum(dot, red);
    n1 = block_sum(n1, red);
    n2 = block_sum(n2, red);
    // F.cosine_similarity(eps = 1e-8): each norm is clamped from below
    const float cosv = dot / (fmaxf(sqrtf(n1), 1e-8f) * fmaxf(sqrtf(n2), 1e-8f));
    const float s = (cosv + 1.0f) * 0.5f;
    for (int d = threadIdx.x; d < D; d += 256) {
        content[m * ldc + d] = pooled[d];
        content[m * ldc + D + d] = s * pooled[D + d];
    }
}

// ---------------------------------------------------------------------------------------------------
// dst[r, :] = src[r % S, :] for the rows whose id is the padding word (0): the q / k / v rows of padding tokens depend on the position
// only, so the training forward computes in_proj over the live tokens and copies these S rows into the rest (cols % 4 == 0, 16-byte rows)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_pad_rows_kernel(const int* __restrict__ ids, const float* __restrict__ src, long lds,
                                                             float* __restrict__ dst, long ldd, long rows, int S, int c4n) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const long total = rows * c4n;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / c4n;
        const int c = (int)(q - r * c4n) * 4;
        if (ids[r] == 0) *reinterpret_cast<v4*>(dst + r * ldd + c) = *reinterpret_cast<const v4*>(src + (r % S) * lds + c);
    }
}

// ---------------------------------------------------------------------------------------------------
// additive attention pooling over the tokens of a sequence (MHSA news encoder): one workgroup / sequence
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void additive_pool_kernel(const float* __restrict__ hidden, long ldh,
                                                             const float* __restrict__ aff2, int A, const float* __restrict__ x,
                                                             long ldx, int D, const unsigned char* __restrict__ mask,
                                                             float* __restrict__ out, long ldo, int S, const int* __restrict__ n_seq_dev) {
    __shared__ float alpha[512];
    __shared__ float red[4];
    const long s = blockIdx.x;
    // a compacted batch: only min(*n_seq_dev, grid) sequences are live (the rows behind them hold stale data: 100 MB of reads at config 2)
    if (n_seq_dev != nullptr && (int)blockIdx.x >= *n_seq_dev) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < S; t += 4) {
        const float* h = hidden + (s * S + t) * ldh;
        float part = 0.f;
        for (int j = lane; j < A; j += 64) part += h[j] * aff2[j];
        part = wave_sum(part);
        if (lane == 0) alpha[t] = (mask && mask[s * S + t] == 0) ? -1e9f : part;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < S; t += 256) mx = fmaxf(mx, alpha[t]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float part = 0.f;
    for (int t = threadIdx.x; t < S; t += 256) {
        const float e = expf(alpha[t] - mx);
        alpha[t] = e;
        part += e;
    }
    const float inv = 1.0f / block_sum(part, red);
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < S; ++t) acc += (alpha[t] * inv) * x[(s * S + t) * ldx + d];
        out[s * ldo + d] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// candidate-aware attention weights: one workgroup per impression row (heads one after the other: large N / H)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_attn_weights_serial_kernel(const float* __restrict__ qp, const float* __restrict__ kp,
                                                                 const unsigned char* __restrict__ mask, float* __restrict__ agg,
                                                                 int N, int H, int D, int n_head) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hd = D / n_head;
    const int hdp = hd + 1;                        // odd pitch: conflict-free column walks
    float* Qh = sm;                                // [N][hdp]   this head's query tile
    float* Kh = Qh + N * hdp;                      // [H][hdp]   this head's key tile
    float* asum = Kh + H * hdp;                    // [N][H]     sum over heads of the softmaxed weights
    float* qn2 = asum + N * H;                     // [N]        squared norm of the full query
    float* red = qn2 + N;                          // [4]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_scale = 1.0f / sqrtf((float)D);
    for (int e = threadIdx.x; e < N * H; e += 256) asum[e] = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) qn2[e] = 0.f;
    for (int head = 0; head < n_head; ++head) {
        __syncthreads();
        // per-head K / Q tiles, coalesced along the head dim
        for (int e = threadIdx.x; e < N * hd; e += 256) {
            const int n = e / hd, j = e - n * hd;
            Qh[n * hdp + j] = qp[((long)b * N + n) * D + head * hd + j];
        }
        for (int e = threadIdx.x; e < H * hd; e += 256) {
            const int h = e / hd, j = e - h * hd;
            Kh[h * hdp + j] = kp[((long)b * H + h) * D + head * hd + j];
        }
        __syncthreads();
        if (threadIdx.x < N) {
            float q2 = 0.f;
            for (int j = 0; j < hd; ++j) q2 += Qh[threadIdx.x * hdp + j] * Qh[threadIdx.x * hdp + j];
            qn2[threadIdx.x] += q2;
        }
        // one wave per query row; keys across the 64 lanes; row max / sum by wave shuffles
        for (int n = wave; n < N; n += 4) {
            float sc[8];                                            // H <= 512 -> at most 8 keys per lane
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int h = c * 64 + lane;
                float v = -INFINITY;
                if (h < H) {
                    float dot = 0.f;
                    for (int j = 0; j < hd; ++j) dot += Qh[n * hdp + j] * Kh[h * hdp + j];
                    v = dot * inv_scale;
                    if (mask[(long)b * H + h] == 0) v = -1e9f;       // layers.py:72
                }
                sc[c] = v;
                mx = fmaxf(mx, v);
            }
            mx = wave_max(mx);
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                sc[c] = (c * 64 + lane < H) ? expf(sc[c] - mx) : 0.f;
                den += sc[c];
            }
            const float inv = 1.0f / wave_sum(den);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int h = c * 64 + lane;
                if (h < H) asum[n * H + h] += sc[c] * inv;            // this wave owns row n: no race
            }
        }
    }
    __syncthreads();
    // query weights: softmax over the candidates of ||Q_n||_2 (layers.py:79); N <= 128
    float* qw = Qh;                                                    // reuse
    {
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) mx = fmaxf(mx, sqrtf(qn2[n]));
        float den = 0.f;
        for (int n = 0; n < N; ++n) den += expf(sqrtf(qn2[n]) - mx);
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += 256) qw[n] = expf(sqrtf(qn2[n]) - mx) / den;
    }
    __syncthreads();
    // agg = softmax_H(sum_n qw_n * asum[n][h])  (layers.py:80-81)
    float* v = Kh;                                                     // reuse, H floats
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += asum[n * H + h] * qw[n];
        v[h] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float part = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float e = expf(v[h] - mx);
        v[h] = e;
        part += e;
    }
    const float inv = 1.0f / block_sum(part, red);
    for (int h = threadIdx.x; h < H; h += 256) agg[(long)b * H + h] = v[h] * inv;
}

// Same result, heads spread over the four waves: wave w owns heads w, w + 4, ... with its own K / Q tiles and its own
// partial sums, so the head loop needs no workgroup barrier (the serial kernel spends most of its 80 us in them).
// Per (head, candidate) the softmax terms are identical; the sum over heads is taken wave 0..3 in a fixed order.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void cand_attn_weights_kernel(const float* __restrict__ qp, const float* __restrict__ kp,
                                                                 const unsigned char* __restrict__ mask, float* __restrict__ agg,
                                                                 int N, int H, int D, int n_head) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hd = D / n_head;
    const int hdp = hd + 1;                        // odd pitch: conflict-free column walks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (N + H) * hdp + N * H + N;
    float* Qh = sm + wave * per_wave;              // [N][hdp]   this wave's query tile
    float* Kh = Qh + N * hdp;                      // [H][hdp]   this wave's key tile
    float* asum = Kh + H * hdp;                    // [N][H]     this wave's sum over its heads of the softmaxed weights
    float* qn2 = asum + N * H;                     // [N]        this wave's share of the squared query norm
    float* red = sm + 4 * per_wave;                // [4]
    const int b = blockIdx.x;
    const float inv_scale = 1.0f / sqrtf((float)D);
    for (int e = lane; e < N * H; e += 64) asum[e] = 0.f;
    for (int e = lane; e < N; e += 64) qn2[e] = 0.f;
    for (int head = wave; head < n_head; head += 4) {
        wave_lds_fence();
        for (int e = lane; e < N * hd; e += 64) {
            const int n = e / hd, j = e - n * hd;
            Qh[n * hdp + j] = qp[((long)b * N + n) * D + head * hd + j];
        }
        for (int e = lane; e < H * hd; e += 64) {
            const int h = e / hd, j = e - h * hd;
            Kh[h * hdp + j] = kp[((long)b * H + h) * D + head * hd + j];
        }
        wave_lds_fence();
        for (int n = lane; n < N; n += 64) {
            float q2 = 0.f;
            for (int j = 0; j < hd; ++j) q2 += Qh[n * hdp + j] * Qh[n * hdp + j];
            qn2[n] += q2;
        }
        for (int n = 0; n < N; ++n) {              // keys across the 64 lanes; row max / sum by wave shuffles
            float sc[8];                           // H <= 512 -> at most 8 keys per lane
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int h = c * 64 + lane;
                float v = -INFINITY;
                if (h < H) {
                    float dot = 0.f;
                    for (int j = 0; j < hd; ++j) dot += Qh[n * hdp + j] * Kh[h * hdp + j];
                    v = dot * inv_scale;
                    if (mask[(long)b * H + h] == 0) v = -1e9f;       // layers.py:72
                }
                sc[c] = v;
                mx = fmaxf(mx, v);
            }
            mx = wave_max(mx);
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                sc[c] = (c * 64 + lane < H) ? expf(sc[c] - mx) : 0.f;
                den += sc[c];
            }
            const float inv = 1.0f / wave_sum(den);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int h = c * 64 + lane;
                if (h < H) asum[n * H + h] += sc[c] * inv;
            }
        }
    }
    __syncthreads();
    // fold the four waves' partials into wave 0's arrays (fixed order)
    float* asum0 = sm + (N + H) * hdp;
    float* qn20 = asum0 + N * H;
    for (int e = threadIdx.x; e < N * H; e += 256) {
        float t = asum0[e];
        for (int w = 1; w < 4; ++w) t += sm[w * per_wave + (N + H) * hdp + e];
        asum0[e] = t;
    }
    for (int e = threadIdx.x; e < N; e += 256) {
        float t = qn20[e];
        for (int w = 1; w < 4; ++w) t += sm[w * per_wave + (N + H) * hdp + N * H + e];
        qn20[e] = t;
    }
    __syncthreads();
    // query weights: softmax over the candidates of ||Q_n||_2 (layers.py:79)
    float* qw = sm;                                // reuse wave 0's Q tile
    {
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) mx = fmaxf(mx, sqrtf(qn20[n]));
        float den = 0.f;
        for (int n = 0; n < N; ++n) den += expf(sqrtf(qn20[n]) - mx);
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += 256) qw[n] = expf(sqrtf(qn20[n]) - mx) / den;
    }
    __syncthreads();
    // agg = softmax_H(sum_n qw_n * asum[n][h])  (layers.py:80-81)
    float* v = sm + N * hdp;                       // reuse wave 0's K tile, H floats
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += asum0[n * H + h] * qw[n];
        v[h] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float part = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float e = expf(v[h] - mx);
        v[h] = e;
        part += e;
    }
    const float inv = 1.0f / block_sum(part, red);
    for (int h = threadIdx.x; h < H; h += 256) agg[(long)b * H + h] = v[h] * inv;
}

// ---- the same weights with (row, head) parallelism: B = 32 rows put 32 workgroups on 256 CUs and the launch took 64 us ------------
// pass 1: one wave per (row b, head): that head's softmaxed weights P[b][head][n][h] and its share of ||Q_n||^2 -> workspace
// hist_div > 1: kp / mask hold ONE history for hist_div consecutive rows b (Model.score_impressions: the K candidates of an impression)
__global__ __launch_bounds__(64) void cand_attn_head_kernel(const float* __restrict__ qp, const float* __restrict__ kp,
                                                            const unsigned char* __restrict__ mask, float* __restrict__ ws,
                                                            int N, int H, int D, int n_head, int hist_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hd = D / n_head, hdp = hd + 1;       // odd pitch: conflict-free column walks
    const int lane = threadIdx.x;
    const int b = blockIdx.x / n_head, head = blockIdx.x - b * n_head;
    const long bh = b / hist_div;                  // the history (keys, mask) of this row
    float* Qh = sm;                                // [N][hdp]
    float* Kh = sm + N * hdp;                      // [H][hdp]
    float* P = ws + ((long)blockIdx.x * N) * (H + 1);             // [N][H] probabilities, then [N] squared-norm shares behind them
    float* q2out = P + (long)N * H;
    // a row's head slice is hd contiguous floats: one coalesced load per row, eight rows in flight (a load -> LDS store chain per
    // row made the 55 rows 55 L2 round trips)
    for (int j = lane; j < hd; j += 64) {
        for (int r0 = 0; r0 < N + H; r0 += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                const float* src = r < N ? qp + ((long)b * N + r) * D + head * hd : kp + (bh * H + (r - N)) * D + head * hd;
                t[u] = r < N + H ? src[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + u < N + H) sm[(r0 + u) * hdp + j] = t[u];
        }
    }
    wave_lds_fence();
    for (int n = lane; n < N; n += 64) {
        float q2 = 0.f;
        for (int j = 0; j < hd; ++j) q2 += Qh[n * hdp + j] * Qh[n * hdp + j];
        q2out[n] = q2;
    }
    const float inv_scale = 1.0f / sqrtf((float)D);
    for (int n = 0; n < N; ++n) {                  // keys across the 64 lanes; row max / sum by wave shuffles
        float sc[8];                               // H <= 512 -> at most 8 keys per lane
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int h = c * 64 + lane;
            float v = -INFINITY;
            if (h < H) {
                float dot = 0.f;
                for (int j = 0; j < hd; ++j) dot += Qh[n * hdp + j] * Kh[h * hdp + j];
                v = dot * inv_scale;
                if (mask[bh * H + h] == 0) v = -1e9f;                // layers.py:72
            }
            sc[c] = v;
            mx = fmaxf(mx, v);
            if ((c + 1) * 64 >= H) break;
        }
        mx = wave_max(mx);
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            sc[c] = (c * 64 + lane < H) ? expf(sc[c] - mx) : 0.f;
            den += sc[c];
            if ((c + 1) * 64 >= H) break;
        }
        const float inv = 1.0f / wave_sum(den);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int h = c * 64 + lane;
            if (h < H) P[(long)n * H + h] = sc[c] * inv;
            if ((c + 1) * 64 >= H) break;
        }
    }
}

// pass 2: one workgroup per row: sums over the heads in head order, query weights softmax_N(||Q_n||) (layers.py:79), agg (:80-81)
__global__ __launch_bounds__(256) void cand_attn_finish_kernel(const float* __restrict__ ws, float* __restrict__ agg, int N, int H, int n_head) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qn = sm;                                // [N]  ||Q_n||
    float* v = sm + N;                             // [H]
    float* red = v + H;                            // [4]
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = ws + ((long)b * n_head * N) * (H + 1);
    const long hstride = (long)N * (H + 1);
    for (int n = threadIdx.x; n < N; n += 256) {
        float t = 0.f;
        for (int h0 = 0; h0 < n_head; h0 += 16) {
            float u[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k] = h0 + k < n_head ? base[(h0 + k) * hstride + (long)N * H + n] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += u[k];               // head order
        }
        qn[n] = sqrtf(t);
    }
    __syncthreads();
    float qmx = -INFINITY, qden = 0.f;
    for (int n = 0; n < N; ++n) qmx = fmaxf(qmx, qn[n]);
    for (int n = 0; n < N; ++n) qden += expf(qn[n] - qmx);
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            float a = 0.f;
            for (int h0 = 0; h0 < n_head; h0 += 16) {              // a batch of heads in flight, added in head order
                float u[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k] = h0 + k < n_head ? base[(h0 + k) * hstride + (long)n * H + h] : 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) a += u[k];
            }
            acc += a * (expf(qn[n] - qmx) / qden);
        }
        v[h] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float part = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float e = expf(v[h] - mx);
        v[h] = e;
        part += e;
    }
    const float inv = 1.0f / block_sum(part, red);
    for (int h = threadIdx.x; h < H; h += 256) agg[(long)b * H + h] = v[h] * inv;
}

// ---------------------------------------------------------------------------------------------------
// GraphSAGE mean over the first n_src node slots of cat[hist[b], user_nodes]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sage_mean_kernel(const float* __restrict__ hist, const float* __restrict__ user_nodes,
                                                         float* __restrict__ out, int H, int n_src, int D) {
    const long b = blockIdx.x;
    const float inv = 1.0f / (float)n_src;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        int u = 0;
        for (; u + 8 <= n_src; u += 8) {             // eight node rows in flight, added in slot order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (u + k < H) ? hist[(b * H + u + k) * D + d] : user_nodes[(long)(u + k - H) * D + d];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; u < n_src; ++u) acc += (u < H) ? hist[(b * H + u) * D + d] : user_nodes[(long)(u - H) * D + d];
        out[b * D + d] = acc * inv;
    }
}

// ---------------------------------------------------------------------------------------------------
// history-vs-candidate attention + dot-product interest match + remaining-lifetime weight:
// one workgroup per (impression row, candidate)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interest_match_kernel(const float* __restrict__ kp, const float* __restrict__ qp,
                                                              const float* __restrict__ g, const float* __restrict__ cand,
                                                              const float* __restrict__ remaining, float* __restrict__ user_rep,
                                                              float* __restrict__ logits, int N, int H, int A, int D, float scale,
                                                              float alpha_s, float beta_s, int use_weight, int use_penalty) {
    // one workgroup per (impression row, candidate): B * N workgroups instead of B keep the chip busy
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;                     // [A]  this candidate's query
    float* al = Qs + A;                 // [H]  attention logits, then weights
    float* red = al + H;                // [4]
    const long bn = blockIdx.x;
    const long b = bn / N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < A; e += 256) Qs[e] = qp[bn * A + e];
    __syncthreads();
    // a[h] = kp[b,h,:] . q * scale: a wave streams one key row (coalesced), lanes over A, shuffle reduction
    // (four rows of a wave in flight: the loop is a chain of L2 round trips otherwise; the order of the additions is unchanged)
    for (int h0 = wave; h0 < H; h0 += 16) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = lane; j < A; j += 64) {
            const float q = Qs[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int h = h0 + 4 * u;
                part[u] += (h < H ? kp[(b * H + h) * A + j] : 0.f) * q;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = wave_sum(part[u]);
            if (lane == 0 && h0 + 4 * u < H) al[h0 + 4 * u] = t * scale;
        }
    }
    __syncthreads();
    // softmax over the history (unmasked, userEncoders.py:164)
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < H; h += 256) mx = fmaxf(mx, al[h]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float den = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float e = expf(al[h] - mx);
        al[h] = e;
        den += e;
    }
    const float inv = 1.0f / block_sum(den, red);
    // u = sum_h alpha[h] g[b,h,:];  base = u . cand[b,n,:]
    float part = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float u = 0.f;
        int h = 0;
        for (; h + 8 <= H; h += 8) {                 // eight rows in flight, added in the same order
            float gv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gv[k] = g[(b * H + h + k) * D + d];
#pragma unroll
            for (int k = 0; k < 8; ++k) u += (al[h + k] * inv) * gv[k];
        }
        for (; h < H; ++h) u += (al[h] * inv) * g[(b * H + h) * D + d];
        if (user_rep) user_rep[bn * D + d] = u;
        part += u * cand[bn * D + d];
    }
    const float base = block_sum(part, red);
    if (threadIdx.x == 0 && logits) {
        float out = base;
        if (use_weight) {
            const float r = remaining[bn];
            float w;
            if (use_penalty) {
                // util.py:40-43: positive_mask * w + negative_mask * beta * w
                w = lime_sigmoid(alpha_s * r);
                w = (r >= 0.f ? 1.f : 0.f) * w + (r < 0.f ? 1.f : 0.f) * beta_s * w;
            } else {
                w = lime_sigmoid(alpha_s * fabsf(r));
            }
            out = base * w;
        }
        logits[bn] = out;
    }
}

__global__ __launch_bounds__(256) void lifetime_score_kernel(const float* __restrict__ user, const float* __restrict__ news,
                                                             const float* __restrict__ remaining, float* __restrict__ logits,
                                                             long rows, int D, float alpha_s, float beta_s, int use_weight,
                                                             int use_penalty) {
    // one wave per (row, candidate): lanes over the embedding dim, fixed-order shuffle reduction
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float part = 0.f;
    for (int d = lane; d < D; d += 64) part += user[r * D + d] * news[r * D + d];
    const float base = wave_sum(part);
    if (lane == 0) {
        float out = base;
        if (use_weight) {
            const float rl = remaining[r];
            float w;
            if (use_penalty) {
                w = lime_sigmoid(alpha_s * rl);
                w = (rl >= 0.f ? 1.f : 0.f) * w + (rl < 0.f ? 1.f : 0.f) * beta_s * w;
            } else {
                w = lime_sigmoid(alpha_s * fabsf(rl));
            }
            out = base * w;
        }
        logits[r] = out;
    }
}

__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         float* __restrict__ out, long rows, int D) {
    const long total = rows * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) out[e] = x[e] * scale[e / D];
}

__global__ __launch_bounds__(256) void gate_ln_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float* __restrict__ out, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [D] blended row, then 4 floats of scratch
    float* v = sm;
    float* red = sm + D;
    const long r = blockIdx.x;
    const float s = scale[r];
    float part = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float xv = x[r * D + d];
        const float g = lime_sigmoid(s * y[r * D + d] + bias[d]);
        const float wc = s * xv;
        const float val = g * wc + (1.f - g) * xv;
        v[d] = val;
        part += val;
    }
    const float mean = block_sum(part, red) / (float)D;
    part = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float dv = v[d] - mean;
        part += dv * dv;
    }
    const float rstd = 1.0f / sqrtf(block_sum(part, red) / (float)D + eps);
    for (int d = threadIdx.x; d < D; d += 256) out[r * D + d] = (v[d] - mean) * rstd * gamma[d] + beta[d];
}

// gate_ln over the H history rows of one user row + the GraphSAGE mean of the result, in one pass (userEncoders.py:121,151-157 behind
// layers.py:83-91): workgroup g = one (impression, candidate) row; its history rows are read from impression g / row_div (the
// scoring layout keeps ONE copy of a history for its row_div candidates), blended with this row's attention weights, normalised
// and written to out[g * H + h]; the column sums over the first n_hist node slots + `node_const` (the sum of the user-node rows the
// mean also covers: the same vector for every row) times inv_n are the SAGEConv aggregate mean_out[g].  A wave owns a row at a time
// (row statistics by wave shuffles); the four waves' column sums are added in wave order.  D <= 512.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gate_ln_sage_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ out, const float* __restrict__ node_const,
                                                            float* __restrict__ mean_out, int H, int D, int row_div, int n_hist, float inv_n) {
    __shared__ float part[NW][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long g = blockIdx.x;
    const long src = (g / row_div) * H;
    const int nj = (D + 63) >> 6;                  // <= 8
    float bi[8], ga[8], be[8], csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = lane + 64 * j;
        const bool ok = j < nj && d < D;
        bi[j] = ok ? bias[d] : 0.f;
        ga[j] = ok ? gamma[d] : 0.f;
        be[j] = ok ? beta[d] : 0.f;
        csum[j] = 0.f;
    }
    const float inv_d = 1.0f / (float)D;
    for (int h = wave; h < H; h += NW) {
        const float s = scale[g * H + h];
        const float* xr = x + (src + h) * D;
        const float* yr = y + (src + h) * D;
        float v[8];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = lane + 64 * j;
            v[j] = 0.f;
            if (j < nj && d < D) {
                const float xv = xr[d];
                const float gt = lime_sigmoid(s * yr[d] + bi[j]);
                v[j] = gt * (s * xv) + (1.f - gt) * xv;
                sum += v[j];
            }
        }
        const float mean = wave_sum(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = lane + 64 * j;
            if (j < nj && d < D) { const float dv = v[j] - mean; sq += dv * dv; }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + eps);
        float* orow = out + (g * H + h) * D;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = lane + 64 * j;
            if (j < nj && d < D) {
                const float o = (v[j] - mean) * rstd * ga[j] + be[j];
                orow[d] = o;
                if (h < n_hist) csum[j] += o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[wave][lane + 64 * j] = csum[j];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += NW * 64) {
        float t = part[0][d];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += part[w][d];          // wave order: fixed
        if (node_const) t += node_const[d];
        mean_out[g * D + d] = t * inv_n;
    }
}

__global__ __launch_bounds__(256) void pad_heads_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd,
                                                         int n_blk, int hd, int hs, int cols) {
    const long total = (long)n_blk * hs * cols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / cols;
        const int c = (int)(e - row * cols);
        const int blk = (int)(row / hs), d = (int)(row - (long)blk * hs);
        dst[row * ldd + c] = d < hd ? src[((long)blk * hd + d) * lds_ + c] : 0.f;
    }
}

inline unsigned grid_for(long total, int per_block, unsigned cap = 2048) {
    long b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

extern "C" int lime_embed_pe_f32(const int32_t* ids, const float* table, int64_t ld_table, const float* pe, int64_t ld_pe,
                                 int32_t period, float* out, int64_t ldo, int64_t rows, int32_t dim, void* stream) {
    LIME_REQUIRE(ids && table && out, LIME_ERR_BAD_ARG, "lime_embed_pe_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && ld_table >= dim && ldo >= dim, LIME_ERR_BAD_ARG, "lime_embed_pe_f32: bad dims");
    LIME_REQUIRE(!pe || (period > 0 && ld_pe >= dim), LIME_ERR_BAD_ARG, "lime_embed_pe_f32: pe needs period > 0, ld_pe >= dim");
    if (rows == 0) return LIME_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = dim % 4 == 0 && ld_table % 4 == 0 && ldo % 4 == 0 && (!pe || ld_pe % 4 == 0) &&
                    ((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!pe || (uintptr_t)pe % 16 == 0);
    if (v4)
        hipLaunchKernelGGL((embed_pe_kernel<4>), dim3(grid_for(rows * (dim / 4), 256, 8192)), dim3(256), 0, s, ids, table,
                           (long)ld_table, pe, (long)ld_pe, period, out, (long)ldo, (long)rows, dim);
    else
        hipLaunchKernelGGL(embed_pe_scalar_kernel, dim3(grid_for(rows * dim, 256, 8192)), dim3(256), 0, s, ids, table,
                           (long)ld_table, pe, (long)ld_pe, period, out, (long)ldo, (long)rows, dim);
    return lime_check_launch("lime_embed_pe_f32");
}

static int mean_pool(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim, const int32_t* n_seq_dev,
                     void* stream, const char* who) {
    LIME_REQUIRE(x && out, LIME_ERR_BAD_ARG, "lime_mean_pool_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && dim > 0 && ldx >= dim && ldo >= dim, LIME_ERR_BAD_ARG, "lime_mean_pool_f32: bad dims");
    if (n_seq == 0) return LIME_OK;
    const bool v4 = dim % 4 == 0 && dim <= 1024 && ldx % 4 == 0 && ((uintptr_t)x % 16 == 0);
    LIME_REQUIRE(!n_seq_dev || (v4 && S <= 16), LIME_ERR_BAD_ARG,
                 "lime_mean_pool_count_f32: the counted form covers S <= 16 rows per sequence, dim %% 4 == 0, 16-byte aligned rows");
    if (v4 && S <= 16 && (n_seq >= 512 || n_seq_dev))
        hipLaunchKernelGGL(mean_pool_flat_kernel, dim3((unsigned)(((long)n_seq * (dim >> 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           x, (long)ldx, out, (long)ldo, (long)n_seq, S, dim, n_seq_dev);
    else if (v4)
        hipLaunchKernelGGL(mean_pool_vec4_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, out,
                           (long)ldo, S, dim);
    else
        hipLaunchKernelGGL(mean_pool_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, out, (long)ldo,
                           S, dim);
    return lime_check_launch(who);
}

extern "C" int lime_mean_pool_f32(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                                  void* stream) {
    return mean_pool(x, ldx, out, ldo, n_seq, S, dim, nullptr, stream, "lime_mean_pool_f32");
}

extern "C" int lime_mean_pool_count_f32(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                                        const int32_t* n_seq_dev, void* stream) {
    LIME_REQUIRE(n_seq_dev, LIME_ERR_BAD_ARG, "lime_mean_pool_count_f32: NULL count");
    return mean_pool(x, ldx, out, ldo, n_seq, S, dim, n_seq_dev, stream, "lime_mean_pool_count_f32");
}

extern "C" int lime_bucketize_f32(const float* x, int32_t* out, int64_t n, void* stream) {
    LIME_REQUIRE(x && out, LIME_ERR_BAD_ARG, "lime_bucketize_f32: NULL pointer");
    LIME_REQUIRE(n >= 0, LIME_ERR_BAD_ARG, "lime_bucketize_f32: negative count");
    if (n == 0) return LIME_OK;
    hipLaunchKernelGGL(bucketize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
    return lime_check_launch("lime_bucketize_f32");
}

extern "C" int lime_mhsa_live_ids(const int32_t* ids, const uint8_t* mask, int32_t n_seq, int32_t T, int32_t* ids_eff, void* stream) {
    LIME_REQUIRE(ids && mask && ids_eff, LIME_ERR_BAD_ARG, "lime_mhsa_live_ids: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && T > 0, LIME_ERR_BAD_ARG, "lime_mhsa_live_ids: bad dims");
    if (n_seq == 0) return LIME_OK;
    hipLaunchKernelGGL(mhsa_live_ids_kernel, dim3((unsigned)((n_seq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids, mask, n_seq, T, ids_eff);
    return lime_check_launch("lime_mhsa_live_ids");
}

extern "C" int lime_mhsa_compact_mask(int32_t* ids_c, const int32_t* seq_src, const uint8_t* mask, int32_t n_compact_slots, int32_t T,
                                      uint8_t* mask_c, void* stream) {
    LIME_REQUIRE(ids_c && seq_src && mask && mask_c, LIME_ERR_BAD_ARG, "lime_mhsa_compact_mask: NULL pointer");
    LIME_REQUIRE(n_compact_slots >= 0 && T > 0, LIME_ERR_BAD_ARG, "lime_mhsa_compact_mask: bad dims");
    const long total = (long)n_compact_slots * T;
    if (total == 0) return LIME_OK;
    hipLaunchKernelGGL(mhsa_compact_mask_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ids_c, seq_src, mask, total, T, mask_c);
    return lime_check_launch("lime_mhsa_compact_mask");
}

extern "C" int lime_fuse_rows_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gate, int64_t ldg, float* out,
                                  int64_t ldo, int64_t rows, int32_t cols, void* stream) {
    LIME_REQUIRE(a && b && out, LIME_ERR_BAD_ARG, "lime_fuse_rows_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && cols > 0 && lda >= cols && ldb >= cols && ldo >= cols && (!gate || ldg >= cols), LIME_ERR_BAD_ARG,
                 "lime_fuse_rows_f32: bad dims");
    if (rows == 0) return LIME_OK;
    hipLaunchKernelGGL(fuse_rows_kernel, dim3(grid_for(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, a, (long)lda, b, (long)ldb, gate,
                       (long)ldg, out, (long)ldo, (long)rows, cols);
    return lime_check_launch("lime_fuse_rows_f32");
}

extern "C" int lime_bucketize_cuts_f32(const float* x, const float* cuts, int32_t n_cuts, int32_t* out, int64_t n, void* stream) {
    LIME_REQUIRE(x && out && cuts, LIME_ERR_BAD_ARG, "lime_bucketize_cuts_f32: NULL pointer");
    LIME_REQUIRE(n >= 0 && n_cuts >= 0 && n_cuts <= 4096, LIME_ERR_BAD_ARG, "lime_bucketize_cuts_f32: bad counts");
    if (n == 0) return LIME_OK;
    hipLaunchKernelGGL(bucketize_cuts_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, cuts, n_cuts, out, (long)n);
    return lime_check_launch("lime_bucketize_cuts_f32");
}

extern "C" int lime_gather_rows_f32(const int32_t* idx, const float* table, int64_t ld_table, float* out, int64_t ldo,
                                    int64_t rows, int32_t dim, void* stream) {
    LIME_REQUIRE(idx && table && out, LIME_ERR_BAD_ARG, "lime_gather_rows_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && ld_table >= dim && ldo >= dim, LIME_ERR_BAD_ARG, "lime_gather_rows_f32: bad dims");
    if (rows == 0) return LIME_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * dim, 256)), dim3(256), 0, (hipStream_t)stream, idx, table,
                       (long)ld_table, out, (long)ldo, (long)rows, dim);
    return lime_check_launch("lime_gather_rows_f32");
}

// ---------------------------------------------------------------------------------------------------
// bf16 path helpers (BASELINE config 3)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {          // round to nearest even (finite inputs)
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// four output columns per thread (cols_out % 4 == 0): 8-byte stores
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, long lds, long rows, int cols,
                                                      unsigned short* __restrict__ dst, long ldd, long rows_out, int cols_out) {
    const int q = cols_out >> 2;
    const long total = rows_out * q;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / q;
        const int c = (int)(e - r * q) * 4;
        unsigned short v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r < rows && c + j < cols) ? f32_to_bf16(src[r * lds + c + j]) : (unsigned short)0;
        unsigned lo = v[0] | ((unsigned)v[1] << 16), hi = v[2] | ((unsigned)v[3] << 16);
        *reinterpret_cast<uint2*>(dst + r * ldd + c) = make_uint2(lo, hi);
    }
}

// one workgroup per sequence; thread = 4 columns x row group, fixed-order combine through LDS (as mean_pool_vec4_kernel)
__global__ __launch_bounds__(256) void mean_pool_bf16_kernel(const unsigned short* __restrict__ x, long ldx, float* __restrict__ out,
                                                              long ldo, int S, int dim) {
    __shared__ f32x4 part[256];
    const long s = blockIdx.x;
    const int nc = (dim + 3) >> 2;                 // 4-column groups (<= 256)
    const int rg_n = 256 / nc;
    const int c = threadIdx.x % nc, rg = threadIdx.x / nc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rg < rg_n) {
        const unsigned short* px = x + s * S * ldx + c * 4;
        for (int t = rg; t < S; t += rg_n) {
            const uint2 v = *reinterpret_cast<const uint2*>(px + (long)t * ldx);
            acc[0] += __builtin_bit_cast(float, v.x << 16);
            acc[1] += __builtin_bit_cast(float, v.x & 0xFFFF0000u);
            acc[2] += __builtin_bit_cast(float, v.y << 16);
            acc[3] += __builtin_bit_cast(float, v.y & 0xFFFF0000u);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < rg_n; ++g) acc += part[g * nc + c];
        const float inv = 1.0f / (float)S;
        for (int j = 0; j < 4; ++j)
            if (c * 4 + j < dim) out[s * ldo + c * 4 + j] = acc[j] * inv;
    }
}

extern "C" int lime_to_bf16(const float* src, int64_t lds, int64_t rows, int32_t cols, uint16_t* dst, int64_t ldd, int64_t rows_out,
                            int32_t cols_out, void* stream) {
    LIME_REQUIRE(src && dst, LIME_ERR_BAD_ARG, "lime_to_bf16: NULL pointer");
    LIME_REQUIRE(rows >= 0 && cols > 0 && rows_out >= rows && cols_out >= cols && lds >= cols && ldd >= cols_out, LIME_ERR_BAD_ARG,
                 "lime_to_bf16: bad dims");
    LIME_REQUIRE(cols_out % 4 == 0 && ldd % 4 == 0 && (uintptr_t)dst % 8 == 0, LIME_ERR_BAD_ARG,
                 "lime_to_bf16: cols_out and ldd must be multiples of 4, dst 8-byte aligned");
    if (rows_out == 0) return LIME_OK;
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid_for(rows_out * (cols_out / 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long)lds, (long)rows, cols, dst, (long)ldd, (long)rows_out, cols_out);
    return lime_check_launch("lime_to_bf16");
}

extern "C" int lime_mean_pool_bf16(const uint16_t* x, int64_t ldx, float* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t dim,
                                   void* stream) {
    LIME_REQUIRE(x && out, LIME_ERR_BAD_ARG, "lime_mean_pool_bf16: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && dim > 0 && ldo >= dim, LIME_ERR_BAD_ARG, "lime_mean_pool_bf16: bad dims");
    LIME_REQUIRE(dim <= 1024 && ldx % 4 == 0 && ldx >= (dim + 3) / 4 * 4 && (uintptr_t)x % 8 == 0, LIME_ERR_UNSUPPORTED,
                 "lime_mean_pool_bf16: dim <= 1024, ldx a multiple of 4 and >= dim rounded up to 4, x 8-byte aligned");
    if (n_seq == 0) return LIME_OK;
    hipLaunchKernelGGL(mean_pool_bf16_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, out, (long)ldo,
                       S, dim);
    return lime_check_launch("lime_mean_pool_bf16");
}

// ---------------------------------------------------------------------------------------------------
// multi-table row gather (device-side batch assembly): blockIdx.y = table, blockIdx.x = output row
// ---------------------------------------------------------------------------------------------------
struct GatherTable {
    lime_gather_desc d[LIME_MAX_GATHERS];
};

__global__ __launch_bounds__(64) void gather_rows_multi_kernel(const int* __restrict__ idx, const GatherTable t) {
    const lime_gather_desc d = t.d[blockIdx.y];
    const long r = blockIdx.x;
    const char* src = (const char*)d.table + (long)idx[r] * d.table_stride;
    char* dst = (char*)d.out + r * d.out_stride;
    const int n = d.row_bytes;
    if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
        const int nw = n >> 2;
        for (int i = threadIdx.x; i < nw; i += 64) reinterpret_cast<unsigned*>(dst)[i] = reinterpret_cast<const unsigned*>(src)[i];
        for (int i = (nw << 2) + threadIdx.x; i < n; i += 64) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
    }
}

extern "C" int lime_gather_rows_multi(const int32_t* idx, int64_t n_rows, const lime_gather_desc* descs, int32_t n, void* stream) {
    LIME_REQUIRE(n >= 0 && n <= LIME_MAX_GATHERS, LIME_ERR_BAD_ARG, "lime_gather_rows_multi: n = %d outside [0, %d]", n, LIME_MAX_GATHERS);
    LIME_REQUIRE(n_rows >= 0 && n_rows <= 0x7FFFFFFF, LIME_ERR_BAD_ARG, "lime_gather_rows_multi: bad row count");
    if (n == 0 || n_rows == 0) return LIME_OK;
    LIME_REQUIRE(idx && descs, LIME_ERR_BAD_ARG, "lime_gather_rows_multi: NULL pointer");
    GatherTable t;
    for (int i = 0; i < n; ++i) {
        LIME_REQUIRE(descs[i].table && descs[i].out && descs[i].row_bytes > 0 && descs[i].table_stride >= descs[i].row_bytes &&
                     descs[i].out_stride >= descs[i].row_bytes, LIME_ERR_BAD_ARG, "lime_gather_rows_multi: bad descriptor %d", i);
        t.d[i] = descs[i];
    }
    hipLaunchKernelGGL(gather_rows_multi_kernel, dim3((unsigned)n_rows, (unsigned)n), dim3(64), 0, (hipStream_t)stream, idx, t);
    return lime_check_launch("lime_gather_rows_multi");
}

// ---------------------------------------------------------------------------------------------------
// multi-copy: blockIdx.y = buffer, blockIdx.x = 16 KB piece of it
// ---------------------------------------------------------------------------------------------------
struct CopyTable {
    lime_copy_desc d[LIME_MAX_COPIES];
};
constexpr long COPY_PIECE = 16384;

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTable t) {
    const lime_copy_desc d = t.d[blockIdx.y];
    const long b0 = (long)blockIdx.x * COPY_PIECE;
    if (b0 >= d.bytes) return;
    const long n = d.bytes - b0 < COPY_PIECE ? d.bytes - b0 : COPY_PIECE;
    const char* src = (const char*)d.src + b0;
    char* dst = (char*)d.dst + b0;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const long nv = n >> 4;
        for (long i = threadIdx.x; i < nv; i += 256) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
        for (long i = (nv << 4) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    } else {
        for (long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}

extern "C" int lime_multi_copy(const lime_copy_desc* descs, int32_t n, void* stream) {
    LIME_REQUIRE(n >= 0 && n <= LIME_MAX_COPIES, LIME_ERR_BAD_ARG, "lime_multi_copy: n = %d outside [0, %d]", n, LIME_MAX_COPIES);
    LIME_REQUIRE(n == 0 || descs, LIME_ERR_BAD_ARG, "lime_multi_copy: NULL descriptor table");
    CopyTable t;
    long most = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        LIME_REQUIRE(descs[i].bytes >= 0, LIME_ERR_BAD_ARG, "lime_multi_copy: negative size");
        if (descs[i].bytes == 0) continue;
        LIME_REQUIRE(descs[i].src && descs[i].dst, LIME_ERR_BAD_ARG, "lime_multi_copy: NULL buffer");
        t.d[m++] = descs[i];
        most = descs[i].bytes > most ? descs[i].bytes : most;
    }
    if (m == 0) return LIME_OK;
    const long pieces = (most + COPY_PIECE - 1) / COPY_PIECE;
    LIME_REQUIRE(pieces <= 0x7FFFFFFF, LIME_ERR_UNSUPPORTED, "lime_multi_copy: buffer too large");
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)pieces, (unsigned)m), dim3(256), 0, (hipStream_t)stream, t);
    return lime_check_launch("lime_multi_copy");
}

extern "C" int lime_topic_rep_f32(const int32_t* cat, const int32_t* sub, const float* cat_table, const float* sub_table,
                                  int32_t dc, int32_t ds, const float* w, const float* bias, int32_t dout, float* out,
                                  int64_t ldo, float* emb_out, int64_t ld_emb, int64_t rows, void* stream) {
    LIME_REQUIRE(cat && sub && cat_table && sub_table, LIME_ERR_BAD_ARG, "lime_topic_rep_f32: NULL pointer");
    LIME_REQUIRE(out || emb_out, LIME_ERR_BAD_ARG, "lime_topic_rep_f32: nothing to write");
    LIME_REQUIRE(!out || (w && dout > 0 && ldo >= dout), LIME_ERR_BAD_ARG, "lime_topic_rep_f32: out needs w, dout, ldo");
    LIME_REQUIRE(dc > 0 && ds > 0 && dc + ds <= 256, LIME_ERR_UNSUPPORTED, "lime_topic_rep_f32: dc + ds must be in (0, 256]");
    LIME_REQUIRE(!emb_out || ld_emb >= dc + ds, LIME_ERR_BAD_ARG, "lime_topic_rep_f32: ld_emb too small");
    if (rows <= 0) return rows == 0 ? LIME_OK : LIME_ERR_BAD_ARG;
    hipLaunchKernelGGL(topic_rep_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, cat, sub, cat_table, sub_table, dc,
                       ds, w, bias, dout, out, (long)ldo, emb_out, (long)ld_emb);
    return lime_check_launch("lime_topic_rep_f32");
}

extern "C" int lime_intent_fuse_f32(const float* intents, const float* att_hidden, const float* affine2_t,
                                    const float* affine2_b, float* content, int64_t ldc, int64_t M, int32_t k, int32_t D,
                                    int32_t A, void* stream) {
    LIME_REQUIRE(intents && att_hidden && affine2_t && affine2_b && content, LIME_ERR_BAD_ARG, "lime_intent_fuse_f32: NULL pointer");
    LIME_REQUIRE(M >= 0 && k > 0 && D > 0 && A > 0 && ldc >= 2 * (int64_t)D, LIME_ERR_BAD_ARG, "lime_intent_fuse_f32: bad dims");
    LIME_REQUIRE(k <= MAX_INTENT && D <= 4096, LIME_ERR_UNSUPPORTED, "lime_intent_fuse_f32: k <= 8 and D <= 4096 only");
    if (M == 0) return LIME_OK;
    const size_t lds = (size_t)(2 * D + 4) * sizeof(float);
    hipLaunchKernelGGL(intent_fuse_kernel, dim3((unsigned)M), dim3(256), lds, (hipStream_t)stream, intents, att_hidden, affine2_t,
                       affine2_b, content, (long)ldc, (long)M, k, D, A);
    return lime_check_launch("lime_intent_fuse_f32");
}

extern "C" int lime_additive_pool_count_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x,
                                            int64_t ldx, int32_t D, const uint8_t* mask, const int32_t* n_seq_dev, float* out, int64_t ldo,
                                            int32_t n_seq, int32_t S, void* stream);

extern "C" int lime_additive_pool_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x,
                                      int64_t ldx, int32_t D, const uint8_t* mask, float* out, int64_t ldo, int32_t n_seq,
                                      int32_t S, void* stream) {
    return lime_additive_pool_count_f32(hidden, ldh, affine2, A, x, ldx, D, mask, nullptr, out, ldo, n_seq, S, stream);
}

extern "C" int lime_additive_pool_count_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x,
                                            int64_t ldx, int32_t D, const uint8_t* mask, const int32_t* n_seq_dev, float* out, int64_t ldo,
                                            int32_t n_seq, int32_t S, void* stream) {
    LIME_REQUIRE(hidden && affine2 && x && out, LIME_ERR_BAD_ARG, "lime_additive_pool_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && A > 0 && D > 0 && ldh >= A && ldx >= D && ldo >= D, LIME_ERR_BAD_ARG,
                 "lime_additive_pool_f32: bad dims");
    LIME_REQUIRE(S <= 512, LIME_ERR_UNSUPPORTED, "lime_additive_pool_f32: S %d > 512", S);
    if (n_seq == 0) return LIME_OK;
    hipLaunchKernelGGL(additive_pool_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, hidden, (long)ldh, affine2, A,
                       x, (long)ldx, D, mask, out, (long)ldo, S, n_seq_dev);
    return lime_check_launch("lime_additive_pool_f32");
}

extern "C" int lime_cand_attn_weights_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B,
                                          int32_t N, int32_t H, int32_t D, int32_t n_head, void* stream) {
    LIME_REQUIRE(qp && kp && mask && agg, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_f32: NULL pointer");
    LIME_REQUIRE(B >= 0 && N > 0 && H > 0 && D > 0 && n_head > 0 && D % n_head == 0, LIME_ERR_BAD_ARG,
                 "lime_cand_attn_weights_f32: bad dims B=%d N=%d H=%d D=%d heads=%d", B, N, H, D, n_head);
    LIME_REQUIRE(N <= 128 && H <= 512, LIME_ERR_UNSUPPORTED, "lime_cand_attn_weights_f32: N <= 128 and H <= 512 only");
    if (B == 0) return LIME_OK;
    const int hdp = D / n_head + 1;
    const size_t per_wave = (size_t)((N + H) * hdp + N * H + N);
    const size_t lds_par = (4 * per_wave + 4) * sizeof(float);
    if (lds_par <= 64 * 1024) {                 // heads spread over the waves
        hipLaunchKernelGGL(cand_attn_weights_kernel, dim3((unsigned)B), dim3(256), lds_par, (hipStream_t)stream, qp, kp, mask, agg, N,
                           H, D, n_head);
        return lime_check_launch("lime_cand_attn_weights_f32");
    }
    const size_t lds = (per_wave + 4) * sizeof(float);
    LIME_REQUIRE(lds <= 160 * 1024, LIME_ERR_UNSUPPORTED, "lime_cand_attn_weights_f32: %zu B of LDS needed", lds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)cand_attn_weights_serial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cand_attn_weights_serial_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, qp, kp, mask, agg, N, H,
                       D, n_head);
    return lime_check_launch("lime_cand_attn_weights_f32");
}

extern "C" int64_t lime_cand_attn_weights_workspace(int32_t B, int32_t N, int32_t H, int32_t n_head) {
    return (int64_t)B * n_head * N * (H + 1);
}

extern "C" int lime_cand_attn_weights_shared_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N,
                                                 int32_t H, int32_t D, int32_t n_head, int32_t hist_div, float* workspace,
                                                 int64_t workspace_floats, void* stream);

extern "C" int lime_cand_attn_weights_ws_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N, int32_t H,
                                             int32_t D, int32_t n_head, float* workspace, int64_t workspace_floats, void* stream) {
    return lime_cand_attn_weights_shared_f32(qp, kp, mask, agg, B, N, H, D, n_head, 1, workspace, workspace_floats, stream);
}

extern "C" int lime_cand_attn_weights_shared_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N,
                                                 int32_t H, int32_t D, int32_t n_head, int32_t hist_div, float* workspace,
                                                 int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(hist_div >= 1, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_shared_f32: hist_div < 1");
    LIME_REQUIRE(qp && kp && mask && agg && workspace, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_ws_f32: NULL pointer");
    LIME_REQUIRE(B >= 0 && N > 0 && H > 0 && D > 0 && n_head > 0 && D % n_head == 0, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_ws_f32: bad dims");
    LIME_REQUIRE(N <= 128 && H <= 512, LIME_ERR_UNSUPPORTED, "lime_cand_attn_weights_ws_f32: N <= 128, H <= 512 (N=%d H=%d)", N, H);
    LIME_REQUIRE(workspace_floats >= lime_cand_attn_weights_workspace(B, N, H, n_head), LIME_ERR_BAD_ARG, "lime_cand_attn_weights_ws_f32: workspace too small");
    LIME_REQUIRE((long)B * n_head < 0x7FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_cand_attn_weights_ws_f32: too many (row, head) pairs");
    if (B == 0) return LIME_OK;
    const size_t lds1 = (size_t)(N + H) * (D / n_head + 1) * sizeof(float);
    LIME_REQUIRE(lds1 <= 64 * 1024, LIME_ERR_UNSUPPORTED, "lime_cand_attn_weights_ws_f32: %zu B of LDS needed", lds1);
    hipLaunchKernelGGL(cand_attn_head_kernel, dim3((unsigned)(B * n_head)), dim3(64), lds1, (hipStream_t)stream, qp, kp, mask, workspace, N, H, D, n_head, hist_div);
    hipLaunchKernelGGL(cand_attn_finish_kernel, dim3((unsigned)B), dim3(256), (size_t)(N + H + 4) * sizeof(float), (hipStream_t)stream,
                       (const float*)workspace, agg, N, H, n_head);
    return lime_check_launch("lime_cand_attn_weights_ws_f32");
}

extern "C" int lime_sage_mean_f32(const float* hist, const float* user_nodes, float* out, int32_t B, int32_t H, int32_t n_user,
                                  int32_t n_src, int32_t D, void* stream) {
    LIME_REQUIRE(hist && user_nodes && out, LIME_ERR_BAD_ARG, "lime_sage_mean_f32: NULL pointer");
    LIME_REQUIRE(B >= 0 && H > 0 && n_user >= 0 && D > 0 && n_src > 0, LIME_ERR_BAD_ARG, "lime_sage_mean_f32: bad dims");
    LIME_REQUIRE(n_src <= H + n_user, LIME_ERR_BAD_ARG,
                 "lime_sage_mean_f32: n_src %d exceeds the %d node slots (the reference raises an index error here)", n_src,
                 H + n_user);
    if (B == 0) return LIME_OK;
    hipLaunchKernelGGL(sage_mean_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, hist, user_nodes, out, H, n_src, D);
    return lime_check_launch("lime_sage_mean_f32");
}

extern "C" int lime_interest_match_f32(const float* kp, const float* qp, const float* g, const float* cand,
                                       const float* remaining, float* user_rep, float* logits, int32_t B, int32_t N, int32_t H,
                                       int32_t A, int32_t D, float scale, float alpha_s, float beta_s, int32_t use_weight,
                                       int32_t use_penalty, void* stream) {
    LIME_REQUIRE(kp && qp && g && cand && (logits || user_rep), LIME_ERR_BAD_ARG, "lime_interest_match_f32: NULL pointer");
    LIME_REQUIRE(!(use_weight && logits) || remaining, LIME_ERR_BAD_ARG, "lime_interest_match_f32: remaining is NULL");
    LIME_REQUIRE(B >= 0 && N > 0 && H > 0 && A > 0 && D > 0, LIME_ERR_BAD_ARG, "lime_interest_match_f32: bad dims");
    if (B == 0) return LIME_OK;
    const size_t lds = (size_t)(A + H + 4) * sizeof(float);
    LIME_REQUIRE(lds <= 64 * 1024, LIME_ERR_UNSUPPORTED, "lime_interest_match_f32: %zu B of LDS needed", lds);
    hipLaunchKernelGGL(interest_match_kernel, dim3((unsigned)(B * N)), dim3(256), lds, (hipStream_t)stream, kp, qp, g, cand, remaining,
                       user_rep, logits, N, H, A, D, scale, alpha_s, beta_s, use_weight, use_penalty);
    return lime_check_launch("lime_interest_match_f32");
}

extern "C" int lime_lifetime_score_f32(const float* user, const float* news, const float* remaining, float* logits, int64_t rows,
                                       int32_t D, float alpha_s, float beta_s, int32_t use_weight, int32_t use_penalty,
                                       void* stream) {
    LIME_REQUIRE(user && news && logits, LIME_ERR_BAD_ARG, "lime_lifetime_score_f32: NULL pointer");
    LIME_REQUIRE(!use_weight || remaining, LIME_ERR_BAD_ARG, "lime_lifetime_score_f32: remaining is NULL");
    LIME_REQUIRE(rows >= 0 && D > 0, LIME_ERR_BAD_ARG, "lime_lifetime_score_f32: bad dims");
    if (rows == 0) return LIME_OK;
    hipLaunchKernelGGL(lifetime_score_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, user, news,
                       remaining, logits, (long)rows, D, alpha_s, beta_s, use_weight, use_penalty);
    return lime_check_launch("lime_lifetime_score_f32");
}

extern "C" int lime_row_scale_f32(const float* x, const float* scale, float* out, int64_t rows, int32_t D, void* stream) {
    LIME_REQUIRE(x && scale && out, LIME_ERR_BAD_ARG, "lime_row_scale_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && D > 0, LIME_ERR_BAD_ARG, "lime_row_scale_f32: bad dims");
    if (rows == 0) return LIME_OK;
    hipLaunchKernelGGL(row_scale_kernel, dim3(grid_for(rows * D, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, out, (long)rows,
                       D);
    return lime_check_launch("lime_row_scale_f32");
}

extern "C" int lime_gate_ln_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma,
                                const float* beta, float eps, float* out, int64_t rows, int32_t D, void* stream) {
    LIME_REQUIRE(y && x && scale && bias && gamma && beta && out, LIME_ERR_BAD_ARG, "lime_gate_ln_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && D > 0, LIME_ERR_BAD_ARG, "lime_gate_ln_f32: bad dims");
    LIME_REQUIRE(D <= 8192, LIME_ERR_UNSUPPORTED, "lime_gate_ln_f32: D %d > 8192", D);
    if (rows == 0) return LIME_OK;
    hipLaunchKernelGGL(gate_ln_kernel, dim3((unsigned)rows), dim3(256), (size_t)(D + 4) * sizeof(float), (hipStream_t)stream, y, x,
                       scale, bias, gamma, beta, eps, out, D);
    return lime_check_launch("lime_gate_ln_f32");
}

extern "C" int lime_gate_ln_sage_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma,
                                     const float* beta, float eps, float* out, const float* node_const, float* mean_out, int64_t groups,
                                     int32_t H, int32_t D, int32_t row_div, int32_t n_src, void* stream) {
    LIME_REQUIRE(y && x && scale && bias && gamma && beta && out && mean_out, LIME_ERR_BAD_ARG, "lime_gate_ln_sage_f32: NULL pointer");
    LIME_REQUIRE(groups >= 0 && H > 0 && D > 0 && row_div >= 1 && n_src > 0, LIME_ERR_BAD_ARG, "lime_gate_ln_sage_f32: bad dims");
    LIME_REQUIRE(D <= 512, LIME_ERR_UNSUPPORTED, "lime_gate_ln_sage_f32: D %d > 512 (use lime_gate_ln_f32 + lime_sage_mean_f32)", D);
    LIME_REQUIRE(n_src <= H || node_const, LIME_ERR_BAD_ARG, "lime_gate_ln_sage_f32: n_src %d > H %d needs node_const", n_src, H);
    LIME_REQUIRE(groups < 0x7FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_gate_ln_sage_f32: too many rows");
    if (groups == 0) return LIME_OK;
    // few rows (a training-shape batch: 32 .. 256 user rows): sixteen waves share a row's H history rows, the launch is a latency chain
    // otherwise; many rows (the scoring layout): four waves per workgroup, more workgroups per CU.  (The wave count changes the order
    // in which a column's H values are added: the two forms differ by fp32 rounding, each is deterministic.)
    if (groups < 4096)
        hipLaunchKernelGGL(gate_ln_sage_kernel<16>, dim3((unsigned)groups), dim3(1024), 0, (hipStream_t)stream, y, x, scale, bias, gamma, beta, eps,
                           out, n_src > H ? node_const : nullptr, mean_out, H, D, row_div, n_src < H ? n_src : H, 1.0f / (float)n_src);
    else
        hipLaunchKernelGGL(gate_ln_sage_kernel<4>, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, y, x, scale, bias, gamma, beta, eps,
                           out, n_src > H ? node_const : nullptr, mean_out, H, D, row_div, n_src < H ? n_src : H, 1.0f / (float)n_src);
    return lime_check_launch("lime_gate_ln_sage_f32");
}

extern "C" int lime_pad_heads_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t n_blk, int32_t head_dim,
                                  int32_t head_stride, int32_t cols, void* stream) {
    LIME_REQUIRE(src && dst, LIME_ERR_BAD_ARG, "lime_pad_heads_f32: NULL pointer");
    LIME_REQUIRE(n_blk > 0 && head_dim > 0 && head_stride >= head_dim && cols > 0 && lds >= cols && ldd >= cols, LIME_ERR_BAD_ARG,
                 "lime_pad_heads_f32: bad dims");
    hipLaunchKernelGGL(pad_heads_kernel, dim3(grid_for((long)n_blk * head_stride * cols, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long)lds, dst, (long)ldd, n_blk, head_dim, head_stride, cols);
    return lime_check_launch("lime_pad_heads_f32");
}

extern "C" int lime_fill_pad_rows_f32(const int32_t* ids, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t S,
                                      int32_t cols, void* stream) {
    LIME_REQUIRE(ids && src && dst, LIME_ERR_BAD_ARG, "lime_fill_pad_rows_f32: NULL pointer");
    LIME_REQUIRE(rows >= 0 && S > 0 && cols > 0 && lds >= cols && ldd >= cols, LIME_ERR_BAD_ARG, "lime_fill_pad_rows_f32: bad dimensions");
    LIME_REQUIRE(cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, LIME_ERR_UNSUPPORTED,
                 "lime_fill_pad_rows_f32: needs 16-byte aligned rows and cols % 4 == 0");
    if (rows == 0) return LIME_OK;
    const long total = rows * (cols / 4);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(fill_pad_rows_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, ids, src,
                       (long)lds, dst, (long)ldd, (long)rows, S, cols / 4);
    return lime_check_launch("lime_fill_pad_rows_f32");
}
