// lime_token_attention_f32: softmax(Q K^T * scale [+ key mask]) V per (sequence, head), exact fp32.
//
// The attention core of the two TransformerEncoderLayers (newsEncoders.py:316,320: 10 heads x 30,
// no mask, padded tokens attend and are attended) and of layers.MultiHeadAttention (layers.py:227-237:
// 10 heads x 20, key mask filled with -1e9).  Sequences are LDS-scale (S <= 512), so there is no
// flash-style key loop.  A wave owns 32 queries and computes the TRANSPOSED score strip
// S^T = K Q^T (keys on the MFMA rows, queries on the lanes) into accumulators, 16 registers per 32-key
// tile.  With the query on the lane
//   * the softmax max / sum over the keys is a reduction over a lane's own registers plus ONE
//     cross-half shuffle (the 32x32 C layout splits a column's 32 rows over the two half-waves),
//     instead of a 5-step shuffle tree per row;
//   * a probability register is already laid out as the B operand of the next product
//     O^T = V^T P^T (register r of tile t holds key 32t + (r&3) + 8(r>>2) + 4*half for query = lane),
//     so P never leaves the register file; the A operand is one conflict-free ds_read_b32 of the
//     staged V image per MFMA;
//   * 1/sum is applied to the 16 output registers, not to the S probabilities.
// The output tile O^T (head dim on rows, query on lanes) is transposed through a 4 KiB per-wave LDS
// scratch so the store writes whole 120-byte head rows.  K and V of a (sequence, head) pair are staged
// once per workgroup with 8-byte loads; short sequences pack 4 (S <= 32) or 2 (S <= 64) pairs per
// workgroup.
#include "common.h"

namespace {

constexpr int LDH = 36;  // pitch of K / Q rows in LDS (floats): conflict-free ds_read_b128
constexpr int LDV = 32;  // pitch of V rows: a half-wave reads 32 consecutive floats of one key
constexpr int LDO = 33;  // pitch of the output transpose scratch

struct AttnP {
    const float* q; const float* k; const float* v; long ld; const unsigned char* mask;
    float* out; long ldo; int n_seq, S, n_head, hd; float scale; int n_pair; int vec2;
};

__device__ __forceinline__ void lds_fence() {
    // LDS ops of one wave execute in order; this keeps the compiler from reordering across the point
    // and drains lgkmcnt so that a tile written by some lanes is read back by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// rows [0, rows) x head_dim of one (sequence, head) operand -> LDS image with `pitch`, zero-filled to 32 columns and
// to `rows_padded` rows; `nthr` threads starting at `t0` cooperate.  src points at (first row, head column 0).
__device__ __forceinline__ void stage_rows(float* dst, int pitch, const float* src, long ld, int rows, int rows_padded, int hd,
                                           float scale, bool vec2, int t0, int nthr) {
    if (vec2) {
        const int pairs = 16;                                   // 32 columns as 16 float2
        for (int e = t0; e < rows_padded * pairs; e += nthr) {
            const int c = (e % pairs) * 2, r = e / pairs;
            f32x2 v = {0.f, 0.f};
            if (r < rows && c < hd) v = *reinterpret_cast<const f32x2*>(src + (long)r * ld + c);   // hd even: c + 1 < hd
            dst[r * pitch + c] = v[0] * scale;
            dst[r * pitch + c + 1] = v[1] * scale;
        }
    } else {
        for (int e = t0; e < rows_padded * 32; e += nthr) {
            const int c = e & 31, r = e >> 5;
            float v = 0.f;
            if (r < rows && c < hd) v = src[(long)r * ld + c];
            dst[r * pitch + c] = v * scale;
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void token_attn_kernel(const AttnP p) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);   // (sequence, head) pairs per workgroup
    constexpr int WPP = 4 / G;                     // waves per pair
    constexpr int SP = NT * 32;                    // padded sequence length
    __shared__ __attribute__((aligned(16))) float Ks[G * SP * LDH];
    __shared__ __attribute__((aligned(16))) float Vs[G * SP * LDV];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDH];
    __shared__ float Flag[G * SP];                // 0: key takes part, 1: masked (-1e9), 2: padding (-inf)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S, hd = p.hd;
    const bool need_flags = p.mask != nullptr || (S & 31) != 0;

    // ---- stage K and V of this workgroup's pairs: WPP waves per pair ---------------------------------
    {
        const int g = wave / WPP;
        const int pair = blockIdx.x * G + g;
        const int t0 = (wave % WPP) * 64 + lane, nthr = WPP * 64;
        if (pair < p.n_pair) {
            const int seq = pair / p.n_head, head = pair - seq * p.n_head;
            const long base = (long)seq * S * p.ld + head * hd;
            stage_rows(&Ks[g * SP * LDH], LDH, p.k + base, p.ld, S, SP, hd, 1.0f, p.vec2, t0, nthr);
            stage_rows(&Vs[g * SP * LDV], LDV, p.v + base, p.ld, S, SP, hd, 1.0f, p.vec2, t0, nthr);
            if (need_flags)
                for (int key = t0; key < SP; key += nthr)
                    Flag[g * SP + key] = key >= S ? 2.f : ((p.mask && p.mask[(long)seq * S + key] == 0) ? 1.f : 0.f);
        }
    }
    __syncthreads();

    const int g = wave / WPP;
    const int pair = blockIdx.x * G + g;
    if (pair >= p.n_pair) return;                  // no barrier below this point
    const int seq = pair / p.n_head, head = pair - seq * p.n_head;
    const float* Kg = &Ks[g * SP * LDH];
    const float* Vg = &Vs[g * SP * LDV];
    const float* Fg = &Flag[g * SP];
    float* scr = &Scr[wave * 32 * LDH];
    const int krow = (0) + 4 * fh;                 // key row of accumulator register r: (r & 3) + 8 * (r >> 2) + 4 * fh

    for (int qt = wave % WPP; qt < NT; qt += WPP) {
        if (qt * 32 >= S) break;
        // ---- Q tile -> scratch (scaled), then into B fragments ---------------------------------------
        const int qrows = (S - qt * 32) < 32 ? (S - qt * 32) : 32;
        stage_rows(scr, LDH, p.q + ((long)seq * S + qt * 32) * p.ld + head * hd, p.ld, qrows, 32, hd, p.scale, p.vec2, lane, 64);
        lds_fence();
        f32x4 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(&scr[fi * LDH + kk * 8 + fh * 4]);
        lds_fence();

        // ---- S^T = K Q^T: keys on rows, this lane's query on the column ---------------------------------
        f32x16 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float init = (t * 32 < S) ? 0.f : -INFINITY;      // a tile entirely beyond S is padding
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[t][r] = init;
            if (t * 32 < S) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(&Kg[(t * 32 + fi) * LDH + kk * 8 + fh * 4]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u], qf[kk][u], sc[t], 0, 0, 0);
                }
            }
        }
        // ---- key padding / key mask (the key of register r is the same for a whole half-wave) ----------
        if (need_flags) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float f = Fg[t * 32 + (r & 3) + 8 * (r >> 2) + krow];
                    if (f == 2.f) sc[t][r] = -INFINITY;
                    else if (f == 1.f) sc[t][r] = -1e9f;      // masked_fill(mask == 0, -1e9), layers.py:233
                }
            }
        }
        // ---- softmax over the keys of this lane's query: own registers, then the other half-wave ---------
        float m = sc[0][0];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[t][r]);
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(sc[t][r] - m);
                sc[t][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ---- O^T = V^T P^T: probability registers are the B operand as they stand ----------------------
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t * 32 < S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = Vg[(t * 32 + (r & 3) + 8 * (r >> 2) + krow) * LDV + fi];
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sc[t][r], o, 0, 0, 0);
                }
            }
        }
        // ---- transpose [head dim][query] -> [query][head dim] through the scratch, store whole head rows ----
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[fi * LDO + (r & 3) + 8 * (r >> 2) + krow] = o[r] * inv;
        lds_fence();
        if (fi < hd) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 2 + fh;
                const int qi = qt * 32 + row;
                if (qi < S) p.out[((long)seq * S + qi) * p.ldo + head * hd + fi] = scr[row * LDO + fi];
            }
        }
        lds_fence();
    }
}

template <int NT>
int launch(const AttnP& p, hipStream_t s) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);
    const unsigned blocks = (unsigned)((p.n_pair + G - 1) / G);
    hipLaunchKernelGGL((token_attn_kernel<NT>), dim3(blocks), dim3(256), 0, s, p);
    return lime_check_launch("lime_token_attention_f32");
}

}  // namespace

extern "C" int lime_token_attention_f32(const float* q, const float* k, const float* v, int64_t ld_qkv,
                                        const uint8_t* key_mask, float* out, int64_t ldo, int32_t n_seq, int32_t S,
                                        int32_t n_head, int32_t head_dim, float scale, void* stream) {
    LIME_REQUIRE(q && k && v && out, LIME_ERR_BAD_ARG, "lime_token_attention_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0, LIME_ERR_BAD_ARG,
                 "lime_token_attention_f32: bad dims n_seq=%d S=%d n_head=%d head_dim=%d", n_seq, S, n_head, head_dim);
    LIME_REQUIRE(head_dim <= 32, LIME_ERR_UNSUPPORTED, "lime_token_attention_f32: head_dim %d > 32", head_dim);
    LIME_REQUIRE(S <= 512, LIME_ERR_UNSUPPORTED, "lime_token_attention_f32: S %d > 512", S);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * head_dim && ldo >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG,
                 "lime_token_attention_f32: leading dimension smaller than n_head * head_dim");
    if (n_seq == 0) return LIME_OK;
    // 8-byte loads need an even head_dim and leading dimension and 8-byte aligned bases
    const int vec2 = (head_dim % 2 == 0) && (ld_qkv % 2 == 0) && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 8 == 0);
    AttnP p{q, k, v, (long)ld_qkv, key_mask, out, (long)ldo, n_seq, S, n_head, head_dim, scale, n_seq * n_head, vec2};
    hipStream_t s = (hipStream_t)stream;
    const int nt = (S + 31) / 32;
    if (nt <= 1) return launch<1>(p, s);
    if (nt <= 2) return launch<2>(p, s);
    if (nt <= 3) return launch<3>(p, s);
    if (nt <= 4) return launch<4>(p, s);
    if (nt <= 8) return launch<8>(p, s);
    return launch<16>(p, s);
}
