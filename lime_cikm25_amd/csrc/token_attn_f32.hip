// lime_token_attention_f32: softmax(Q K^T * scale [+ key mask]) V per (sequence, head), exact fp32.
//
// The attention core of the two TransformerEncoderLayers (newsEncoders.py:316,320: 10 heads x 30,
// no mask, padded tokens attend and are attended) and of layers.MultiHeadAttention (layers.py:227-237:
// 10 heads x 20, key mask filled with -1e9).  Sequences are LDS-scale (S <= 512), so there is no
// flash-style key loop: a wave owns 32 query rows, keeps the whole 32 x S score strip in MFMA
// accumulators (16 registers per 32-key tile), does the row softmax in registers with half-wave
// shuffle reductions (the 32x32 C layout puts a row's 32 keys on the 32 lanes of a half-wave), passes
// each probability tile through a 4.5 KiB per-wave LDS scratch to re-enter as the A operand, and
// multiplies by V straight from the staged [key][32] image.  K and V of a (sequence, head) pair are
// staged once per workgroup; short sequences pack 4 (S <= 32) or 2 (S <= 64) pairs per workgroup.
#include "common.h"

namespace {

constexpr int LDH = 36;  // pitch of K / Q / P rows in LDS (floats): conflict-free ds_read_b128
constexpr int LDV = 32;  // pitch of V rows: lanes read 32 consecutive floats of one key

struct AttnP {
    const float* q; const float* k; const float* v; long ld; const unsigned char* mask;
    float* out; long ldo; int n_seq, S, n_head, hd; float scale; int n_pair;
};

__device__ __forceinline__ void lds_fence() {
    // LDS ops of one wave execute in order; this keeps the compiler from reordering across the point
    // and drains lgkmcnt so that a tile written by some lanes is read back by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int NT>
__global__ __launch_bounds__(256) void token_attn_kernel(const AttnP p) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);   // (sequence, head) pairs per workgroup
    constexpr int WPP = 4 / G;                     // waves per pair
    constexpr int SP = NT * 32;                    // padded sequence length
    __shared__ __attribute__((aligned(16))) float Ks[G * SP * LDH];
    __shared__ __attribute__((aligned(16))) float Vs[G * SP * LDV];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S, hd = p.hd;

    // ---- stage K and V of this workgroup's pairs (zero-filled beyond S and beyond head_dim) ---------
    for (int e = tid; e < G * SP * 32; e += 256) {
        const int d = e & 31, key = (e >> 5) % SP, g = (e >> 5) / SP;
        const int pair = blockIdx.x * G + g;
        float kv = 0.f, vv = 0.f;
        if (pair < p.n_pair && key < S && d < hd) {
            const int seq = pair / p.n_head, head = pair - seq * p.n_head;
            const long off = ((long)seq * S + key) * p.ld + head * hd + d;
            kv = p.k[off];
            vv = p.v[off];
        }
        Ks[(g * SP + key) * LDH + d] = kv;
        Vs[(g * SP + key) * LDV + d] = vv;
    }
    __syncthreads();

    const int g = wave / WPP;
    const int pair = blockIdx.x * G + g;
    if (pair >= p.n_pair) return;                  // no barrier below this point
    const int seq = pair / p.n_head, head = pair - seq * p.n_head;
    const float* Kg = &Ks[g * SP * LDH];
    const float* Vg = &Vs[g * SP * LDV];
    float* scr = &Scr[wave * 32 * LDH];

    for (int qt = wave % WPP; qt < NT; qt += WPP) {
        if (qt * 32 >= S) break;
        // ---- Q tile -> scratch (scaled), then into A fragments --------------------------------------
        for (int e = lane; e < 32 * 32; e += 64) {
            const int d = e & 31, i = e >> 5;
            const int qi = qt * 32 + i;
            float x = 0.f;
            if (qi < S && d < hd) x = p.q[((long)seq * S + qi) * p.ld + head * hd + d] * p.scale;
            scr[i * LDH + d] = x;
        }
        lds_fence();
        f32x4 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(&scr[fi * LDH + kk * 8 + fh * 4]);
        lds_fence();

        // ---- scores: 32 queries x S keys in accumulators ---------------------------------------------
        f32x16 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
            if (t * 32 < S) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(&Kg[(t * 32 + fi) * LDH + kk * 8 + fh * 4]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kk][u], kf[u], sc[t], 0, 0, 0);
                }
            }
        }
        // ---- key padding / key mask: column = key index on this lane ---------------------------------
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int key = t * 32 + fi;
            const bool pad = key >= S;
            const bool masked = !pad && p.mask && p.mask[(long)seq * S + key] == 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (pad) sc[t][r] = -INFINITY;
                else if (masked) sc[t][r] = -1e9f;        // masked_fill(mask == 0, -1e9), layers.py:233
            }
        }
        // ---- row softmax: a row's keys sit on the 32 lanes of this half-wave x NT tiles ---------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m = sc[0][r];
#pragma unroll
            for (int t = 1; t < NT; ++t) m = fmaxf(m, sc[t][r]);
            m = wave_half_max(m);
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float e = expf(sc[t][r] - m);
                sc[t][r] = e;
                s += e;
            }
            const float inv = 1.0f / wave_half_sum(s);
#pragma unroll
            for (int t = 0; t < NT; ++t) sc[t][r] *= inv;
        }
        // ---- O = P V: each probability tile re-enters as the A operand through the scratch ------------
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t * 32 < S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * fh) * LDH + fi] = sc[t][r];
                lds_fence();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 pf = *reinterpret_cast<const f32x4*>(&scr[fi * LDH + kk * 8 + fh * 4]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float vv = Vg[(t * 32 + kk * 8 + fh * 4 + u) * LDV + fi];
                        o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[u], vv, o, 0, 0, 0);
                    }
                }
                lds_fence();
            }
        }
        // ---- store: column = head dim on the lane, rows per the 32x32 C layout -----------------------
        if (fi < hd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (qi < S) p.out[((long)seq * S + qi) * p.ldo + head * hd + fi] = o[r];
            }
        }
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
    AttnP p{q, k, v, (long)ld_qkv, key_mask, out, (long)ldo, n_seq, S, n_head, head_dim, scale, n_seq * n_head};
    hipStream_t s = (hipStream_t)stream;
    const int nt = (S + 31) / 32;
    if (nt <= 1) return launch<1>(p, s);
    if (nt <= 2) return launch<2>(p, s);
    if (nt <= 3) return launch<3>(p, s);
    if (nt <= 4) return launch<4>(p, s);
    if (nt <= 8) return launch<8>(p, s);
    return launch<16>(p, s);
}
