// lime_token_attention_bf16: softmax(Q K^T * scale) V per (sequence, head) on the bf16 matrix cores
// (BASELINE config 3: "bf16 (MFMA token-attention path)", fp32 accumulate and softmax).
//
// Same decomposition as the fp32 kernel (token_attn_f32.hip): a wave owns 32 queries and computes the TRANSPOSED score
// strip S^T = K Q^T (keys on the MFMA rows, the query on the lane), so the softmax max / sum is a reduction over a lane's
// own registers plus one cross-half shuffle, and the probability registers feed O^T = V^T P^T without leaving the
// register file.  What changes with v_mfma_f32_32x32x16_bf16 (8 bf16 per lane and operand, k slot (kg, e) = 8 kg + e):
//   * K rows are 64 bytes (32 bf16, heads padded to 32 columns): one ds_read_b128 per 32-key tile and MFMA, from an
//     image with an 80-byte row pitch (conflict-free for 16-lane read phases); Q fragments are two 16-byte global loads.
//   * the score registers of a tile, r = 8 m + e, hold keys 32 t + 16 m + 8 (e >> 2) + 4 kg + (e & 3): the k index of an
//     MFMA is only a summation label, so P is packed to bf16 in exactly that order and V^T is read in the same order --
//     two ds_read_b64 (keys 16 m + 4 kg .. + 3 and + 8) from the transposed image Vt[d][key].
//   * scores are scaled after the MFMA (fp32), so Q is not re-rounded.
// Unmasked, S in {32, 64, 128} (the encoder-layer shapes); longer sequences take the fp32-core variant in
// token_attn_f32.hip.  Workgroups are persistent with the next group's K / V / Q prefetched into registers.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KP = 40;    // pitch of K rows in LDS (bf16): 80 bytes, conflict-free ds_read_b128
constexpr int LDO = 33;   // pitch of the output transpose scratch (floats)
constexpr float LOG2E = 1.4426950408889634f;

struct AttnB {
    const unsigned short* q; const unsigned short* k; const unsigned short* v; long ld;
    unsigned short* out; long ldo; int n_seq, S, n_head, hd; float scale; int n_pair, n_group, out_pad;
    const int* row_map;      // MAP: q / k / v row of token (seq * S + t) is row_map[seq * S + t] (lime_compact_sequences)
    const int* n_seq_dev;    // MAP: optional device-side sequence count
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {          // two floats -> two bf16 (round to nearest even): one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

template <int NT, bool MAP = false>
__global__ __launch_bounds__(256, (NT <= 2) ? 4 : 3) void token_attn_bf16_kernel(const AttnB p) {
    constexpr int G = 4 / NT;                      // (sequence, head) pairs per group: 4, 2, 1
    constexpr int SP = NT * 32;                    // sequence length
    constexpr int VP = SP + 4;                     // pitch of Vt rows (bf16): rows 2 banks apart
    constexpr int NLD = G * SP * 4 / 256;          // 16-byte loads per thread and operand (= 2)
    __shared__ __attribute__((aligned(16))) unsigned short Kb[G * SP * KP];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[G * 32 * VP];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S, hd = p.hd;
    int n_pair = p.n_pair, n_group = p.n_group;
    if constexpr (MAP) {
        if (p.n_seq_dev) {
            int ns = __builtin_amdgcn_readfirstlane(*p.n_seq_dev);
            ns = ns < p.n_seq ? (ns > 0 ? ns : 0) : p.n_seq;
            n_pair = ns * p.n_head;
            n_group = (n_pair + 4 / NT - 1) / (4 / NT);
        }
        if (n_group == 0) return;
    }
    // wave -> (pair of the group, query tile): NT waves per pair, one 32-query tile each
    const int g = wave / NT, qt = wave % NT;

    u32x4 kreg[NLD], vreg[NLD], qreg[2], qnext[2];
    auto fetch = [&](int group) {                  // raw loads only (an invalid pair is clamped, zero-filled in stash)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            const int c = (e & 3) * 8, r = (e >> 2) % SP, gg = (e >> 2) / SP;
            int pair = group * G + gg;
            pair = pair < n_pair ? pair : n_pair - 1;
            const int seq = pair / p.n_head, head = pair - seq * p.n_head;
            long row = (long)seq * S + r;
            if constexpr (MAP) row = p.row_map[row];
            const long off = row * p.ld + head * 32 + c;
            kreg[i] = *reinterpret_cast<const u32x4*>(p.k + off);
            vreg[i] = *reinterpret_cast<const u32x4*>(p.v + off);
        }
        int pr = group * G + g;
        pr = pr < n_pair ? pr : n_pair - 1;
        const int sq = pr / p.n_head, hh = pr - sq * p.n_head;
        long qrow = (long)sq * S + qt * 32 + fi;
        if constexpr (MAP) qrow = p.row_map[qrow];
        const unsigned short* qsrc = p.q + qrow * p.ld + hh * 32 + 8 * fh;
        qnext[0] = *reinterpret_cast<const u32x4*>(qsrc);              // d = 8 fh .. + 7
        qnext[1] = *reinterpret_cast<const u32x4*>(qsrc + 16);         // d = 16 + 8 fh .. + 7
    };
    auto stash = [&](int group) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            const int c = (e & 3) * 8, r = (e >> 2) % SP, gg = (e >> 2) / SP;
            u32x4 kb = kreg[i], vb = vreg[i];
            if (group * G + gg >= n_pair) { kb = u32x4{0u, 0u, 0u, 0u}; vb = kb; }
            *reinterpret_cast<u32x4*>(&Kb[(gg * SP + r) * KP + c]) = kb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w = vb[j >> 1];
                Vt[(gg * 32 + c + j) * VP + r] = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xFFFFu));
            }
        }
    };

    const unsigned short* Kg = &Kb[g * SP * KP];
    const unsigned short* Vg = &Vt[g * 32 * VP];
    float* scr = &Scr[wave * 32 * LDO];
    const float qscale = p.scale * LOG2E;          // scores in the log2 domain: p = exp2(s' - max')

    int group = blockIdx.x;
    fetch(group);
    for (; group < n_group; group += gridDim.x) {
        stash(group);
        qreg[0] = qnext[0];
        qreg[1] = qnext[1];
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (group + (int)gridDim.x < n_group) fetch(group + gridDim.x);       // in flight under this group's work
        __builtin_amdgcn_sched_barrier(0);
        const int pair = group * G + g;
        if (pair < n_pair) {
            const int seq = pair / p.n_head, head = pair - seq * p.n_head;
            // ---- S^T = K Q^T ---------------------------------------------------------------------------------------
            f32x16 sc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(&Kg[(t * 32 + fi) * KP + 16 * m + 8 * fh]);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qreg[m]),
                                                                    sc[t], 0, 0, 0);
                }
            }
            // ---- softmax over the keys of this lane's query (fp32) --------------------------------------------------
            float mx = sc[0][0];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32)) * qscale;                  // scale > 0: max commutes with the scaling
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sc[t][r] * qscale - mx);
                    sc[t][r] = e;
                    sum += e;
                }
            }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            // ---- O^T = V^T P^T: P packed to bf16 in register order, V^T read in the same key order --------------------
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    u32x4 pb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pb[j] = pack2(sc[t][8 * m + 2 * j], sc[t][8 * m + 2 * j + 1]);
                    const unsigned short* vrow = &Vg[fi * VP + t * 32 + 16 * m + 4 * fh];
                    const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);          // keys 16 m + 4 fh + 0..3
                    const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 8);      // keys 16 m + 8 + 4 fh + 0..3
                    const u32x4 va = {v0[0], v0[1], v1[0], v1[1]};
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb), o, 0, 0, 0);
                }
            }
            // ---- transpose [head dim][query] -> [query][head dim] through the scratch, bf16 stores --------------------
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[fi * LDO + (r & 3) + 8 * (r >> 2) + 4 * fh] = o[r] * inv;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // lane -> (query row pair, two head-dim columns): 4-byte stores of two bf16
            {
                const int cpair = fi & 15;                                // columns 2 cpair, 2 cpair + 1
                const int rsel = (fi >> 4) + 2 * fh;                      // rows rsel, rsel + 4, ...
                if (2 * cpair < hd) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = it * 4 + rsel;
                        const unsigned w = pack2(scr[row * LDO + 2 * cpair], scr[row * LDO + 2 * cpair + 1]);
                        unsigned short* dst = p.out + ((long)seq * S + qt * 32 + row) * p.ldo + head * hd + 2 * cpair;
                        if (2 * cpair + 1 < hd) *reinterpret_cast<unsigned*>(dst) = w;
                        else *dst = (unsigned short)(w & 0xFFFFu);
                    }
                }
                if (head == p.n_head - 1 && fi < p.out_pad) {             // zero columns behind the last head
#pragma unroll
                    for (int it = 0; it < 16; ++it)
                        p.out[((long)seq * S + qt * 32 + it * 2 + fh) * p.ldo + p.n_head * hd + fi] = (unsigned short)0;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        lds_barrier();                           // everyone is done with this group's LDS images
    }
}

int attn_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int NT, bool MAP = false>
int launch(AttnB p, hipStream_t s) {
    constexpr int G = 4 / NT;
    p.n_group = (p.n_pair + G - 1) / G;
    long blocks = (long)attn_cus() * (NT <= 2 ? 4 : 3);            // 36 KB LDS, <= 128 / 170 VGPRs: 4 / 3 workgroups per CU
    if (blocks > p.n_group) blocks = p.n_group;
    hipLaunchKernelGGL((token_attn_bf16_kernel<NT, MAP>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return lime_check_launch("lime_token_attention_bf16");
}

}  // namespace

// S = 32 / 64 / 128 with an even head_dim: the bf16-MFMA kernel; returns 1 when the shape is not its (the caller in
// token_attn_f32.hip then takes the fp32-core variant).
int lime_token_attention_bf16_mfma(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, uint16_t* out, int64_t ldo,
                                   int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, float scale, int32_t out_pad,
                                   hipStream_t s) {
    if (!(S == 32 || S == 64 || S == 128) || scale <= 0.f) return 1;
    if ((uintptr_t)out % 4 != 0 || ldo % 2 != 0 || head_dim % 2 != 0) return 1;        // 4-byte output stores
    AttnB p{q, k, v, (long)ld_qkv, out, (long)ldo, n_seq, S, n_head, head_dim, scale, n_seq * n_head, 0, out_pad, nullptr, nullptr};
    switch (S / 32) {
        case 1: return launch<1>(p, s);
        case 2: return launch<2>(p, s);
        default: return launch<4>(p, s);
    }
}

extern "C" int lime_token_attention_rows_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, const int32_t* row_map,
                                              const int32_t* n_seq_dev, uint16_t* out, int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head,
                                              int32_t head_dim, float scale, int32_t out_cols, void* stream) {
    LIME_REQUIRE(q && k && v && out && row_map, LIME_ERR_BAD_ARG, "lime_token_attention_rows_bf16: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && n_head > 0 && head_dim > 0 && head_dim <= 32 && head_dim % 2 == 0 && scale > 0.f, LIME_ERR_BAD_ARG,
                 "lime_token_attention_rows_bf16: bad dims n_seq=%d n_head=%d head_dim=%d", n_seq, n_head, head_dim);
    LIME_REQUIRE(S == 32 || S == 64 || S == 128, LIME_ERR_UNSUPPORTED, "lime_token_attention_rows_bf16: S must be 32, 64 or 128 (got %d)", S);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * 32 && ld_qkv % 8 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0),
                 LIME_ERR_BAD_ARG, "lime_token_attention_rows_bf16: heads are 32 bf16 columns apart, rows 16-byte aligned");
    const int pad = out_cols - n_head * head_dim;
    LIME_REQUIRE(pad >= 0 && pad <= 32 && ldo >= out_cols && (uintptr_t)out % 4 == 0 && ldo % 2 == 0, LIME_ERR_BAD_ARG,
                 "lime_token_attention_rows_bf16: out_cols must be in [n_head * head_dim, n_head * head_dim + 32] and <= ldo (even)");
    if (n_seq == 0) return LIME_OK;
    AttnB p{q, k, v, (long)ld_qkv, out, (long)ldo, n_seq, S, n_head, head_dim, scale, n_seq * n_head, 0, pad, row_map, n_seq_dev};
    hipStream_t s = (hipStream_t)stream;
    switch (S / 32) {
        case 1: return launch<1, true>(p, s);
        case 2: return launch<2, true>(p, s);
        default: return launch<4, true>(p, s);
    }
}
