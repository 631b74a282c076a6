// lime_token_attention_f32 / lime_token_attention_rows_f32 on the bf16 matrix cores with fp32-level arithmetic (the "split product" of
// gemm_sp_f32.hip): the unmasked shapes of the encoder layers (S = 32, 64 or 128 tokens per sequence, heads 32 columns apart).
//
// Structure as token_attn_f32.hip -- persistent four-wave workgroups over groups of (sequence, head) pairs, K / V of the next group
// prefetched into registers, transposed scores S^T = K Q^T (keys on the MFMA rows, this lane's query on the column) so that a
// probability register IS the B operand of O^T = V^T P^T -- with both products on v_mfma_f32_32x32x16_bf16:
//   * every fp32 operand is three bf16 terms x = hi + mid + lo (exact residuals), a product is six MFMAs (hh, hm, mh, mm, hl, lh;
//     small terms first, fp32 accumulation): the error of one fp32 rounding per product.  A 32 x 32 x 32 block costs 12 MFMAs x 32
//     cycles against 16 x 64 of v_mfma_f32_32x32x2_f32.
//   * K and V are split ONCE per (sequence, head), on their way into LDS (three bf16 images each: K as [key][32 dims], pitch 80 B;
//     V transposed as [dim][key position], pitch 2 S + 16 B -- every fragment is one conflict-free ds_read_b128 per term).
//     Q is split once per 32-query tile, the probabilities (16 registers per 32-key tile) in registers before P V.
//   * the MFMA's k index is a summation label: the 16 keys of a P V step sit in the accumulator layout's order (registers 8 s .. 8 s + 7
//     of a lane = keys 16 s + {0..3, 8..11} + 4 half), and the V image stores its keys in that order (bits 2 and 3 of the key swapped).
// The softmax (log2 domain, v_exp_f32) and the output transpose are those of token_attn_f32.hip.
#include "common.h"
#include "gemm_pp.h"
#include "lds_dma.h"
#include "split_mfma.h"
#include "dropout.h"

using namespace lime_dev;

namespace {

constexpr int KP = 40;                    // K image pitch in bf16 (80 bytes: 16 x an odd number -> conflict-free b128 over 16 lanes)
constexpr int LDO = 33;                   // pitch of the output transpose scratch (floats)
constexpr float LOG2E = 1.4426950408889634f;

struct SpAttnP {
    const float* q; const float* k; const float* v; long ld;
    float* out; long ldo; int n_seq, S, n_head, hd; float scale; int n_pair, n_group;
    const int* row_map; const int* n_seq_dev;
    LimeDropout drop;    // thresh != 0 (the S <= 128 kernel): attention-probability dropout, element (pair, query, key) of the mask
};

using Split = SplitFrag;                            // split_mfma.h
__device__ __forceinline__ Split split8(const float (&x)[8]) { return split_frag(x); }
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, u32x2& h, u32x2& m, u32x2& l) { split_quad(x0, x1, x2, x3, h, m, l); }
__device__ __forceinline__ f32x16 mfma6(const Split& w, const Split& a, f32x16 c) { return split_mfma32(w, a, c); }
__device__ __forceinline__ void lds_fence() {          // (see token_attn_f32.hip: wave-level, vmcnt left alone)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// NT: 32-token tiles per sequence (1, 2, 4); a group is 4 / NT (sequence, head) pairs = 128 key rows.  MAP: operand rows through
// p.row_map (the compacted-sequence variant), the sequence count optionally from device memory.
template <int NT, bool MAP>
__global__ __launch_bounds__(256, 2) void token_attn_sp_kernel(const SpAttnP p) {
    constexpr int G = 4 / NT, SP = NT * 32;
    constexpr int VP = SP + 8;                          // V^T image pitch in bf16 (2 S + 16 bytes: 16 x odd for S = 32, 64, 128)
    constexpr int K_TERM = 128 * KP, V_TERM = G * 32 * VP;
    constexpr bool PREFETCH = NT == 4;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[3 * K_TERM];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[3 * V_TERM];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S;
    int n_pair = p.n_pair, n_group = p.n_group;
    if (p.n_seq_dev) {
        int ns = __builtin_amdgcn_readfirstlane(*p.n_seq_dev);
        ns = ns < p.n_seq ? (ns > 0 ? ns : 0) : p.n_seq;
        n_pair = ns * p.n_head;
        n_group = (n_pair + G - 1) / G;
        if (n_group == 0) return;
    }
    const int krow = 4 * fh;

    // this thread's share of a group's K / V: columns c .. c + 3 of the four rows 4 rg .. 4 rg + 3 of the group's 128
    const int c = (tid & 7) * 4, rg = tid >> 3;
    const int sg = (4 * rg) / SP, sr = (4 * rg) % SP;   // pair inside the group, first row inside the pair
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int group) {
        int pair = group * G + sg;
        pair = pair < n_pair ? pair : n_pair - 1;       // an invalid pair is clamped here and zero-filled in stash()
        const int seq = pair / p.n_head, head = pair - seq * p.n_head;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long row = (long)seq * S + sr + i;
            if constexpr (MAP) row = p.row_map[row];
            const long off = row * p.ld + head * 32 + c;
            kreg[i] = *reinterpret_cast<const f32x4*>(p.k + off);
            vreg[i] = *reinterpret_cast<const f32x4*>(p.v + off);
        }
    };
    auto stash = [&](int group) {
        const bool dead = group * G + sg >= n_pair;
        // K: row R = 4 rg + i, dims c .. c + 3 -> 8 bytes per term
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 kv = kreg[i];
            if (dead) kv = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x2 h, m, l;
            split4(kv[0], kv[1], kv[2], kv[3], h, m, l);
            unsigned short* const d = &Ks[(4 * rg + i) * KP + c];
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + K_TERM) = m;
            *reinterpret_cast<u32x2*>(d + 2 * K_TERM) = l;
        }
        // V^T: dim c + j, keys sr .. sr + 3 at position (sr with bits 2 and 3 swapped) -> 8 bytes per term
        const int pos = (sr & ~12) | ((sr & 4) << 1) | ((sr & 8) >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0 = vreg[0][j], x1 = vreg[1][j], x2 = vreg[2][j], x3 = vreg[3][j];
            if (dead) { x0 = 0.f; x1 = 0.f; x2 = 0.f; x3 = 0.f; }
            u32x2 h, m, l;
            split4(x0, x1, x2, x3, h, m, l);
            unsigned short* const d = &Vs[(sg * 32 + c + j) * VP + pos];
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + V_TERM) = m;
            *reinterpret_cast<u32x2*>(d + 2 * V_TERM) = l;
        }
    };

    const int g = wave / NT, qt = wave % NT;            // this wave's pair inside the group and its 32-query tile
    const unsigned short* const Kg = &Ks[(g * SP + fi) * KP + 8 * fh];
    const unsigned short* const Vg = &Vs[(g * 32 + fi) * VP + 8 * fh];
    float* const scr = &Scr[wave * 32 * LDO];
    const float qscale = p.scale * LOG2E;               // scores in the log2 domain: p = exp2(s' - max')

    // Q fragments straight from global memory: lane (query fi, half fh) holds dims 16 step + 8 fh .. + 7
    f32x4 qraw[4], qnext[4];
    auto q_rows = [&](int grp, f32x4* dst) {
        int pr = grp * G + g;
        pr = pr < n_pair ? pr : n_pair - 1;
        const int sq = pr / p.n_head, hh = pr - sq * p.n_head;
        long qrow = (long)sq * S + qt * 32 + fi;
        if constexpr (MAP) qrow = p.row_map[qrow];
        const float* const qsrc = p.q + qrow * p.ld + hh * 32 + 8 * fh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            dst[2 * s] = *reinterpret_cast<const f32x4*>(qsrc + 16 * s);
            dst[2 * s + 1] = *reinterpret_cast<const f32x4*>(qsrc + 16 * s + 4);
        }
    };

    int group = blockIdx.x;
    if (PREFETCH) { fetch(group); q_rows(group, qnext); }
    for (; group < n_group; group += gridDim.x) {
        if (!PREFETCH) fetch(group);
        stash(group);
        lds_barrier();
        const int pair = group * G + g;
        const bool live = pair < n_pair;
        const int seq = live ? pair / p.n_head : 0, head = live ? pair - seq * p.n_head : 0;
        if (!PREFETCH) { if (live) q_rows(group, qraw); }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) qraw[i] = qnext[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PREFETCH && group + (int)gridDim.x < n_group) {     // the next group's K / V / Q: in flight under the MFMAs
            fetch(group + gridDim.x);
            q_rows(group + gridDim.x, qnext);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
            Split qs[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 a = qraw[2 * s] * qscale, b = qraw[2 * s + 1] * qscale;
                const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                qs[s] = split8(x);
            }
            // ---- S^T = K Q^T: keys on rows, this lane's query on the column ---------------------------------------------
            f32x16 sc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned short* const kp = Kg + t * 32 * KP + 16 * s;
                    Split ks;
                    ks.h = *reinterpret_cast<const bf16x8*>(kp);
                    ks.m = *reinterpret_cast<const bf16x8*>(kp + K_TERM);
                    ks.l = *reinterpret_cast<const bf16x8*>(kp + 2 * K_TERM);
                    sc[t] = mfma6(ks, qs[s], sc[t]);
                }
            }
            // ---- softmax over the keys of this lane's query: own registers, then the other half-wave --------------------
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
                    const float e = __builtin_amdgcn_exp2f(sc[t][r] - m);
                    sc[t][r] = e;
                    sum += e;
                }
            }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            if (p.drop.thresh != 0) {
                // nn.MultiheadAttention's dropout on the probabilities (newsEncoders.py:244-247, training mode): keep / (1 - p) goes
                // onto the unnormalised e (the sum above is over all of them).  Registers 4 g .. 4 g + 3 of a tile are four consecutive
                // keys: one hash each, element index (pair S + query) S + key as the backward regenerates it.
                const uint64_t mrow = ((uint64_t)pair * (uint64_t)S + (uint64_t)(qt * 32 + fi)) * (uint64_t)S;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const unsigned keep = lime_keep4(p.drop, (mrow + (uint64_t)(32 * t + 8 * gq + krow)) >> 2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sc[t][4 * gq + e] = (keep >> e) & 1u ? sc[t][4 * gq + e] * p.drop.scale : 0.f;
                    }
            }
            // ---- O^T = V^T P^T: registers 8 s .. 8 s + 7 of a probability tile are the B operand of step s ----------------
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float x[8] = {sc[t][8 * s], sc[t][8 * s + 1], sc[t][8 * s + 2], sc[t][8 * s + 3],
                                        sc[t][8 * s + 4], sc[t][8 * s + 5], sc[t][8 * s + 6], sc[t][8 * s + 7]};
                    const Split ps = split8(x);
                    const unsigned short* const vp = Vg + t * 32 + 16 * s;
                    Split vs;
                    vs.h = *reinterpret_cast<const bf16x8*>(vp);
                    vs.m = *reinterpret_cast<const bf16x8*>(vp + V_TERM);
                    vs.l = *reinterpret_cast<const bf16x8*>(vp + 2 * V_TERM);
                    o = mfma6(vs, ps, o);
                }
            }
            // ---- transpose [head dim][query] -> [query][head dim] through the scratch, store whole head rows --------------
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[fi * LDO + (r & 3) + 8 * (r >> 2) + krow] = o[r] * inv;
            lds_fence();
            if (fi < p.hd) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int row = it * 2 + fh;
                    p.out[((long)seq * S + qt * 32 + row) * p.ldo + head * p.hd + fi] = scr[row * LDO + fi];
                }
            }
            lds_fence();
        }
        lds_barrier();                                   // everyone is done with this group's LDS images
    }
}

// Long sequences (S = 256 or 512: the 512-token bodies of BASELINE configs[3]): a task is one (sequence, head) pair x one block of
// 128 queries (a wave per 32-query tile, its Q split once); the keys go through LDS in blocks of 128 -- K / V images as above, the
// next block's rows prefetched into registers -- with a running maximum: o and the probability sum are rescaled by 2^(m_old - m_new)
// when a block raises the maximum.  Optionally writes the log2-domain log-sum-exp of every query (p.lse: what the blocked backward
// needs, lime_token_attention_lse_f32).
template <bool MAP>
__global__ __launch_bounds__(256, 2) void token_attn_sp_long_kernel(const SpAttnP p, float* __restrict__ lse) {
    constexpr int SP = 128, VP = SP + 8;
    constexpr int K_TERM = 128 * KP, V_TERM = 32 * VP;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[3 * K_TERM];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[3 * V_TERM];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S, n_blk = S / 128;
    int n_pair = p.n_pair;
    if (p.n_seq_dev) {
        int ns = __builtin_amdgcn_readfirstlane(*p.n_seq_dev);
        ns = ns < p.n_seq ? (ns > 0 ? ns : 0) : p.n_seq;
        n_pair = ns * p.n_head;
    }
    const int n_task = n_pair * n_blk;                  // task = pair * n_blk + query block
    if (n_task == 0) return;
    const int krow = 4 * fh;
    const int c = (tid & 7) * 4, rg = tid >> 3, sr = 4 * rg;
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int task, int kb) {                // key block kb of the task's pair: rows 128 kb + sr .. + 3, columns c .. c + 3
        const int pair = task / n_blk;
        const int seq = pair / p.n_head, head = pair - seq * p.n_head;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long row = (long)seq * S + kb * 128 + sr + i;
            if constexpr (MAP) row = p.row_map[row];
            const long off = row * p.ld + head * 32 + c;
            kreg[i] = *reinterpret_cast<const f32x4*>(p.k + off);
            vreg[i] = *reinterpret_cast<const f32x4*>(p.v + off);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 h, m, l;
            split4(kreg[i][0], kreg[i][1], kreg[i][2], kreg[i][3], h, m, l);
            unsigned short* const d = &Ks[(sr + i) * KP + c];
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + K_TERM) = m;
            *reinterpret_cast<u32x2*>(d + 2 * K_TERM) = l;
        }
        const int pos = (sr & ~12) | ((sr & 4) << 1) | ((sr & 8) >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x2 h, m, l;
            split4(vreg[0][j], vreg[1][j], vreg[2][j], vreg[3][j], h, m, l);
            unsigned short* const d = &Vs[(c + j) * VP + pos];
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + V_TERM) = m;
            *reinterpret_cast<u32x2*>(d + 2 * V_TERM) = l;
        }
    };
    const unsigned short* const Kg = &Ks[fi * KP + 8 * fh];
    const unsigned short* const Vg = &Vs[fi * VP + 8 * fh];
    float* const scr = &Scr[wave * 32 * LDO];
    const float qscale = p.scale * LOG2E;

    int task = blockIdx.x;
    if (task < n_task) fetch(task, 0);
    for (; task < n_task; task += gridDim.x) {
        const int pair = task / n_blk, qb = task - pair * n_blk;
        const int seq = pair / p.n_head, head = pair - seq * p.n_head;
        const int q0 = qb * 128 + wave * 32;            // this wave's 32 queries
        Split qs[2];
        {
            long qrow = (long)seq * S + q0 + fi;
            if constexpr (MAP) qrow = p.row_map[qrow];
            const float* const qsrc = p.q + qrow * p.ld + head * 32 + 8 * fh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(qsrc + 16 * s) * qscale, b = *reinterpret_cast<const f32x4*>(qsrc + 16 * s + 4) * qscale;
                const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                qs[s] = split8(x);
            }
        }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float m = -INFINITY, sum = 0.f;
        for (int kb = 0; kb < n_blk; ++kb) {
            stash();
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < n_blk) fetch(task, kb + 1);                     // the next key block (or the next task's first): in flight under the MFMAs
            else if (task + (int)gridDim.x < n_task) fetch(task + gridDim.x, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned short* const kp = Kg + t * 32 * KP + 16 * s;
                    Split ks;
                    ks.h = *reinterpret_cast<const bf16x8*>(kp);
                    ks.m = *reinterpret_cast<const bf16x8*>(kp + K_TERM);
                    ks.l = *reinterpret_cast<const bf16x8*>(kp + 2 * K_TERM);
                    sc[t] = mfma6(ks, qs[s], sc[t]);
                }
            }
            float mc = sc[0][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mc = fmaxf(mc, sc[t][r]);
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            const float m_new = fmaxf(m, mc);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);       // 0 on the first block (m = -inf), 1 when the maximum stands
            sum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sc[t][r] - m_new);
                    sc[t][r] = e;
                    sum += e;
                }
            }
            m = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float x[8] = {sc[t][8 * s], sc[t][8 * s + 1], sc[t][8 * s + 2], sc[t][8 * s + 3],
                                        sc[t][8 * s + 4], sc[t][8 * s + 5], sc[t][8 * s + 6], sc[t][8 * s + 7]};
                    const Split ps = split8(x);
                    const unsigned short* const vp = Vg + t * 32 + 16 * s;
                    Split vs;
                    vs.h = *reinterpret_cast<const bf16x8*>(vp);
                    vs.m = *reinterpret_cast<const bf16x8*>(vp + V_TERM);
                    vs.l = *reinterpret_cast<const bf16x8*>(vp + 2 * V_TERM);
                    o = mfma6(vs, ps, o);
                }
            }
            lds_barrier();                               // everyone is done with this block's images
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        if (lse && fh == 0) lse[((long)seq * S + q0 + fi) * p.n_head + head] = m + log2f(sum);
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[fi * LDO + (r & 3) + 8 * (r >> 2) + krow] = o[r] * inv;
        lds_fence();
        if (fi < p.hd) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 2 + fh;
                p.out[((long)seq * S + q0 + row) * p.ldo + head * p.hd + fi] = scr[row * LDO + fi];
            }
        }
        lds_fence();
    }
}

int sp_attn_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int NT>
int launch(SpAttnP p, hipStream_t s) {
    constexpr int G = 4 / NT;
    p.n_group = (p.n_pair + G - 1) / G;
    long blocks = (long)sp_attn_cus() * 2;
    if (blocks > p.n_group) blocks = p.n_group;
    if (p.row_map) hipLaunchKernelGGL((token_attn_sp_kernel<NT, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((token_attn_sp_kernel<NT, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return lime_check_launch("token_attn_sp_kernel");
}

}  // namespace

// LIME_OK / error: launched; LIME_PP_NOT_APPLICABLE: the caller takes token_attn_f32.hip's kernels (masks, other lengths, unpadded
// heads, the split product switched off).  lse (optional, S = 256 / 512 only): [tokens, n_head] log2-domain log-sum-exp.
int lime_token_attention_sp(const float* q, const float* k, const float* v, long ld, const int* row_map, const int* n_seq_dev,
                            float* out, long ldo, int n_seq, int S, int n_head, int hd, float scale, float* lse, hipStream_t s,
                            const LimeDropout* drop) {
    if (!(lime_split_mode() & 1)) return LIME_PP_NOT_APPLICABLE;
    const bool is_long = S == 256 || S == 512;
    if (!(S == 32 || S == 64 || S == 128 || is_long)) return LIME_PP_NOT_APPLICABLE;
    if (lse && !is_long) return LIME_PP_NOT_APPLICABLE;
    if (drop && drop->thresh != 0 && (is_long || row_map)) return LIME_PP_NOT_APPLICABLE;       // probability dropout: the S <= 128 kernel only
    if (ld % 4 != 0 || ld < (long)n_head * 32 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16) != 0 || hd > 32) return LIME_PP_NOT_APPLICABLE;
    SpAttnP p{q, k, v, ld, out, ldo, n_seq, S, n_head, hd, scale, n_seq * n_head, 0, row_map, n_seq_dev, drop ? *drop : LimeDropout{0, 0, 1.f}};
    if (is_long) {
        const long n_task = (long)p.n_pair * (S / 128);
        long blocks = (long)sp_attn_cus() * 2;
        if (blocks > n_task) blocks = n_task;
        if (row_map) hipLaunchKernelGGL((token_attn_sp_long_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, p, lse);
        else hipLaunchKernelGGL((token_attn_sp_long_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, p, lse);
        return lime_check_launch("token_attn_sp_long_kernel");
    }
    switch (S / 32) {
        case 1: return launch<1>(p, s);
        case 2: return launch<2>(p, s);
        default: return launch<4>(p, s);
    }
}
