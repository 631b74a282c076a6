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
//     so P never leaves the register file; V is staged transposed so that one conflict-free ds_read_b128
//     supplies the A operands of four consecutive MFMAs;
//   * 1/sum is applied to the 16 output registers, not to the S probabilities.
// The output tile O^T (head dim on rows, query on lanes) is transposed through a 4 KiB per-wave LDS
// scratch so the store writes whole 120-byte head rows.  Workgroups are persistent: K and V of the next
// (sequence, head) group are prefetched into registers (8-byte loads) while the current one is computed from
// LDS; Q fragments come straight from global memory in MFMA operand form; short sequences pack 4 (S <= 32)
// or 2 (S <= 64) pairs per group.  Scores live in the log2 domain so the exponential is one v_exp_f32.
#include "common.h"
#include "gemm_pp.h"
#include <type_traits>

#ifdef LIME_STAMPS
// Diagnostic build only (tools/attn_stamps.py): per-wave s_memtime sums of the loop segments; never in liblime_hip.so.
static unsigned long long* g_attn_stamp_buf = nullptr;
extern "C" void lime_debug_set_attn_stamp_buffer(unsigned long long* p) { g_attn_stamp_buf = p; }
#define ASTAMP(i)                                                           \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 \
        tsum[i] += t_ - tlast;                                              \
        tlast = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
    }
#else
#define ASTAMP(i)
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 36;  // pitch of K rows in LDS (floats): conflict-free ds_read_b128
// V is staged TRANSPOSED, Vt[head dim][key] with pitch SP + 4: accumulator registers 4g .. 4g+3 of a probability tile
// are keys 32t + 8g + 4*half + 0..3, so the matching A operands of four consecutive PV MFMAs are one ds_read_b128
constexpr int LDO = 33;  // pitch of the output transpose scratch
constexpr float LOG2E = 1.4426950408889634f;

struct AttnP {
    const float* q; const float* k; const float* v; long ld; const unsigned char* mask;
    float* out; long ldo; int n_seq, S, n_head, hd, hs; float scale; int n_pair; int vec2; int n_group;
    int out_pad;         // BF: zero columns written behind the last head (the next GEMM reads K rounded up to 8)
    const int* row_map;  // MAP: q / k / v row of token (seq * S + t) is row_map[seq * S + t] (padding tokens share S table rows)
    const int* n_seq_dev;  // optional device-side sequence count (min(*n_seq_dev, n_seq) sequences are computed)
    float* lse;          // optional [tokens, n_head]: log2-domain log-sum-exp of every query's scaled scores (what the blocked backward
                         // otherwise recomputes with a Q K^T pass of its own)
#ifdef LIME_STAMPS
    unsigned long long* stamps;
#endif
};

__device__ __forceinline__ void lds_fence() {
    // LDS ops of one wave execute in order; drain lgkmcnt so that a tile written by some lanes is read back by
    // others, and keep the compiler from moving memory ops across the point.  Deliberately NOT a workgroup-scope
    // fence: that also waits vmcnt(0), i.e. for the K / V prefetch and the output stores still in flight.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Two floats of row `r`, columns c, c+1 of a (sequence, head) operand; zero outside [0, rows) x [0, hd) (general path).
__device__ __forceinline__ f32x2 load2(const float* src, long ld, int r, int c, int rows, int hd, bool vec2) {
    f32x2 v = {0.f, 0.f};
    if (r < rows) {
        const float* p = src + (long)r * ld + c;
        if (vec2) {
            if (c < hd) v = *reinterpret_cast<const f32x2*>(p);          // hd even: c + 1 < hd as well
        } else {
            if (c < hd) v[0] = p[0];
            if (c + 1 < hd) v[1] = p[1];
        }
    }
    return v;
}

// A workgroup walks groups of G (sequence, head) pairs: K / V of the next group are prefetched into registers while
// the current group is computed from LDS (the staging latency of a one-shot workgroup was half its life).
// FAST: S == NT * 32, no key mask, 8-byte loads -- no flag pass, no bounds branches (the title / body shapes of the
// encoder layers); otherwise the general path with per-key flags.  The score strip never crosses an if-merge: hipcc
// copies the whole accumulator array at a merge.
// BF (FAST only): q / k / v / out are bf16 (lime_token_attention_bf16); they are widened to fp32 on the way into LDS /
// the fragments and the products stay on the exact-fp32 MFMA, so scores, softmax and P V are computed as in the fp32 path.
// MAP (FAST, fp32 only): the compacted-sequence variant (lime_token_attention_rows_f32) -- operand rows are looked up in
// p.row_map (one dependent 4-byte load per 16-byte operand load; it rides in the same prefetch), the sequence count may come
// from device memory.  Output rows stay dense (seq * S + t).
template <int NT, bool FAST, bool BF = false, bool MAP = false>
__global__ __launch_bounds__(256, (NT <= 2) ? 3 : (NT <= 4 ? 2 : 1)) void token_attn_kernel(const AttnP p) {
    static_assert(!BF || FAST, "the bf16 variant exists for the FAST shapes only");
    static_assert(!MAP || (FAST && !BF), "the row-map variant exists for the fp32 FAST shapes only");
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);   // (sequence, head) pairs per group
    constexpr int WPP = 4 / G;                     // waves per pair
    constexpr int SP = NT * 32;                    // padded sequence length
    constexpr int NLD = BF ? G * SP * 4 / 256 : (FAST ? G * SP * 8 / 256 : G * SP * 16 / 256);   // 16-byte (FAST) / 8-byte loads per thread and operand
    constexpr bool PREFETCH = NT == 3 || NT == 4;  // short sequences are latency-bound either way; long ones need the registers
    __shared__ __attribute__((aligned(16))) float Ks[G * SP * LDH];
    constexpr int LDVT = SP + 4;                   // pitch of Vt rows (keys of one head dim)
    __shared__ __attribute__((aligned(16))) float Vs[G * 32 * LDVT];
    __shared__ __attribute__((aligned(16))) float Scr[4 * 32 * LDO];
    __shared__ float Flag[G * SP];                // 0: key takes part, 1: masked (-1e9), 2: padding (-inf)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int S = p.S, hd = p.hd, hs = p.hs;
    int n_pair = p.n_pair, n_group = p.n_group;
    if (p.n_seq_dev) {                                 // device-side sequence count (compacted batches; uniform scalar load)
        int ns = __builtin_amdgcn_readfirstlane(*p.n_seq_dev);
        ns = ns < p.n_seq ? (ns > 0 ? ns : 0) : p.n_seq;
        n_pair = ns * p.n_head;
        n_group = (n_pair + (NT >= 3 ? 1 : 4 / NT) - 1) / (NT >= 3 ? 1 : 4 / NT);
        if (n_group == 0) return;
    }
    const bool vec2 = p.vec2 != 0;
    const int krow = 4 * fh;                       // key row of accumulator register r: (r & 3) + 8 * (r >> 2) + 4 * fh

    typedef typename std::conditional<FAST, f32x4, f32x2>::type ld_t;
    ld_t kreg[NLD], vreg[NLD];
    // FAST (heads padded to 32 columns, 16-byte aligned; S == NT * 32): unconditional 16-byte loads that carry no
    // arithmetic -- an invalid pair is clamped to the last pair and zero-filled in stash().  A select right behind each
    // load made hipcc wait for every load before issuing the next one (s_memtime stamps: 58 % of a wave's life was this
    // "prefetch", 16 serialized round trips).
    auto fetch = [&](int group) {                  // this thread's share of the group's K / V -> registers
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            if constexpr (BF) {                    // 8 bf16 = 16 bytes per load: four loads cover a 32-column head row
                const int c = (e & 3) * 8, r = (e >> 2) % SP, g = (e >> 2) / SP;
                int pair = group * G + g;
                pair = pair < n_pair ? pair : n_pair - 1;
                const int seq = pair / p.n_head, head = pair - seq * p.n_head;
                const long off = ((long)seq * S + r) * p.ld + head * hs + c;
                kreg[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(p.k) + off);
                vreg[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(p.v) + off);
            } else if constexpr (FAST) {
                const int c = (e & 7) * 4, r = (e >> 3) % SP, g = (e >> 3) / SP;
                int pair = group * G + g;
                pair = pair < n_pair ? pair : n_pair - 1;
                const int seq = pair / p.n_head, head = pair - seq * p.n_head;
                long row = (long)seq * S + r;
                if constexpr (MAP) row = p.row_map[row];
                const long off = row * p.ld + head * hs + c;
                kreg[i] = *reinterpret_cast<const f32x4*>(p.k + off);
                vreg[i] = *reinterpret_cast<const f32x4*>(p.v + off);
            } else {
                const int c = (e & 15) * 2, r = (e >> 4) % SP, g = (e >> 4) / SP;
                const int pair = group * G + g;
                f32x2 kv = {0.f, 0.f}, vv = {0.f, 0.f};
                if (pair < n_pair) {
                    const int seq = pair / p.n_head, head = pair - seq * p.n_head;
                    const long base = (long)seq * S * p.ld + head * hs;
                    kv = load2(p.k + base, p.ld, r, c, S, hd, vec2);
                    vv = load2(p.v + base, p.ld, r, c, S, hd, vec2);
                }
                kreg[i] = kv;
                vreg[i] = vv;
            }
        }
    };
    auto stash = [&](int group) {                  // registers -> LDS images (+ key flags)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            if constexpr (BF) {
                const int c = (e & 3) * 8, r = (e >> 2) % SP, g = (e >> 2) / SP;
                u32x4 kb = __builtin_bit_cast(u32x4, kreg[i]), vb = __builtin_bit_cast(u32x4, vreg[i]);
                if (group * G + g >= n_pair) { kb = u32x4{0u, 0u, 0u, 0u}; vb = kb; }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    f32x4 kf;
                    kf[0] = __builtin_bit_cast(float, kb[2 * h2] << 16);
                    kf[1] = __builtin_bit_cast(float, kb[2 * h2] & 0xFFFF0000u);
                    kf[2] = __builtin_bit_cast(float, kb[2 * h2 + 1] << 16);
                    kf[3] = __builtin_bit_cast(float, kb[2 * h2 + 1] & 0xFFFF0000u);
                    *reinterpret_cast<f32x4*>(&Ks[(g * SP + r) * LDH + c + 4 * h2]) = kf;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned w = vb[j >> 1];
                    Vs[(g * 32 + c + j) * LDVT + r] = __builtin_bit_cast(float, (j & 1) ? (w & 0xFFFF0000u) : (w << 16));
                }
            } else if constexpr (FAST) {
                const int c = (e & 7) * 4, r = (e >> 3) % SP, g = (e >> 3) / SP;
                f32x4 kv = kreg[i], vv = vreg[i];
                if (group * G + g >= n_pair) { kv = f32x4{0.f, 0.f, 0.f, 0.f}; vv = kv; }
                *reinterpret_cast<f32x4*>(&Ks[(g * SP + r) * LDH + c]) = kv;
#pragma unroll
                for (int j = 0; j < 4; ++j) Vs[(g * 32 + c + j) * LDVT + r] = vv[j];
            } else {
                const int c = (e & 15) * 2, r = (e >> 4) % SP, g = (e >> 4) / SP;
                *reinterpret_cast<f32x2*>(&Ks[(g * SP + r) * LDH + c]) = kreg[i];
                Vs[(g * 32 + c) * LDVT + r] = vreg[i][0];
                Vs[(g * 32 + c + 1) * LDVT + r] = vreg[i][1];
            }
        }
        if constexpr (!FAST) {
            for (int e = tid; e < G * SP; e += 256) {
                const int key = e % SP, pair = group * G + e / SP;
                float f = key >= S ? 2.f : 0.f;
                if (f == 0.f && p.mask && pair < n_pair && p.mask[(long)(pair / p.n_head) * S + key] == 0) f = 1.f;
                Flag[e] = f;
            }
        }
    };

    const int g = wave / WPP;
    const float* Kg = &Ks[g * SP * LDH];
    const float* Vg = &Vs[g * 32 * LDVT];
    const float* Fg = &Flag[g * SP];
    float* scr = &Scr[wave * 32 * LDO];
    const float qscale = p.scale * LOG2E;          // scores in the log2 domain: p = exp2(s' - max')
    const float masked = -1e9f * LOG2E;

    // Q fragments straight from global memory: lane (query fi, half fh) needs Q[q][8kk + 4fh .. +3]
    // FAST: four 16-byte loads (the pad columns are real zeros); general: eight 8-byte / scalar loads
    typedef typename std::conditional<FAST, f32x4, f32x2>::type q_t;
    constexpr int NQ = FAST ? 4 : 8;
    q_t qraw[NQ], qnext[NQ];
    auto q_rows = [&](int grp, int qt, q_t* dst) {
        int pr = grp * G + g;
        if constexpr (BF) {                           // four 8-byte loads of 4 bf16, widened
            pr = pr < n_pair ? pr : n_pair - 1;
            const int sq = pr / p.n_head, hh = pr - sq * p.n_head;
            const unsigned short* qsrc = reinterpret_cast<const unsigned short*>(p.q) + ((long)sq * S + qt * 32 + fi) * p.ld + hh * hs + fh * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint2 w = *reinterpret_cast<const uint2*>(qsrc + kk * 8);
                f32x4 f;
                f[0] = __builtin_bit_cast(float, w.x << 16); f[1] = __builtin_bit_cast(float, w.x & 0xFFFF0000u);
                f[2] = __builtin_bit_cast(float, w.y << 16); f[3] = __builtin_bit_cast(float, w.y & 0xFFFF0000u);
                dst[kk] = f;
            }
        } else if constexpr (FAST) {                  // S == NT * 32: every query row exists
            pr = pr < n_pair ? pr : n_pair - 1;
            const int sq = pr / p.n_head, hh = pr - sq * p.n_head;
            long qrow = (long)sq * S + qt * 32 + fi;
            if constexpr (MAP) qrow = p.row_map[qrow];
            const float* qsrc = p.q + qrow * p.ld + hh * hs + fh * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const f32x4*>(qsrc + kk * 8);
        } else {
            if (pr >= n_pair || qt * 32 >= S) return;
            const int sq = pr / p.n_head, hh = pr - sq * p.n_head;
            const float* qsrc = p.q + ((long)sq * S + qt * 32) * p.ld + hh * hs;
            const int qrows = S - qt * 32;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                dst[2 * kk] = load2(qsrc, p.ld, fi, kk * 8 + fh * 4, qrows, hd, vec2);
                dst[2 * kk + 1] = load2(qsrc, p.ld, fi, kk * 8 + fh * 4 + 2, qrows, hd, vec2);
            }
        }
    };
    auto load_q = [&](int grp, int qt) { q_rows(grp, qt, qraw); };
    auto prefetch_q = [&](int grp) { q_rows(grp, wave % WPP, qnext); };

#ifdef LIME_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    int group = blockIdx.x;
    if (PREFETCH) { fetch(group); prefetch_q(group); }
    for (; group < n_group; group += gridDim.x) {
        if (!PREFETCH) fetch(group);
        stash(group);
        ASTAMP(0)
        lds_barrier();
        ASTAMP(1)
        const int pair = group * G + g;
        const bool live = pair < n_pair;
        const int seq = live ? pair / p.n_head : 0, head = live ? pair - seq * p.n_head : 0;
        const int qt0 = wave % WPP;
        if (!PREFETCH) { if (live && qt0 * 32 < S) load_q(group, qt0); }
        else {
#pragma unroll
            for (int i = 0; i < NQ; ++i) qraw[i] = qnext[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PREFETCH && group + (int)gridDim.x < n_group) {          // next group's K / V / Q: in flight under the MFMAs
            fetch(group + gridDim.x);
            prefetch_q(group + gridDim.x);
        }
        __builtin_amdgcn_sched_barrier(0);
        ASTAMP(2)
        if (live) {
            for (int qt = qt0; qt < NT; qt += WPP) {
                if (qt * 32 >= S) break;
                if (qt != qt0) load_q(group, qt);
                f32x4 qf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if constexpr (FAST) {
                        qf[kk] = qraw[kk] * qscale;
                    } else {
                        qf[kk][0] = qraw[2 * kk][0] * qscale; qf[kk][1] = qraw[2 * kk][1] * qscale;
                        qf[kk][2] = qraw[2 * kk + 1][0] * qscale; qf[kk][3] = qraw[2 * kk + 1][1] * qscale;
                    }
                }
                f32x16 o;
                float inv;
                if constexpr (NT > 8) {
                    // ---- long sequences (S > 256: the 512-token bodies of BASELINE configs[3]): the score strip of a query would be
                    // NT x 16 = 256 registers (the whole-strip form below spilled 158-273 of them to scratch).  Keys go in blocks of
                    // four 32-key tiles with a running maximum: o and the probability sum are rescaled by exp2(m_old - m_new) when a
                    // block raises the maximum (both half-waves of a query see the same m: it is taken across the halves).
                    constexpr int KB = 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.f;
                    float m = -INFINITY, sum = 0.f;
                    for (int c0 = 0; c0 < NT; c0 += KB) {
                        f32x16 sc[KB];
#pragma unroll
                        for (int tt = 0; tt < KB; ++tt) {
                            const int t = c0 + tt;
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[tt][r] = 0.f;
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Kg[(t * 32 + fi) * LDH + kk * 8 + fh * 4]);
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    sc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u], qf[kk][u], sc[tt], 0, 0, 0);
                            }
                        }
                        if constexpr (!FAST) {
#pragma unroll
                            for (int tt = 0; tt < KB; ++tt) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float f = Fg[(c0 + tt) * 32 + (r & 3) + 8 * (r >> 2) + krow];
                                    const float v = sc[tt][r];
                                    sc[tt][r] = (f == 2.f) ? -INFINITY : ((f == 1.f) ? masked : v);   // masked_fill(mask == 0, -1e9), layers.py:233
                                }
                            }
                        }
                        float mc = sc[0][0];
#pragma unroll
                        for (int tt = 0; tt < KB; ++tt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) mc = fmaxf(mc, sc[tt][r]);
                        mc = fmaxf(mc, __shfl_xor(mc, 32));
                        const float m_new = fmaxf(m, mc);      // finite from the first block on (key 0 is never padding)
                        const float alpha = __builtin_amdgcn_exp2f(m - m_new);     // 0 on the first block (m = -inf), 1 when the maximum stands
                        sum *= alpha;
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
                        for (int tt = 0; tt < KB; ++tt) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float e = __builtin_amdgcn_exp2f(sc[tt][r] - m_new);
                                sc[tt][r] = e;
                                sum += e;
                            }
                        }
#pragma unroll
                        for (int tt = 0; tt < KB; ++tt) {
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vg[fi * LDVT + (c0 + tt) * 32 + gq * 8 + krow]);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[e], sc[tt][4 * gq + e], o, 0, 0, 0);
                            }
                        }
                        m = m_new;
                    }
                    sum += __shfl_xor(sum, 32);
                    inv = 1.0f / sum;
                    if (p.lse && fh == 0 && qt * 32 + fi < S) p.lse[((long)seq * S + qt * 32 + fi) * p.n_head + head] = m + log2f(sum);
                    ASTAMP(5)
                } else {
                // ---- S^T = K Q^T: keys on rows, this lane's query on the column (rows beyond S are zeros in LDS) ------
                f32x16 sc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const f32x4 kf = *reinterpret_cast<const f32x4*>(&Kg[(t * 32 + fi) * LDH + kk * 8 + fh * 4]);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u], qf[kk][u], sc[t], 0, 0, 0);
                    }
                }
                ASTAMP(3)
                // ---- key padding / key mask: a select per register (its key is the same for a whole half-wave) -----
                if constexpr (!FAST) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float f = Fg[t * 32 + (r & 3) + 8 * (r >> 2) + krow];
                            const float v = sc[t][r];
                            sc[t][r] = (f == 2.f) ? -INFINITY : ((f == 1.f) ? masked : v);   // masked_fill(mask == 0, -1e9), layers.py:233
                        }
                    }
                }
                // ---- softmax over the keys of this lane's query: own registers, then the other half-wave -----------
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
                        const float e = __builtin_amdgcn_exp2f(sc[t][r] - m);      // v_exp_f32: 1 ulp
                        sc[t][r] = e;
                        sum += e;
                    }
                }
                sum += __shfl_xor(sum, 32);
                inv = 1.0f / sum;
                if (p.lse && fh == 0 && qt * 32 + fi < S) p.lse[((long)seq * S + qt * 32 + fi) * p.n_head + head] = m + log2f(sum);
                ASTAMP(4)
                // ---- O^T = V^T P^T: probability registers are the B operand as they stand --------------------------
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        // keys 32t + 8gq + 4*half + 0..3 of head dim `fi` (rows beyond S are zeros)
                        const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vg[fi * LDVT + t * 32 + gq * 8 + krow]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[e], sc[t][4 * gq + e], o, 0, 0, 0);
                    }
                }
                ASTAMP(5)
                }
                // ---- transpose [head dim][query] -> [query][head dim] through the scratch, store whole head rows ----
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[fi * LDO + (r & 3) + 8 * (r >> 2) + krow] = o[r] * inv;
                lds_fence();
                if constexpr (BF) {
                    unsigned short* ob = reinterpret_cast<unsigned short*>(p.out);
                    if (fi < hd) {
#pragma unroll
                        for (int it = 0; it < 16; ++it) {
                            const int row = it * 2 + fh;
                            unsigned u = __builtin_bit_cast(unsigned, scr[row * LDO + fi]);
                            u += 0x7FFFu + ((u >> 16) & 1u);                                       // round to nearest even
                            ob[((long)seq * S + qt * 32 + row) * p.ldo + head * hd + fi] = (unsigned short)(u >> 16);
                        }
                    }
                    if (head == p.n_head - 1 && fi < p.out_pad) {                                  // zero columns behind the last head
#pragma unroll
                        for (int it = 0; it < 16; ++it)
                            ob[((long)seq * S + qt * 32 + it * 2 + fh) * p.ldo + p.n_head * hd + fi] = (unsigned short)0;
                    }
                } else if (fi < hd) {
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        const int row = it * 2 + fh;
                        const int qi = qt * 32 + row;
                        if (qi < S) p.out[((long)seq * S + qi) * p.ldo + head * hd + fi] = scr[row * LDO + fi];
                    }
                }
                lds_fence();
                ASTAMP(6)
            }
        }
        lds_barrier();                           // everyone is done with this group's LDS images
        ASTAMP(7)
    }
#ifdef LIME_STAMPS
    if (p.stamps && lane == 0) {
        for (int i = 0; i < 8; ++i) p.stamps[((long)blockIdx.x * 4 + wave) * 8 + i] = tsum[i];
    }
#endif
}

int attn_num_cus() {
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
int launch_bf16(AttnP p, hipStream_t s) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);
    p.n_group = (p.n_pair + G - 1) / G;
    const int per_cu = NT <= 2 ? 3 : (NT <= 4 ? 2 : 1);
    long blocks = (long)attn_num_cus() * per_cu;
    if (blocks > p.n_group) blocks = p.n_group;
    hipLaunchKernelGGL((token_attn_kernel<NT, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return lime_check_launch("lime_token_attention_bf16");
}

template <int NT>
int launch(AttnP p, hipStream_t s) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);
    p.n_group = (p.n_pair + G - 1) / G;
    // LDS per workgroup decides how many are resident per CU; a few persistent workgroups per CU
    const int per_cu = NT <= 2 ? 3 : (NT <= 4 ? 2 : 1);
    long blocks = (long)attn_num_cus() * per_cu;
    if (blocks > p.n_group) blocks = p.n_group;
#ifdef LIME_STAMPS
    p.stamps = g_attn_stamp_buf;
#endif
    // FAST: no mask, S a multiple of 32, heads padded to 32 columns, 16-byte aligned operands
    const bool fast = p.mask == nullptr && p.S == NT * 32 && p.hs == 32 && p.ld % 4 == 0 &&
                      (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) % 16 == 0);
    if (fast) hipLaunchKernelGGL((token_attn_kernel<NT, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((token_attn_kernel<NT, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return lime_check_launch("lime_token_attention_f32");
}

template <int NT>
int launch_map(AttnP p, hipStream_t s) {
    constexpr int G = (NT >= 3) ? 1 : (4 / NT);
    p.n_group = (p.n_pair + G - 1) / G;
    const int per_cu = NT <= 2 ? 3 : (NT <= 4 ? 2 : 1);
    long blocks = (long)attn_num_cus() * per_cu;
    if (blocks > p.n_group) blocks = p.n_group;
    hipLaunchKernelGGL((token_attn_kernel<NT, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return lime_check_launch("lime_token_attention_rows_f32");
}

}  // namespace

extern "C" int lime_token_attention_rows_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const int32_t* row_map,
                                             const int32_t* n_seq_dev, float* out, int64_t ldo, int32_t n_seq, int32_t S,
                                             int32_t n_head, int32_t head_dim, float scale, void* stream) {
    LIME_REQUIRE(q && k && v && out && row_map, LIME_ERR_BAD_ARG, "lime_token_attention_rows_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && n_head > 0 && head_dim > 0 && head_dim <= 32, LIME_ERR_BAD_ARG,
                 "lime_token_attention_rows_f32: bad dims n_seq=%d n_head=%d head_dim=%d", n_seq, n_head, head_dim);
    LIME_REQUIRE(S > 0 && S <= 512 && S % 32 == 0 && (S / 32 <= 4 || S / 32 == 8 || S / 32 == 16), LIME_ERR_UNSUPPORTED,
                 "lime_token_attention_rows_f32: S must be 32, 64, 96, 128, 256 or 512 (got %d)", S);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * 32 && ld_qkv % 4 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0),
                 LIME_ERR_BAD_ARG, "lime_token_attention_rows_f32: heads are 32 columns apart (zero padded), rows 16-byte aligned");
    LIME_REQUIRE(ldo >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG, "lime_token_attention_rows_f32: ldo smaller than n_head * head_dim");
    if (n_seq == 0) return LIME_OK;
    hipStream_t s = (hipStream_t)stream;
    {   // S <= 128: the split-product kernel (token_attn_sp_f32.hip) unless lime_set_split_gemm(0)
        const int st = lime_token_attention_sp(q, k, v, (long)ld_qkv, row_map, n_seq_dev, out, (long)ldo, n_seq, S, n_head, head_dim, scale, nullptr, s);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    AttnP p{q, k, v, (long)ld_qkv, nullptr, out, (long)ldo, n_seq, S, n_head, head_dim, 32, scale, n_seq * n_head, 1, 0, 0, row_map, n_seq_dev};
    switch (S / 32) {
        case 1: return launch_map<1>(p, s);
        case 2: return launch_map<2>(p, s);
        case 3: return launch_map<3>(p, s);
        case 4: return launch_map<4>(p, s);
        case 8: return launch_map<8>(p, s);
        default: return launch_map<16>(p, s);
    }
}

extern "C" int lime_token_attention_count_f32(const float* q, const float* k, const float* v, int64_t ld_qkv,
                                              const uint8_t* key_mask, const int32_t* n_seq_dev, float* out, int64_t ldo, int32_t n_seq,
                                              int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride, float scale, void* stream) {
    LIME_REQUIRE(q && k && v && out, LIME_ERR_BAD_ARG, "lime_token_attention_count_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0, LIME_ERR_BAD_ARG,
                 "lime_token_attention_count_f32: bad dims n_seq=%d S=%d n_head=%d head_dim=%d", n_seq, S, n_head, head_dim);
    LIME_REQUIRE(head_dim <= 32, LIME_ERR_UNSUPPORTED, "lime_token_attention_count_f32: head_dim %d > 32", head_dim);
    LIME_REQUIRE(head_stride >= head_dim, LIME_ERR_BAD_ARG, "lime_token_attention_count_f32: head_stride %d < head_dim %d", head_stride, head_dim);
    LIME_REQUIRE(S <= 512, LIME_ERR_UNSUPPORTED, "lime_token_attention_count_f32: S %d > 512", S);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * head_stride && ldo >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG,
                 "lime_token_attention_count_f32: leading dimension smaller than n_head * head_dim");
    if (n_seq == 0) return LIME_OK;
    // 8-byte loads need an even head_dim and leading dimension and 8-byte aligned bases
    const int vec2 = (head_dim % 2 == 0) && (head_stride % 2 == 0) && (ld_qkv % 2 == 0) &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (key_mask == nullptr && head_stride == 32) {   // the encoder layers' shapes: the split-product kernel (token_attn_sp_f32.hip)
        const int st = lime_token_attention_sp(q, k, v, (long)ld_qkv, nullptr, n_seq_dev, out, (long)ldo, n_seq, S, n_head, head_dim, scale, nullptr, s);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    AttnP p{q, k, v, (long)ld_qkv, key_mask, out, (long)ldo, n_seq, S, n_head, head_dim, head_stride, scale, n_seq * n_head, vec2, 0, 0, nullptr, n_seq_dev};
    const int nt = (S + 31) / 32;
    if (nt <= 1) return launch<1>(p, s);
    if (nt <= 2) return launch<2>(p, s);
    if (nt <= 3) return launch<3>(p, s);
    if (nt <= 4) return launch<4>(p, s);
    if (nt <= 8) return launch<8>(p, s);
    return launch<16>(p, s);
}


extern "C" int lime_token_attention_lse_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, int64_t ldo,
                                            float* lse, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride,
                                            float scale, void* stream) {
    LIME_REQUIRE(q && k && v && out && lse, LIME_ERR_BAD_ARG, "lime_token_attention_lse_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0, LIME_ERR_BAD_ARG,
                 "lime_token_attention_lse_f32: bad dims n_seq=%d S=%d n_head=%d head_dim=%d", n_seq, S, n_head, head_dim);
    LIME_REQUIRE(head_dim <= 32 && head_stride >= head_dim && S <= 512, LIME_ERR_UNSUPPORTED,
                 "lime_token_attention_lse_f32: needs head_dim <= 32, head_stride >= head_dim, S <= 512");
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * head_stride && ldo >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG,
                 "lime_token_attention_lse_f32: leading dimension smaller than n_head * head_dim");
    if (n_seq == 0) return LIME_OK;
    const int vec2 = (head_dim % 2 == 0) && (head_stride % 2 == 0) && (ld_qkv % 2 == 0) &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (head_stride == 32) {                          // S = 256 / 512, padded heads: the split-product kernel with key blocks
        const int st = lime_token_attention_sp(q, k, v, (long)ld_qkv, nullptr, nullptr, out, (long)ldo, n_seq, S, n_head, head_dim, scale, lse, s);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    AttnP p{q, k, v, (long)ld_qkv, nullptr, out, (long)ldo, n_seq, S, n_head, head_dim, head_stride, scale, n_seq * n_head, vec2, 0, 0, nullptr, nullptr, lse};
    const int nt = (S + 31) / 32;
    if (nt <= 1) return launch<1>(p, s);
    if (nt <= 2) return launch<2>(p, s);
    if (nt <= 3) return launch<3>(p, s);
    if (nt <= 4) return launch<4>(p, s);
    if (nt <= 8) return launch<8>(p, s);
    return launch<16>(p, s);
}

extern "C" int lime_token_attention_f32(const float* q, const float* k, const float* v, int64_t ld_qkv,
                                        const uint8_t* key_mask, float* out, int64_t ldo, int32_t n_seq, int32_t S,
                                        int32_t n_head, int32_t head_dim, int32_t head_stride, float scale, void* stream) {
    return lime_token_attention_count_f32(q, k, v, ld_qkv, key_mask, nullptr, out, ldo, n_seq, S, n_head, head_dim, head_stride, scale, stream);
}

int lime_token_attention_bf16_mfma(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, uint16_t* out, int64_t ldo,
                                   int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, float scale, int32_t out_pad,
                                   hipStream_t s);          // token_attn_bf16.hip

extern "C" int lime_token_attention_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld_qkv, uint16_t* out,
                                         int64_t ldo, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, float scale,
                                         int32_t out_cols, void* stream) {
    LIME_REQUIRE(q && k && v && out, LIME_ERR_BAD_ARG, "lime_token_attention_bf16: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0 && head_dim <= 32, LIME_ERR_BAD_ARG,
                 "lime_token_attention_bf16: bad dims n_seq=%d S=%d n_head=%d head_dim=%d", n_seq, S, n_head, head_dim);
    LIME_REQUIRE(S == 32 || S == 64 || S == 128 || S == 256 || S == 512, LIME_ERR_UNSUPPORTED,
                 "lime_token_attention_bf16: S must be 32, 64, 128, 256 or 512 (got %d)", S);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * 32 && ld_qkv % 8 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0),
                 LIME_ERR_BAD_ARG, "lime_token_attention_bf16: heads are 32 bf16 columns apart, rows 16-byte aligned");
    const int pad = out_cols - n_head * head_dim;
    LIME_REQUIRE(pad >= 0 && pad <= 32 && ldo >= out_cols, LIME_ERR_BAD_ARG,
                 "lime_token_attention_bf16: out_cols must be in [n_head * head_dim, n_head * head_dim + 32] and <= ldo");
    if (n_seq == 0) return LIME_OK;
    hipStream_t s = (hipStream_t)stream;
    {   // S <= 128: scores and P.V on the bf16 matrix cores; longer sequences: the fp32-core variant below
        const int st = lime_token_attention_bf16_mfma(q, k, v, ld_qkv, out, ldo, n_seq, S, n_head, head_dim, scale, pad, s);
        if (st != 1) return st;
    }
    AttnP p{(const float*)q, (const float*)k, (const float*)v, (long)ld_qkv, nullptr, (float*)out, (long)ldo, n_seq, S, n_head,
            head_dim, 32, scale, n_seq * n_head, 1, 0, pad, nullptr, nullptr};
    switch (S / 32) {
        case 1: return launch_bf16<1>(p, s);
        case 2: return launch_bf16<2>(p, s);
        case 4: return launch_bf16<4>(p, s);
        case 8: return launch_bf16<8>(p, s);
        default: return launch_bf16<16>(p, s);
    }
}
