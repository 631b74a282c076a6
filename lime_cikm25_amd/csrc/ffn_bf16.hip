// lime_encoder_ffn_bf16 / lime_encoder_block_bf16: the feed-forward half of a TransformerEncoderLayer (newsEncoders.py:244-247, 316-321 --
// linear1, ReLU, linear2, residual, norm2, and the token mean pooling behind the last layer) in ONE launch on the bf16
// matrix cores.  The 512-wide hidden state never leaves the registers, the layer input is read from HBM once.
//
// Why (profiles/r02_notes.md): at bf16 rates the K = 300 encoder GEMMs of gemm_pp_f32.hip are bound by what a CU can pull
// through its L2 -> LDS path (~21 B/clk/CU on L2 hits, less from HBM): a 128 x 320 x 304 tile stages 287 KB for 6.4k
// cycles of MFMA work, and linear1 -> linear2 additionally writes and re-reads the hidden state (2 KB per token).  Here
//   * a workgroup (4 waves, ONE per CU, 160 KB of LDS) keeps a 128-token tile of the layer input STATIONARY in LDS
//     ([chunk][row][64 bytes], the swizzled image of gemm_pp_f32.hip) and streams only the weights: a ring of four 19 KB
//     slots filled by LDS-DMA three steps ahead of their use (counted s_waitcnt vmcnt, raw s_barrier: the DMAs stay in
//     flight across the barriers).  All weight bytes are L2 hits; per tile 622 KB for 20.5k cycles of MFMA work.
//   * a wave owns 32 tokens x all columns.  A pass computes 128 hidden columns for its tokens (five steps, two 32-deep
//     k chunks each), applies ReLU, rounds to bf16 IN REGISTER ORDER and feeds them straight back as the B operand of
//     linear2 (four steps, 304 output columns): the k index of an MFMA is only a summation label, so linear2's weight
//     columns are stored in the order the hidden registers come out (lime_ffn_pack_bf16) -- the trick of the bf16
//     attention kernel's P V product.  No LDS round trip, no cross-lane movement.
//   * linear1's bias rides in the GEMM: the input has zero pad columns (E = 300 carried as 304); the LDS image gets 1.0
//     in column E and the packed weight holds the bias there.
//   * residual (the stationary tile), LayerNorm and the 32-token block means are the fp32 epilogue; with `pool32` only
//     [M / 32, 304] floats are written.
// Tokens of a wave never meet another wave's: the stationary image needs no barrier, only the weight ring does (one per step).
//
// lime_encoder_block_bf16 (the OPROJ instantiations) puts out_proj + residual + norm1 in front: the image first holds the
// attention output of the tile; ten more steps stream out_proj's weight; behind each of them the residual rows (word rows by
// id, or the layer input) follow the attention output into the SAME image chunk by chunk; the epilogue adds them and the fp32
// add_rows (bias + positional rows, fetched two steps ahead), applies LayerNorm and writes the bf16 result back over the image:
// the feed-forward half's input and residual, which therefore never exists in HBM.
#include <type_traits>

#include "lds_dma.h"

using namespace lime_dev;

#ifdef LIME_STAMPS
// Diagnostic build only (tools/ffn_stamps.py): per-wave s_memtime sums of the step segments; never in liblime_hip.so.
static unsigned long long* g_ffn_stamp_buf = nullptr;
extern "C" void lime_debug_set_ffn_stamp_buffer(unsigned long long* p) { g_ffn_stamp_buf = p; }
#define FSTAMP(i)                                                           \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 \
        tsum[i] += t_ - tlast;                                              \
        tlast = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
    }
#else
#define FSTAMP(i)
#endif

namespace {

constexpr int BM = 128;                    // tokens per tile (4 waves x 32)
constexpr int ND = 19, DP = 16 * ND;       // model columns carried: 304
constexpr int NCH = 10;                    // 32-deep k chunks of the layer input (320 >= 304; the tail is zero-filled)
constexpr int PW = 128;                    // hidden columns per pass
constexpr int NT1 = PW / 16;               // linear1 output tiles per pass
constexpr int SLAB = BM * 64;              // bytes of one chunk of the stationary image
constexpr int XS_BYTES = NCH * SLAB;       // 81,920
constexpr int SLOT = DP * 64;              // 19,456: one ring slot (two linear1 weight chunks, or one linear2 chunk)
constexpr int NSLOT = 4;
constexpr int CONST_OFF = XS_BYTES + NSLOT * SLOT;
constexpr int LDS_BYTES = CONST_OFF + 3 * DP * 4;      // + b2, gamma, beta = 163,392 of the CU's 163,840
constexpr int STEPS = 9;                   // per pass: five linear1 steps, four linear2 steps

struct FfnP {
    const uint16_t* x; long ldx;
    const uint16_t* w1p;
    const uint16_t* w2p;
    const float* b2; const float* g; const float* beta; float eps;
    void* out; long ldo;
    int M, E, F;
    const int* m_dev;
    // OPROJ instantiations: x is the attention output; the block starts with out_proj + residual + LayerNorm (norm1)
    const uint16_t* w0p; const float* g1; const float* beta1; float eps1;
    int res_kind; const uint16_t* res; long ldr; const int* res_ids; const float* res_pe; long ldpe; int res_period;
#ifdef LIME_STAMPS
    unsigned long long* stamps;
#endif
};

// The lane id, recomputed where it is called (the opaque zero keeps hipcc from hoisting it -- and everything derived from it -- out
// of the tile loop, where the values would sit in registers through all 36 steps or be spilled: scratch reloads wait vmcnt(0)).
__device__ __forceinline__ int lane_here() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

constexpr int step_dmas(int pos) { return pos < 5 ? 4 : 5; }                       // DMA instructions per wave that fill the slot of step `pos`
// A tile's steps by "tile position": 0..9 the out_proj chunks (OPROJ instantiations only), 10 + pos the feed-forward steps of a pass
constexpr int TP_FFN = NCH;
constexpr int tp_dmas(int tp) { return tp < TP_FFN ? 5 : step_dmas(tp - TP_FFN); }

template <bool POOL, bool OPROJ>
__global__ __launch_bounds__(256, 1) void ffn_bf16_kernel(const FfnP p) {
    // ONE __shared__ object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kg = lane >> 4;
    int M = p.M;
    if (p.m_dev) {
        const int m = __builtin_amdgcn_readfirstlane(*p.m_dev);
        M = m < M ? (m > 0 ? m : 0) : M;
    }
    const int ntiles = (M + BM - 1) / BM;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int NP = p.F / PW;
    const int E = p.E;

    float* const cs = reinterpret_cast<float*>(lds + CONST_OFF);
    for (int c = tid; c < DP; c += 256) {
        cs[c] = c < E ? p.b2[c] : 0.f;
        cs[DP + c] = c < E ? p.g[c] : 0.f;
        cs[2 * DP + c] = c < E ? p.beta[c] : 0.f;
    }
    __syncthreads();                                   // nothing in flight yet

    // ---- loader: DMA instruction `idx` of an image covers rows 16 idx .. 16 idx + 15; lane l fills row 16 idx + (l >> 2),
    // physical segment l & 3 = logical segment (l & 3) ^ swz4((l >> 4) & 3)
    const int srow = lane >> 2;
    const int lseg = (lane & 3) ^ swz4((lane >> 4) & 3);
    const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(p.w1p), rs_w2 = make_rsrc(p.w2p);
    const unsigned ldxb = (unsigned)p.ldx * 2u;
    unsigned char* const ring = lds + XS_BYTES;

    // The packed weights (lime_ffn_pack_bf16) hold every ring slot as ONE contiguous block in the order of its LDS image, so a DMA
    // instruction reads 1 KB of whole 128-byte lines (rows 608 / 1024 bytes apart would be 64-byte pieces that land on a quarter of
    // the L2 channels): block + 1024 idx + 64 (row in the instruction) + 16 (logical segment).
    const unsigned w_lane = (unsigned)srow * 64u + (unsigned)lseg * 16u;
    // the slot of a linear1 step: weight rows PW pass .. + 127, chunks 2 j and 2 j + 1 -> [2][128 rows][64 B]; 16 instructions, 4 per wave
    auto issue_w1 = [&](int slot, int pass, int j, int i) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 2       // tools/ffn_stamps.py: no weight traffic (results are garbage)
        return;
#endif
        const int idx = 4 * wave + i;
        dma16(rs_w1, ring + slot * SLOT + idx * 1024, w_lane, (pass * NCH + 2 * j) * SLAB + idx * 1024);      // the wave-uniform part rides in the scalar offset: one offset register for every weight DMA
    };
    // the slot of a linear2 step: all DP weight rows, hidden columns 32 (4 pass + kc) .. + 31 -> [304 rows][64 B]; 19 instructions,
    // 5 per wave (wave 3 repeats the last one: the same bytes to the same place, so that every wave counts alike)
    auto issue_w2 = [&](int slot, int pass, int kc, int i) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 2
        return;
#endif
        int idx = 5 * wave + i;
        idx = idx < ND ? idx : ND - 1;
        dma16(rs_w2, ring + slot * SLOT + idx * 1024, w_lane, (NT1 / 2 * pass + kc) * SLOT + idx * 1024);
    };
    // the slot of an out_proj step: all DP weight rows, k chunk c -> [304 rows][64 B], 19 instructions as above
    const __amdgpu_buffer_rsrc_t rs_w0 = make_rsrc(OPROJ ? p.w0p : p.w2p);
    auto issue_w0 = [&](int slot, int c, int i) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 2
        return;
#endif
        int idx = 5 * wave + i;
        idx = idx < ND ? idx : ND - 1;
        dma16(rs_w0, ring + slot * SLOT + idx * 1024, w_lane, c * SLOT + idx * 1024);
    };
    // instruction i of this wave's share of the slot of the step at tile position TP (of pass `pass`)
    auto issue_one = [&](auto tp_c, int slot, int pass, int i) {
        constexpr int TP = decltype(tp_c)::value;
        if constexpr (TP < TP_FFN) issue_w0(slot, TP, i);
        else if constexpr (TP - TP_FFN < 5) issue_w1(slot, pass, TP - TP_FFN, i);
        else issue_w2(slot, pass, TP - TP_FFN - 5, i);
    };
    auto issue_step = [&](auto tp_c, int slot, int pass) {
        constexpr int TP = decltype(tp_c)::value;
#pragma unroll
        for (int i = 0; i < tp_dmas(TP); ++i) issue_one(tp_c, slot, pass, i);
    };
    // this wave's 32 rows of the stationary tile: 2 x NCH instructions
    int rid_next[2] = {0, 0};                           // OPROJ, gathered residual: the ids of this lane's two residual rows of the tile load_x issues
    auto load_x = [&](int t) {
        const long row0 = (long)t * BM;
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x + row0 * p.ldx);
        const int ln = lane_here();                    // recompute the offsets here: hoisted out of the tile loop they end up spilled
        const int srow_ = ln >> 2, lseg_ = (ln & 3) ^ swz4((ln >> 4) & 3);
        if constexpr (OPROJ) {
            if (p.res_kind == 2) {                     // used behind that tile's first wait (everything waited for), so this load costs no drain
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long row = row0 + 16 * (2 * wave + j) + srow_;
                    rid_next[j] = p.res_ids[row < M ? row : 0];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = 16 * (2 * wave + j) + srow_;
            const unsigned vo = (row0 + rl < M) ? (unsigned)rl * ldxb + (unsigned)lseg_ * 16u : OOB;
#pragma unroll
            for (int c = 0; c < NCH; ++c) dma16(rs_x, lds + c * SLAB + (2 * wave + j) * 1024, (c * 32 + lseg_ * 8 < DP) ? vo : OOB, c * 64);
        }
    };

    // ---- compute state: v_mfma_f32_16x16x32_bf16, a = weight rows (output columns), b = tokens: lane (i, kg) reads ONE
    // b128 = k 8 kg .. 8 kg + 7 of its row per operand and chunk; the result tile is D^T: lane (token i, kg) holds columns
    // 16 t + 4 kg + r, r = 0..3
    const int pseg = (kg ^ swz4((fi >> 2) & 3)) * 16;
    const int x_off = (32 * wave + fi) * 64 + pseg;                 // + tt * 1024 + c * SLAB
    const int w_off = fi * 64 + pseg;                               // + t * 1024
    f32x4 acc1[2][NT1], acc2[2][ND];
    bf16x8 hb[2][NT1 / 2];

    // A step's MFMAs run in four groups; `part(g)` behind group g issues this wave's DMA instruction g of the slot that has just
    // come free (issued all at once behind the barrier, the workgroup's 16-19 instructions queue up in the CU's one address path
    // and every wave sits in "DMA issue" for a quarter of its time -- s_memtime stamps; spread out, a wave meets an idle path).
    // The fragments of group g + 1 are read before the MFMAs of group g (the sched_barriers that pin the DMA issue also keep
    // hipcc from hoisting those reads) -- and those of the NEXT step's first group before this step's last MFMAs: a slot is
    // published one barrier before the step that consumes it, so the reads cross the barrier and no step starts with an
    // exposed LDS round trip.  nw / nx carry them over.
    bf16x8 nw[4], nx[2];
    // the first four weight fragments of the step that reads `slot` (the same addresses for every kind of slot); xc >= 0: also the
    // stationary fragments of its first chunk
    auto prefetch = [&](int slot, int xc) {
        const unsigned char* const sb = ring + slot * SLOT + w_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(sb + t * 1024);
        if (xc >= 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) nx[tt] = *reinterpret_cast<const bf16x8*>(lds + xc * SLAB + x_off + tt * 1024);
        }
    };
    auto compute1 = [&](int slot, int j, bool have_x, auto&& part, auto&& tail) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 1       // tools/ffn_stamps.py: no fragment reads, no MFMAs
        for (int g = 0; g < 4; ++g) part(g);
        return;
#endif
        const unsigned char* const sb = ring + slot * SLOT + w_off;
        constexpr int GT = NT1 / 2;                                // 4 tiles per group: (chunk cc, tile half)
        bf16x8 af[2][2], wf[2][GT];
        auto read_group = [&](int g, int buf) {
            const int cc = g >> 1, th = g & 1;
            if (th == 0) {
                const unsigned char* const xb = lds + (2 * j + cc) * SLAB + x_off;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) af[cc][tt] = *reinterpret_cast<const bf16x8*>(xb + tt * 1024);
            }
#pragma unroll
            for (int t = 0; t < GT; ++t) wf[buf][t] = *reinterpret_cast<const bf16x8*>(sb + cc * 8192 + (th * GT + t) * 1024);
        };
#pragma unroll
        for (int t = 0; t < GT; ++t) wf[0][t] = nw[t];
        if (have_x) {
            af[0][0] = nx[0];
            af[0][1] = nx[1];
        } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) af[0][tt] = *reinterpret_cast<const bf16x8*>(lds + (2 * j) * SLAB + x_off + tt * 1024);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) read_group(g + 1, (g + 1) & 1);
            else tail();
            const int cc = g >> 1, th = g & 1;
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 3       // fragment reads, no MFMAs
                    acc1[tt][th * GT + t][0] += __builtin_bit_cast(f32x4, wf[g & 1][t])[0] + __builtin_bit_cast(f32x4, af[cc][tt])[0];
#else
                    acc1[tt][th * GT + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g & 1][t], af[cc][tt], acc1[tt][th * GT + t], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
            part(g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // linear2 (b = the packed hidden chunk) and out_proj (b = the stationary fragments of a chunk): 19 tiles x 2 token halves
    auto compute2 = [&](int slot, const bf16x8& b0, const bf16x8& b1, auto&& part, auto&& tail) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 1
        for (int g = 0; g < 5; ++g) part(g);
        return;
#endif
        const unsigned char* const sb = ring + slot * SLOT + w_off;
        constexpr int GT = 4, NG = (ND + GT - 1) / GT;             // five groups of 4, 4, 4, 4, 3 tiles
        bf16x8 wf[2][GT];
#pragma unroll
        for (int t = 0; t < GT; ++t) wf[0][t] = nw[t];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int t = 0; t < GT; ++t)
                    if ((g + 1) * GT + t < ND) wf[(g + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(sb + ((g + 1) * GT + t) * 1024);
            } else {
                tail();
            }
#pragma unroll
            for (int t = 0; t < GT; ++t)
                if (g * GT + t < ND) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 3
                        acc2[tt][g * GT + t][0] += __builtin_bit_cast(f32x4, wf[g & 1][t])[0];
#else
                        acc2[tt][g * GT + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g & 1][t], tt ? b1 : b0, acc2[tt][g * GT + t], 0, 0, 0);
#endif
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            part(g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Step s reads slot s & 3.  In front of it: this wave's part of slot s + 1 has landed (counted wait: the DMAs of step s + 2
    // stay in flight), then the barrier -- behind it slot s + 1 is complete for everyone (it is read from the end of this step
    // on) and slot s - 1 is free: the DMAs of step s + 3 go there, between this step's MFMA groups.
    int gs = 0;
    bool last = false;
#ifdef LIME_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    // OPROJ: the residual rows (word-table rows by id, or the layer input's rows) follow the attention output into the SAME image,
    // chunk by chunk behind the out_proj steps that have consumed it.  Per-lane source offsets of this lane's two rows (set at a
    // tile's first step), and chunk c of them -> the image
    unsigned res_voff[2] = {OOB, OOB};
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(OPROJ ? (const void*)p.res : (const void*)p.x);
    auto set_res_rows = [&](long row0) {
        if constexpr (OPROJ) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long row = row0 + 16 * (2 * wave + j) + srow;
                const unsigned rsel = p.res_kind == 3 ? (unsigned)row : (unsigned)rid_next[j];
                res_voff[j] = row < M ? rsel * (unsigned)(p.ldr * 2) + (unsigned)lseg * 16u : OOB;
            }
        }
    };
    auto issue_res = [&](int c) {
        if constexpr (OPROJ) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dma16(rs_res, lds + c * SLAB + (2 * wave + j) * 1024, (c * 32 + lseg * 8 < DP) ? res_voff[j] : OOB, c * 64);
        }
    };
    auto step = [&](auto tp_c, int pass) {
        constexpr int TP = decltype(tp_c)::value;
        constexpr bool K0 = TP < TP_FFN;                // an out_proj step
        constexpr int POS = K0 ? 0 : TP - TP_FFN;       // feed-forward steps: position in the pass
        const bool tile_start = OPROJ ? TP == 0 : (TP == TP_FFN && pass == 0);
        if (tile_start) {
            wait_vm<0>();                              // the stationary tile (issued in front of the previous tile's epilogue)
            if constexpr (!OPROJ) {
                const int lp = lane_here();
                if (lp < 32) {                         // 1.0 in column E of this wave's rows: linear1's bias column
                    const int row = 32 * wave + lp;
                    const int koff = E & 31;
                    *reinterpret_cast<unsigned short*>(lds + (E >> 5) * SLAB + row * 64 + (((koff >> 3) ^ swz4((row >> 2) & 3)) * 16) +
                                                       (koff & 7) * 2) = 0x3F80;
                }
            }
        } else if (!K0 && last && pass == NP - 1 && POS >= STEPS - 2) {
            wait_vm<0>();                              // the ring runs dry behind the last tile
        } else {
            // the slot after next stays in flight: 5 instructions per wave for an out_proj / linear2 slot, 4 for a linear1 slot (where the
            // next tile's first steps may be either, the smaller count: waiting for one instruction more is always safe)
            wait_vm<K0 ? (TP + 2 < TP_FFN ? 5 : 4) + (TP >= 2 ? 2 : 0) : step_dmas((POS + 2) % STEPS)>();   // K0: + the previous step's residual chunk
        }
        FSTAMP(0)                                      // 0: this wave's DMAs of the next step have landed
        ring_barrier();
        FSTAMP(1)                                      // 1: barrier
        // the step three ahead, whose slot has just come free: inside this tile, or the next tile's first steps
        constexpr int FTP = K0 ? (TP + 3 < TP_FFN ? TP + 3 : TP_FFN + (TP + 3 - TP_FFN)) : TP_FFN + (POS + 3) % STEPS;
        constexpr int WTP = OPROJ ? (POS + 3) % STEPS : FTP;         // ... when it wraps into the next tile
        int fp = K0 ? 0 : pass + (POS + 3 >= STEPS ? 1 : 0);
        bool go = true, wrap = false;
        if (!K0 && fp == NP) { fp = 0; go = !last; wrap = true; }
        const int fslot = (gs + 3) & 3;
        constexpr int NGRP = (!K0 && POS < 5) ? 4 : 5;  // MFMA groups of this step; the refilled slot takes 4 or 5 instructions
        auto part = [&](int g) {
            if (!go) return;
            if (OPROJ && wrap) {
                if (g < tp_dmas(WTP)) issue_one(std::integral_constant<int, WTP>{}, fslot, 0, g);
                if (g == NGRP - 1 && NGRP < tp_dmas(WTP)) issue_one(std::integral_constant<int, WTP>{}, fslot, 0, NGRP);
            } else {
                if (g < tp_dmas(FTP)) issue_one(std::integral_constant<int, FTP>{}, fslot, fp, g);
                if (g == NGRP - 1 && NGRP < tp_dmas(FTP)) issue_one(std::integral_constant<int, FTP>{}, fslot, fp, NGRP);
            }
        };
        // the next step's first fragments; its stationary fragments too while the image it reads is in place
        int nxc = -1;
        if constexpr (K0) {
            if (TP + 1 < TP_FFN) nxc = TP + 1;         // (the step behind the last out_proj chunk reads the LayerNorm output: not there yet)
        } else if constexpr (POS < 4) {
            nxc = 2 * (POS + 1);
        } else if constexpr (POS == STEPS - 1) {
            if (pass + 1 < NP) nxc = 0;
        }
        auto tail = [&]() { prefetch((gs + 1) & 3, nxc); };
        if constexpr (K0) {
            if constexpr (TP == 0) set_res_rows((long)tile * BM);
            else issue_res(TP - 1);                     // chunk TP - 1 of the attention output is done with (this wave's rows)
            bf16x8 a0, a1;
            if (TP > 0) { a0 = nx[0]; a1 = nx[1]; }
            else {
                a0 = *reinterpret_cast<const bf16x8*>(lds + TP * SLAB + x_off);
                a1 = *reinterpret_cast<const bf16x8*>(lds + TP * SLAB + x_off + 1024);
            }
            compute2(gs & 3, a0, a1, part, tail);
        } else if constexpr (POS < 5) {
            compute1(gs & 3, POS, !(POS == 0 && pass == 0), part, tail);
        } else {
            compute2(gs & 3, hb[0][POS - 5], hb[1][POS - 5], part, tail);
        }
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(K0 ? 2 : (POS < 5 ? 3 : 4))             // 2 / 3 / 4: fragment reads + MFMAs + DMA issue: out_proj / linear1 / linear2 step
        ++gs;
    };
    auto zero_acc2 = [&]() {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int t = 0; t < ND; ++t) acc2[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    constexpr int TP0 = OPROJ ? 0 : TP_FFN;             // a tile's first step
    issue_step(std::integral_constant<int, TP0>{}, 0, 0);
    issue_step(std::integral_constant<int, TP0 + 1>{}, 1, 0);
    issue_step(std::integral_constant<int, TP0 + 2>{}, 2, 0);
    load_x(tile);
    wait_vm<tp_dmas(TP0 + 1) + tp_dmas(TP0 + 2) + 2 * NCH>();     // slot 0
    ring_barrier();
    prefetch(0, -1);
    for (; tile < ntiles; tile += gridDim.x) {
        last = tile + (int)gridDim.x >= ntiles;
        zero_acc2();
        if constexpr (OPROJ) {
            step(std::integral_constant<int, 0>{}, 0);
            step(std::integral_constant<int, 1>{}, 0);
            step(std::integral_constant<int, 2>{}, 0);
            step(std::integral_constant<int, 3>{}, 0);
            step(std::integral_constant<int, 4>{}, 0);
            step(std::integral_constant<int, 5>{}, 0);
            step(std::integral_constant<int, 6>{}, 0);
            // ---- out_proj epilogue: + residual (its last chunk goes out behind step 9; the rest followed the attention output into
            // the image) + add_rows (fp32, by r % add_period: out_proj's bias, with the positional rows where the residual is the bare
            // word rows), LayerNorm (norm1) -> bf16, written over this wave's rows of the image with 1.0 in column E: the
            // feed-forward half's input, residual and bias column.  The fp32 rows are plain loads (cache resident) issued two steps
            // ahead; gamma / beta are loaded in the epilogue: hipcc drains vmcnt for them, once per tile.
            const long row0 = (long)tile * BM;
            const int le = lane_here();
            const int fi_ = le & 15, kg_ = le >> 4;
            const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(p.g1), rs_e = make_rsrc(p.beta1), rs_pe = make_rsrc(p.res_pe);
            unsigned pof[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const long row = row0 + 32 * wave + 16 * tt + fi_;
                pof[tt] = ((unsigned)row % (unsigned)p.res_period) * (unsigned)(p.ldpe * 4) + (unsigned)kg_ * 16u;      // 32-bit: rows < 2^31
            }
            f32x4 pe[2][ND];
            auto load_add_rows = [&](int t0, int t1) {
#pragma unroll
                for (int t = t0; t < t1; ++t) {
                    const bool colok = 16 * t + 4 * kg_ < E;       // E % 4 == 0: a lane's four columns are real or pad together
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        pe[tt][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_pe, colok ? pof[tt] + (unsigned)t * 64u : OOB, 0, 0));
                }
            };
            load_add_rows(0, 10);
            step(std::integral_constant<int, 7>{}, 0);
            load_add_rows(10, ND);
            step(std::integral_constant<int, 8>{}, 0);
            step(std::integral_constant<int, 9>{}, 0);
            issue_res(NCH - 1);
            unsigned char* const wbase = lds + (32 * wave + fi_) * 64 + 8 * (kg_ & 1);
            const int rswz = swz4((fi_ >> 2) & 3), rsg = kg_ >> 1;
            float sum[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
            wait_vm<0>();                                  // the residual chunks
#pragma unroll
            for (int t = 0; t < ND; ++t) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int seg = 2 * (t & 1) + rsg;
                    const f32x4 r = unpack_bf16x4(*reinterpret_cast<const u32x2*>(wbase + (t >> 1) * SLAB + tt * 1024 + ((seg ^ rswz) * 16)));
                    const f32x4 v = acc2[tt][t] + pe[tt][t] + r;
                    acc2[tt][t] = v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sum[tt] += v[j]; sq[tt] += v[j] * v[j]; }
                }
            }
            float mean[2], rstd[2];
            const float inv_n = 1.0f / (float)E;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float s1 = sum[tt], s2 = sq[tt];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                mean[tt] = s1 * inv_n;
                rstd[tt] = rsqrtf(fmaxf(s2 * inv_n - mean[tt] * mean[tt], 0.f) + p.eps1);
            }
            const int te = E >> 4, kge = (E & 15) >> 2, re = E & 3;
#pragma unroll
            for (int t = 0; t < ND; ++t) {
                const bool colok = 16 * t + 4 * kg_ < E;
                const unsigned co = colok ? (unsigned)(16 * t + 4 * kg_) * 4u : OOB;
                const f32x4 ga = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, co, 0, 0));
                const f32x4 be = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_e, co, 0, 0));
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x4 y = (acc2[tt][t] - mean[tt]) * rstd[tt] * ga + be;       // pad columns: gamma = beta = 0 -> exact zeros
                    if (t == te && kg_ == kge) y[re] = 1.0f;
                    u32x2 o;
                    o[0] = pack_bf16(y[0], y[1]);
                    o[1] = pack_bf16(y[2], y[3]);
                    const int seg = 2 * (t & 1) + rsg;
                    *reinterpret_cast<u32x2*>(wbase + (t >> 1) * SLAB + tt * 1024 + ((seg ^ rswz) * 16)) = o;
                }
            }
            zero_acc2();
            FSTAMP(6)                                  // 6: out_proj epilogue (and ReLU / pack)
        }
        for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < NT1; ++t) acc1[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            step(std::integral_constant<int, TP_FFN + 0>{}, pass);
            step(std::integral_constant<int, TP_FFN + 1>{}, pass);
            step(std::integral_constant<int, TP_FFN + 2>{}, pass);
            step(std::integral_constant<int, TP_FFN + 3>{}, pass);
            step(std::integral_constant<int, TP_FFN + 4>{}, pass);
            mfma_settle();
            // ReLU, round to bf16: tiles 2 kc and 2 kc + 1 side by side are the lane's 8 k slots of hidden chunk kc
            // (k = 32 kc + 16 a + 4 kg + r at slot 4 a + r: the order lime_ffn_pack_bf16 gives linear2's weight columns)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int kc = 0; kc < NT1 / 2; ++kc) {
                    u32x4 h;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const f32x4 v = acc_read4(acc1[tt][2 * kc + a]);
                        h[2 * a] = pack_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
                        h[2 * a + 1] = pack_bf16(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                    }
                    hb[tt][kc] = __builtin_bit_cast(bf16x8, h);
                }
            step(std::integral_constant<int, TP_FFN + 5>{}, pass);
            step(std::integral_constant<int, TP_FFN + 6>{}, pass);
            step(std::integral_constant<int, TP_FFN + 7>{}, pass);
            step(std::integral_constant<int, TP_FFN + 8>{}, pass);
        }
        FSTAMP(6)                                      // 6: ReLU / pack (and loop overhead)

        // + b2 + residual (this wave's rows of the stationary tile; column E is the bias column: not part of it) -- first, so that
        // the NEXT tile can stream into the image while the rest of the epilogue runs (every CU asks HBM for its 80 KB at about
        // the same time: ~9k cycles if nothing covers them).  LayerNorm over the E real columns (the pad columns are exact
        // zeros: zero weight rows, zero b2 / gamma / beta).
        const long row0 = (long)tile * BM;
        const int te = E >> 4, kge = (E & 15) >> 2, re = E & 3;
        const int le = lane_here();                    // as in load_x: epilogue-only offsets are computed here, not carried through the tile
        const int fi_ = le & 15, kg_ = le >> 4;
        const unsigned char* const rbase = lds + (32 * wave + fi_) * 64 + 8 * (kg_ & 1);
        const int rswz = swz4((fi_ >> 2) & 3), rsg = kg_ >> 1;
        const float* const cs_ = cs + 4 * kg_;
        f32x4 val[2][ND];                              // the accumulators leave the AccVGPRs here (lds_dma.h, acc_read)
        mfma_settle();
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(cs_ + 16 * t);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int seg = 2 * (t & 1) + rsg;
                f32x4 r = unpack_bf16x4(*reinterpret_cast<const u32x2*>(rbase + (t >> 1) * SLAB + tt * 1024 + ((seg ^ rswz) * 16)));
                if (t == te && kg_ == kge) r[re] = 0.f;
                val[tt][t] = acc_read4(acc2[tt][t]) + b + r;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the image has been read ... before the DMAs overwrite those rows
        if (!last) load_x(tile + (int)gridDim.x);
        FSTAMP(5)                                      // 5: residual added, next tile issued
#if defined(LIME_FFN_ABLATE) && LIME_FFN_ABLATE == 5       // no LayerNorm / pooling / stores
        if (p.eps != 12345.f) continue;
#endif
        float sum[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const f32x4 v = val[tt][t];
#pragma unroll
                for (int j = 0; j < 4; ++j) { sum[tt] += v[j]; sq[tt] += v[j] * v[j]; }
            }
        float mean[2], rstd[2];
        const float inv_n = 1.0f / (float)E;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float s1 = sum[tt], s2 = sq[tt];
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            mean[tt] = s1 * inv_n;
            rstd[tt] = rsqrtf(fmaxf(s2 * inv_n - mean[tt] * mean[tt], 0.f) + p.eps);
        }
        const float* const gs_ = cs_ + DP;
        const float* const es_ = cs_ + 2 * DP;
        if constexpr (POOL) {
            // block row (row0 + 32 wave) / 32: the column means over this wave's 32 tokens (all valid or all beyond M)
            const int rl0 = 32 * wave;
            const __amdgpu_buffer_rsrc_t rs_p = make_rsrc((char*)p.out + ((row0 + rl0) >> 5) * p.ldo * 4);
            const bool rows_ok = row0 + rl0 < M;
#pragma unroll
            for (int t = 0; t < ND; ++t) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gs_ + 16 * t);
                const f32x4 be = *reinterpret_cast<const f32x4*>(es_ + 16 * t);
                f32x4 y = (val[0][t] - mean[0]) * rstd[0] * ga + be;
                y += (val[1][t] - mean[1]) * rstd[1] * ga + be;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = row16_sum(y[j]) * (1.0f / 32.0f);
                buf_store4(y, rs_p, (rows_ok && fi_ == 0) ? (unsigned)(16 * t + 4 * kg_) * 4u : OOB, 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs_c = make_rsrc((char*)p.out + row0 * p.ldo * 2);
            unsigned cof[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int rl = 32 * wave + 16 * tt + fi_;
                cof[tt] = (row0 + rl < M) ? (unsigned)rl * (unsigned)p.ldo * 2u + (unsigned)kg_ * 8u : OOB;
            }
#pragma unroll
            for (int t = 0; t < ND; ++t) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gs_ + 16 * t);
                const f32x4 be = *reinterpret_cast<const f32x4*>(es_ + 16 * t);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) buf_store4_bf16((val[tt][t] - mean[tt]) * rstd[tt] * ga + be, rs_c, cof[tt] + (unsigned)t * 32u, 0);
            }
        }
        FSTAMP(7)                                      // 7: epilogue
    }
#ifdef LIME_STAMPS
    if (p.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.stamps[((long)blockIdx.x * 4 + wave) * 8 + i] = tsum[i];
    }
#endif
}

// The weights as the kernel's ring slots (bf16, every slot one contiguous block in the order of its LDS image):
//   w1p [F / 128 passes][10 chunks][128 rows][32 k]: W1[128 pass + row, 32 chunk + k] for k < E, b1 at k = E, zero behind (K = 320);
//   w2p [F / 32 chunks][304 rows][32]: zero rows n >= E; position 8 kg + 4 a + r of a row holds W2[n, 32 chunk + 16 a + 4 kg + r] --
//   the order in which the MFMA result registers of two neighbouring 16-column tiles (a = 0, 1) of the hidden state sit in a lane.
__global__ void ffn_pack_kernel(const float* __restrict__ w1, long ldw1, const float* __restrict__ b1, const float* __restrict__ w2, long ldw2,
                                int E, int F, uint16_t* __restrict__ w1p, uint16_t* __restrict__ w2p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n1 = (long)F * (NCH * 32), n2 = (long)F * DP;
    auto bf = [](float v) { return (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu); };
    if (i < n1) {
        const int kk = (int)(i & 31), row = (int)((i >> 5) % PW), c = (int)((i / (32 * PW)) % NCH), pass = (int)(i / (32L * PW * NCH));
        const int n = PW * pass + row, k = 32 * c + kk;
        w1p[i] = k < E ? bf(w1[n * ldw1 + k]) : (k == E ? bf(b1[n]) : (uint16_t)0);
    } else if (i < n1 + n2) {
        const long o = i - n1;
        const int s_ = (int)(o & 31), n = (int)((o >> 5) % DP), blk = (int)(o / (32 * DP));
        const int kg = s_ >> 3, a = (s_ >> 2) & 1, r = s_ & 3;
        w2p[o] = n < E ? bf(w2[n * ldw2 + 32 * blk + 16 * a + 4 * kg + r]) : (uint16_t)0;
    }
}

// out_proj weight fp32 [E, E] -> bf16 [10 chunks][304 rows][32 k]: zero rows n >= E, zero columns k >= E
__global__ void oproj_pack_kernel(const float* __restrict__ w, long ldw, int E, uint16_t* __restrict__ wp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)NCH * DP * 32) return;
    const int kk = (int)(i & 31), n = (int)((i >> 5) % DP), c = (int)(i / (32 * DP));
    const int k = 32 * c + kk;
    wp[i] = (n < E && k < E) ? (uint16_t)(pack_bf16(w[n * ldw + k], 0.f) & 0xFFFFu) : (uint16_t)0;
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

extern "C" int lime_ffn_pack_bf16(const float* w1, int64_t ldw1, const float* b1, const float* w2, int64_t ldw2, int32_t E, int32_t F,
                                  uint16_t* w1p, uint16_t* w2p, void* stream) {
    LIME_REQUIRE(w1 && b1 && w2 && w1p && w2p, LIME_ERR_BAD_ARG, "lime_ffn_pack_bf16: NULL pointer");
    LIME_REQUIRE(E > 0 && E < DP && E >= DP - 15 && F > 0 && F % PW == 0, LIME_ERR_UNSUPPORTED,
                 "lime_ffn_pack_bf16: built for %d <= E < %d (E = %d) and F a multiple of %d (F = %d)", DP - 15, DP, E, PW, F);
    LIME_REQUIRE(ldw1 >= E && ldw2 >= F, LIME_ERR_BAD_ARG, "lime_ffn_pack_bf16: leading dimension < row");
    const long n = (long)F * (NCH * 32) + (long)F * DP;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, (long)ldw1, b1, w2, (long)ldw2,
                       E, F, w1p, w2p);
    return lime_check_launch("lime_ffn_pack_bf16");
}

extern "C" int32_t lime_ffn_bf16_model_columns(void) { return DP; }
extern "C" int64_t lime_ffn_pack_bf16_size(int32_t F, int32_t which) { return which == 0 ? (int64_t)F * (NCH * 32) : (int64_t)F * DP; }

extern "C" int lime_encoder_ffn_bf16(const lime_ffn_bf16_args* a, void* stream) {
    LIME_REQUIRE(a != nullptr, LIME_ERR_BAD_ARG, "lime_encoder_ffn_bf16: args is NULL");
    LIME_REQUIRE(a->x && a->w1p && a->w2p && a->b2 && a->ln_gamma && a->ln_beta && a->out, LIME_ERR_BAD_ARG, "lime_encoder_ffn_bf16: NULL pointer");
    LIME_REQUIRE(a->M >= 0 && a->E > 0 && a->E < DP && a->E >= DP - 15 && a->F > 0 && a->F % PW == 0, LIME_ERR_UNSUPPORTED,
                 "lime_encoder_ffn_bf16: built for %d <= E < %d (E = %d: one spare zero column carries linear1's bias) and F a multiple of %d (F = %d)",
                 DP - 15, DP, a->E, PW, a->F);
    LIME_REQUIRE(a->ldx >= DP && a->ldx % 8 == 0 && (uintptr_t)a->x % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_ffn_bf16: x rows must hold %d bf16 columns, 16-byte aligned (ldx %% 8 == 0)", DP);
    LIME_REQUIRE((uintptr_t)a->w1p % 16 == 0 && (uintptr_t)a->w2p % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_ffn_bf16: packed weights must be 16-byte aligned (see lime_ffn_pack_bf16)");
    LIME_REQUIRE(a->pool32 == 0 || a->pool32 == 1, LIME_ERR_BAD_ARG, "lime_encoder_ffn_bf16: pool32 must be 0 / 1");
    if (a->pool32)
        LIME_REQUIRE(a->M % 32 == 0 && a->ldo >= DP && a->ldo % 4 == 0 && (uintptr_t)a->out % 16 == 0, LIME_ERR_BAD_ARG,
                     "lime_encoder_ffn_bf16: pool32 needs M %% 32 == 0 and fp32 out rows of >= %d columns, 16-byte aligned", DP);
    else
        LIME_REQUIRE(a->ldo >= DP && a->ldo % 4 == 0 && (uintptr_t)a->out % 8 == 0, LIME_ERR_BAD_ARG,
                     "lime_encoder_ffn_bf16: bf16 out rows of >= %d columns, 8-byte aligned", DP);
    const long lim = 0x7FFFFFF0L;
    LIME_REQUIRE(128L * a->ldx * 2 < lim && (long)a->F * 320 * 2 < lim && 128L * a->ldo * 4 < lim,
                 LIME_ERR_UNSUPPORTED, "lime_encoder_ffn_bf16: operand too large for 32-bit offsets");
    if (a->M == 0) return LIME_OK;
    FfnP p{};
    p.x = a->x; p.ldx = a->ldx; p.w1p = a->w1p; p.w2p = a->w2p;
    p.b2 = a->b2; p.g = a->ln_gamma; p.beta = a->ln_beta; p.eps = a->ln_eps;
    p.out = a->out; p.ldo = a->ldo; p.M = a->M; p.E = a->E; p.F = a->F; p.m_dev = a->m_dev;
#ifdef LIME_STAMPS
    p.stamps = g_ffn_stamp_buf;
#endif
    const long ntiles = ((long)a->M + BM - 1) / BM;
    long nwg = num_cus();
    if (nwg > ntiles) nwg = ntiles;
    hipStream_t s = (hipStream_t)stream;
    if (a->pool32) hipLaunchKernelGGL((ffn_bf16_kernel<true, false>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((ffn_bf16_kernel<false, false>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return lime_check_launch("lime_encoder_ffn_bf16");
}

extern "C" int64_t lime_oproj_pack_bf16_size(void) { return (int64_t)NCH * DP * 32; }

extern "C" int lime_oproj_pack_bf16(const float* w, int64_t ldw, int32_t E, uint16_t* wp, void* stream) {
    LIME_REQUIRE(w && wp, LIME_ERR_BAD_ARG, "lime_oproj_pack_bf16: NULL pointer");
    LIME_REQUIRE(E > 0 && E < DP && E >= DP - 15 && ldw >= E, LIME_ERR_UNSUPPORTED, "lime_oproj_pack_bf16: built for %d <= E < %d (E = %d)", DP - 15, DP, E);
    const long n = (long)NCH * DP * 32;
    hipLaunchKernelGGL(oproj_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (long)ldw, E, wp);
    return lime_check_launch("lime_oproj_pack_bf16");
}

extern "C" int lime_encoder_block_bf16(const lime_encoder_block_bf16_args* a, void* stream) {
    LIME_REQUIRE(a != nullptr, LIME_ERR_BAD_ARG, "lime_encoder_block_bf16: args is NULL");
    LIME_REQUIRE(a->attn && a->w0p && a->add_rows && a->ln1_gamma && a->ln1_beta && a->res && a->w1p && a->w2p && a->b2 && a->ln2_gamma &&
                 a->ln2_beta && a->out, LIME_ERR_BAD_ARG, "lime_encoder_block_bf16: NULL pointer");
    LIME_REQUIRE(a->M >= 0 && a->E > 0 && a->E < DP && a->E >= DP - 15 && a->E % 4 == 0 && a->F > 0 && a->F % PW == 0, LIME_ERR_UNSUPPORTED,
                 "lime_encoder_block_bf16: built for %d <= E < %d, E %% 4 == 0 (E = %d) and F a multiple of %d (F = %d)", DP - 15, DP, a->E, PW, a->F);
    LIME_REQUIRE(a->lda >= DP && a->lda % 8 == 0 && (uintptr_t)a->attn % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_block_bf16: attn rows must hold %d bf16 columns, 16-byte aligned (lda %% 8 == 0)", DP);
    LIME_REQUIRE((uintptr_t)a->w0p % 16 == 0 && (uintptr_t)a->w1p % 16 == 0 && (uintptr_t)a->w2p % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_block_bf16: packed weights must be 16-byte aligned");
    LIME_REQUIRE((uintptr_t)a->ln1_gamma % 16 == 0 && (uintptr_t)a->ln1_beta % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_block_bf16: norm1 parameters must be 16-byte aligned");
    LIME_REQUIRE(a->add_period > 0 && a->ld_add >= a->E && a->ld_add % 4 == 0 && (uintptr_t)a->add_rows % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_block_bf16: add_rows must be fp32 [add_period >= 1, >= E], 16-byte aligned rows");
    LIME_REQUIRE(a->res_kind == 2 || a->res_kind == 3, LIME_ERR_BAD_ARG, "lime_encoder_block_bf16: res_kind must be 2 (rows gathered by res_ids) or 3 (rows)");
    LIME_REQUIRE(a->ldr >= DP && a->ldr % 4 == 0 && (uintptr_t)a->res % 8 == 0, LIME_ERR_BAD_ARG,
                 "lime_encoder_block_bf16: residual rows must hold %d bf16 columns, 8-byte aligned", DP);
    if (a->res_kind == 2) LIME_REQUIRE(a->res_ids != nullptr, LIME_ERR_BAD_ARG, "lime_encoder_block_bf16: the gathered residual needs res_ids");
    LIME_REQUIRE(a->pool32 == 0 || a->pool32 == 1, LIME_ERR_BAD_ARG, "lime_encoder_block_bf16: pool32 must be 0 / 1");
    if (a->pool32)
        LIME_REQUIRE(a->M % 32 == 0 && a->ldo >= DP && a->ldo % 4 == 0 && (uintptr_t)a->out % 16 == 0, LIME_ERR_BAD_ARG,
                     "lime_encoder_block_bf16: pool32 needs M %% 32 == 0 and fp32 out rows of >= %d columns, 16-byte aligned", DP);
    else
        LIME_REQUIRE(a->ldo >= DP && a->ldo % 4 == 0 && (uintptr_t)a->out % 8 == 0, LIME_ERR_BAD_ARG,
                     "lime_encoder_block_bf16: bf16 out rows of >= %d columns, 8-byte aligned", DP);
    const long lim = 0x7FFFFFF0L;
    LIME_REQUIRE(128L * a->lda * 2 < lim && (long)a->F * 320 * 2 < lim && 128L * a->ldo * 4 < lim && 128L * a->ldr * 2 < lim &&
                 (a->res_kind == 3 ? (long)a->M : (long)a->res_rows) * a->ldr * 2 < lim && (long)a->add_period * a->ld_add * 4 < lim,
                 LIME_ERR_UNSUPPORTED, "lime_encoder_block_bf16: operand too large for 32-bit offsets");
    if (a->M == 0) return LIME_OK;
    FfnP p{};
    p.x = a->attn; p.ldx = a->lda; p.w1p = a->w1p; p.w2p = a->w2p;
    p.b2 = a->b2; p.g = a->ln2_gamma; p.beta = a->ln2_beta; p.eps = a->ln2_eps;
    p.out = a->out; p.ldo = a->ldo; p.M = a->M; p.E = a->E; p.F = a->F; p.m_dev = a->m_dev;
    p.w0p = a->w0p; p.g1 = a->ln1_gamma; p.beta1 = a->ln1_beta; p.eps1 = a->ln1_eps;
    p.res_kind = a->res_kind; p.res = a->res; p.ldr = a->ldr; p.res_ids = a->res_ids; p.res_pe = a->add_rows; p.ldpe = a->ld_add;
    p.res_period = a->add_period;
#ifdef LIME_STAMPS
    p.stamps = g_ffn_stamp_buf;
#endif
    const long ntiles = ((long)a->M + BM - 1) / BM;
    long nwg = num_cus();
    if (nwg > ntiles) nwg = ntiles;
    hipStream_t s = (hipStream_t)stream;
    if (a->pool32) hipLaunchKernelGGL((ffn_bf16_kernel<true, true>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((ffn_bf16_kernel<false, true>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return lime_check_launch("lime_encoder_block_bf16");
}
