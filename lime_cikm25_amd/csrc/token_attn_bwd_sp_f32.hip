// lime_token_attention_bwd_f32 for 64 < S <= 128 with ALL FIVE products on the bf16 matrix cores as split products (split_mfma.h:
// three bf16 terms per fp32 value, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block, fp32 accumulation).  gfx950 only.
//
// token_attn_bwd_kernel (backward_f32.hip) keeps P / dS as an fp32 image in LDS and takes dV, dQ, dK from v_mfma_f32_16x16x4_f32
// (256 matrix cycles per 16 x 16 x 32 block against 96 here); three workgroup barriers per (sequence, head).  Here a 16 x 16 result
// tile never leaves the registers before it is an operand again.  The MFMA result layout puts a tile's COLUMN on the lane (fi) and
// its ROWS on the lane group and the registers (4 kg + r), and the k index of the next MFMA is only a summation label: registers
// of two row tiles ARE the eight k values of a B operand, for a product that sums over the tile's rows and keeps its column.  So
// the scores are computed in both orientations, each by the wave that owns the column:
//   phase A (a wave's 16 queries i on the columns, all keys j on the rows):
//       S^T = K Q^T, dP^T = V dO^T  ->  softmax over j and delta_i = sum_j P~ dP in the lane + two shuffles  ->  dS^T
//       dQ^T[d, i] = sum_j K^T[d, j] dS^T[j, i]                                         (K^T: transposed LDS reads, below)
//   phase B (a wave's 16 keys j on the columns, all queries i on the rows; row maximum, 1 / sum and delta of phase A through LDS):
//       S = Q K^T, dP = dO V^T  ->  P~, dS
//       dV^T[d, j] = sum_i dO^T[d, i] P~[i, j],   dK^T[d, j] = sum_i Q^T[d, i] dS[i, j]
// Seven products instead of five, all at the split rate (56 blocks x 96 cycles per wave against 40 x 256), ONE barrier between
// the phases, no P image.  Q, K, V, dO are split once per (sequence, head) on their way into LDS: three bf16 images [128][32]
// each (64-byte rows with swizzled 16-byte chunks: split_mfma.h).  The row-major image serves both operand kinds: as rows (ds_read_b128:
// eight consecutive head dims of one token) for the score products, and TRANSPOSED for the gradient products through ds_read_b64_tr_b16 -- per 16-lane group a block
// of four token rows x 16 head dims, delivered column-major: lane fi gets head dim fi of rows 4 kg .. 4 kg + 3 of a 16-row tile,
// which is exactly the k order the result registers carry.  The results come out transposed (head dims on the registers): a lane
// stores four consecutive head dims of its token, 16 bytes.
// The attention-probability dropout mask is regenerated from the element index as in the fp32 kernel (same seed / site / index).
#include "common.h"
#include "gemm_pp.h"
#include "lds_dma.h"
#include "split_mfma.h"
#include "dropout.h"

using namespace lime_dev;

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int SPB = 128;                              // rows per image
constexpr int TERM = SPB * SWZ_ROW;                   // one bf16 image (unsigned shorts): 64-byte rows
static_assert(SWZ_ROW == 32, "images hold 32 head dims per row");
constexpr int IMG = 3 * TERM;                         // the three terms of one operand
constexpr int LDS_BYTES = 4 * IMG * 2 + 3 * SPB * 4;  // Q, K, V, dO + the row statistics

struct BwdSpP {
    const float* q; const float* k; const float* v; long ld;
    const float* dout; long ldo;
    float* dq; float* dk; float* dv; long ldd;
    int n_seq, S, n_head, head_dim; float scale;
    LimeDropout drop;
};

// Image layout and its readers (swz_row_load, swz_tr_load, swz_store*): split_mfma.h.
__device__ __forceinline__ SplitFrag row_load(const unsigned short* img, int r, int kg) { return swz_row_load(img, TERM, r, kg); }
__device__ __forceinline__ SplitFrag tr_load(const unsigned short* img, int t0, int c, int fi, int kg) { return swz_tr_load(img, TERM, t0, c, fi, kg); }

template <bool FULL>      // S == 128: no rows beyond S to mask
__global__ __launch_bounds__(512) void attn_bwd_sp_kernel(const BwdSpP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* const Qi = smem;
    unsigned short* const Ki = Qi + IMG;
    unsigned short* const Vi = Ki + IMG;
    unsigned short* const Oi = Vi + IMG;
    float* const st_mx = reinterpret_cast<float*>(Oi + IMG);
    float* const st_inv = st_mx + SPB;
    float* const st_dl = st_inv + SPB;

    const int tid0 = threadIdx.x;
    const int S = p.S;
    const long n_prob = (long)p.n_seq * p.n_head;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float c2 = p.scale * LOG2E;
    const bool dropping = p.drop.thresh != 0;
    constexpr bool full = FULL;

    // the next problem's rows travel in registers while this one computes: q / k / v rows are 32 floats on 16-byte boundaries (heads
    // 32 columns apart, zero pad columns), dO rows head_dim floats on 8-byte boundaries
    f32x4 rq[2], rk[2], rv[2];
    f32x2 ro[4];
    auto fetch = [&](long prob) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                 // (addresses recomputed per problem, see below)
        const int seq = (int)(prob / p.n_head), head = (int)(prob % p.n_head);
        const long row_base = (long)seq * S;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u, r = e >> 3, c = (e & 7) * 4;
            const bool ok = r < S;
            const long g = (row_base + (ok ? r : 0)) * p.ld + (long)head * 32 + c;
            rq[u] = ok ? *reinterpret_cast<const f32x4*>(p.q + g) : z4;
            rk[u] = ok ? *reinterpret_cast<const f32x4*>(p.k + g) : z4;
            rv[u] = ok ? *reinterpret_cast<const f32x4*>(p.v + g) : z4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 512 * u, r = e >> 4, c = (e & 15) * 2;
            f32x2 d = {0.f, 0.f};
            if (r < S && c < p.head_dim) d = *reinterpret_cast<const f32x2*>(p.dout + (row_base + r) * p.ldo + (long)head * p.head_dim + c);
            ro[u] = d;
        }
    };
    auto commit = [&]() {                             // registers -> the split images
        int tid = tid0;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u, r = e >> 3, c = (e & 7) * 4;
            swz_store4(Qi, TERM, r, c, rq[u][0], rq[u][1], rq[u][2], rq[u][3]);
            swz_store4(Ki, TERM, r, c, rk[u][0], rk[u][1], rk[u][2], rk[u][3]);
            swz_store4(Vi, TERM, r, c, rv[u][0], rv[u][1], rv[u][2], rv[u][3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 512 * u, r = e >> 4, c = (e & 15) * 2;
            swz_store2(Oi, TERM, r, c, ro[u][0], ro[u][1]);
        }
    };

    if ((long)blockIdx.x < n_prob) fetch(blockIdx.x);
    for (long prob = blockIdx.x; prob < n_prob; prob += gridDim.x) {
        const int seq = (int)(prob / p.n_head), head = (int)(prob % p.n_head);
        commit();
        lds_barrier();                                // (LDS only: the dq / dk / dv stores and the prefetch stay in flight)
        if (prob + gridDim.x < n_prob) fetch(prob + gridDim.x);
        // per-lane offsets are recomputed per problem from a laundered lane id: hoisted out of this loop they are ~100 registers of
        // loop-invariant addresses and mask indices (spilled)
        int tl = tid0;
        asm volatile("" : "+v"(tl));
        int lane = tl & 63;
        const int X0 = 16 * __builtin_amdgcn_readfirstlane(tl >> 6);      // this wave's 16 queries (phase A) / keys (phase B)

        // ================= phase A: this wave's queries on the columns =================================================
        // Every loop body pairs one tile's MFMAs with the arithmetic of the tile before it (independent instruction streams in one
        // scheduling region): the two waves of a SIMD run the same code in step, so whatever is not interleaved inside a wave is serial.
        {
            const int fi = lane & 15, kg = lane >> 4;
            const SplitFrag qB = row_load(Qi, X0 + fi, kg), oB = row_load(Oi, X0 + fi, kg);
            f32x4 st[8], dt[8];                       // S^T -> exp2 terms, dP^T -> dS^T: [key 16 t + 4 kg + r][query X0 + fi]
            SplitFrag fa[2];                          // the next tile's fragments are requested before this tile's MFMAs
            // ---- S^T = K Q^T and the row maximum (on the raw scores: scale > 0)
            float mxr = -INFINITY;
            auto tile_max = [&](int t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) mxr = fmaxf(mxr, (full || 16 * t + 4 * kg + r < S) ? st[t][r] : -INFINITY);
            };
            fa[0] = row_load(Ki, fi, kg);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t + 1 < 8) fa[(t + 1) & 1] = row_load(Ki, 16 * (t + 1) + fi, kg);
                st[t] = split_mfma16(fa[t & 1], qB, z4);
                if (t > 0) tile_max(t - 1);
                __builtin_amdgcn_sched_barrier(0);    // (unrolled: hipcc otherwise hoists all the fragments, 192 registers)
            }
            tile_max(7);
            mxr = fmaxf(mxr, __shfl_xor(mxr, 16)); mxr = fmaxf(mxr, __shfl_xor(mxr, 32));
            const float mx = mxr * c2, nmx = -mx;
            // ---- dP^T = V dO^T;  e = exp2(s - max), sum e, sum e dP~  (dP~ = keep * dP / (1 - p): the forward multiplied keep * P / (1 - p) into V)
            const uint64_t mrow = ((uint64_t)prob * S + (uint64_t)(X0 + fi)) * (uint64_t)S;
            float sum = 0.f, dlu = 0.f;
            auto tile_exp = [&](int t) {
                const unsigned keep = dropping ? lime_keep4(p.drop, (mrow + (uint64_t)(16 * t + 4 * kg)) >> 2) : 0xFu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (full || 16 * t + 4 * kg + r < S) ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], c2, nmx)) : 0.f;
                    const float dpv = dropping ? ((keep >> r) & 1u ? dt[t][r] * p.drop.scale : 0.f) : dt[t][r];
                    st[t][r] = e;
                    dt[t][r] = dpv;
                    sum += e;
                    dlu = __builtin_fmaf(e, dpv, dlu);
                }
            };
            fa[0] = row_load(Vi, fi, kg);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t + 1 < 8) fa[(t + 1) & 1] = row_load(Vi, 16 * (t + 1) + fi, kg);
                dt[t] = split_mfma16(fa[t & 1], oB, z4);
                if (t > 0) tile_exp(t - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_exp(7);
            sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
            dlu += __shfl_xor(dlu, 16); dlu += __shfl_xor(dlu, 32);
            const float inv = 1.0f / sum, dl = dlu * inv, si = p.scale * inv, ndl = -dl * si;
            if (kg == 0) { st_mx[X0 + fi] = mx; st_inv[X0 + fi] = inv; st_dl[X0 + fi] = dl * p.scale; }
            // ---- dS^T = scale P (dP~ - delta);  dQ^T[d, i] = sum_j K^T[d, j] dS^T[j, i]: a step's twelve MFMAs beside the next step's
            // dS^T and its split
            auto pair_ds = [&](int s2) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = st[2 * s2 + (e >> 2)][e & 3] * __builtin_fmaf(dt[2 * s2 + (e >> 2)][e & 3], si, ndl);
                return split_frag(x);
            };
            f32x4 aq0 = z4, aq1 = z4;
            SplitFrag b[2];
            b[0] = pair_ds(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const SplitFrag a0 = tr_load(Ki, 2 * s, 0, fi, kg), a1 = tr_load(Ki, 2 * s, 1, fi, kg);
                if (s + 1 < 4) b[(s + 1) & 1] = pair_ds(s + 1);
                aq0 = split_mfma16(a0, b[s & 1], aq0);
                aq1 = split_mfma16(a1, b[s & 1], aq1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (X0 + fi < S) {
                float* const d = p.dq + ((long)seq * S + X0 + fi) * p.ldd + (long)head * 32 + 4 * kg;
                *reinterpret_cast<f32x4*>(d) = aq0;
                *reinterpret_cast<f32x4*>(d + 16) = aq1;
            }
        }
        lds_barrier();                                // every query's statistics are in LDS

        // ================= phase B: this wave's keys on the columns ====================================================
        {
            asm volatile("" : "+v"(lane));
            const int fi = lane & 15, kg = lane >> 4;
            const SplitFrag kB = row_load(Ki, X0 + fi, kg), vB = row_load(Vi, X0 + fi, kg);
            f32x4 pt[8], ds[8];                       // S -> P~ and dP -> dS: [query 16 t + 4 kg + r][key X0 + fi]
            const bool key_ok = X0 + fi < S;
            const int jq = (X0 + fi) >> 2, jb = fi & 3;
            auto tile_pds = [&](int t) {
                const f32x4 mx4 = *reinterpret_cast<const f32x4*>(st_mx + 16 * t + 4 * kg);
                const f32x4 inv4 = *reinterpret_cast<const f32x4*>(st_inv + 16 * t + 4 * kg);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(st_dl + 16 * t + 4 * kg);
                // the mask is hashed per four consecutive KEYS of one query: the four lanes of a quad (keys 4 jq .. 4 jq + 3) need the same
                // four hashes (queries r = 0..3) -- each lane computes the one of query r = its position, the quad exchanges them by DPP
                unsigned k0 = 0xFu, k1 = 0xFu, k2 = 0xFu, k3 = 0xFu;
                if (dropping) {
                    const uint64_t idx4 = (((uint64_t)prob * S + (uint64_t)(16 * t + 4 * kg + jb)) * (uint64_t)S >> 2) + (uint64_t)jq;
                    const int mine = (int)lime_keep4(p.drop, idx4);
                    k0 = (unsigned)__builtin_amdgcn_mov_dpp(mine, 0x00, 0xF, 0xF, true);      // quad_perm [0, 0, 0, 0]
                    k1 = (unsigned)__builtin_amdgcn_mov_dpp(mine, 0x55, 0xF, 0xF, true);      // [1, 1, 1, 1]
                    k2 = (unsigned)__builtin_amdgcn_mov_dpp(mine, 0xAA, 0xF, 0xF, true);      // [2, 2, 2, 2]
                    k3 = (unsigned)__builtin_amdgcn_mov_dpp(mine, 0xFF, 0xF, 0xF, true);      // [3, 3, 3, 3]
                }
                const unsigned kr[4] = {k0, k1, k2, k3};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = (full || key_ok) ? __builtin_amdgcn_exp2f(__builtin_fmaf(pt[t][r], c2, -mx4[r])) * inv4[r] : 0.f;
                    const float f = dropping ? ((kr[r] >> jb) & 1u ? p.drop.scale : 0.f) : 1.f;
                    const float dpv = dropping ? ds[t][r] * f : ds[t][r];
                    pt[t][r] = dropping ? pv * f : pv;
                    ds[t][r] = pv * __builtin_fmaf(dpv, p.scale, -dl4[r]);        // st_dl holds scale * delta
                }
            };
            SplitFrag qA[2], oA[2];
            qA[0] = row_load(Qi, fi, kg); oA[0] = row_load(Oi, fi, kg);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t + 1 < 8) { qA[(t + 1) & 1] = row_load(Qi, 16 * (t + 1) + fi, kg); oA[(t + 1) & 1] = row_load(Oi, 16 * (t + 1) + fi, kg); }
                pt[t] = split_mfma16(qA[t & 1], kB, z4);
                ds[t] = split_mfma16(oA[t & 1], vB, z4);
                if (t > 0) tile_pds(t - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_pds(7);
            // dV^T[d, j] = sum_i dO^T[d, i] P~[i, j];  dK^T[d, j] = sum_i Q^T[d, i] dS[i, j]
            f32x4 av0 = z4, av1 = z4, ak0 = z4, ak1 = z4;
            SplitFrag b[2];
            b[0] = split_frag(pt[0], pt[1]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const SplitFrag o0 = tr_load(Oi, 2 * s, 0, fi, kg), o1 = tr_load(Oi, 2 * s, 1, fi, kg);
                b[(s + 1) & 1] = s + 1 < 4 ? split_frag(pt[2 * s + 2], pt[2 * s + 3]) : split_frag(ds[0], ds[1]);
                av0 = split_mfma16(o0, b[s & 1], av0);
                av1 = split_mfma16(o1, b[s & 1], av1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const SplitFrag q0 = tr_load(Qi, 2 * s, 0, fi, kg), q1 = tr_load(Qi, 2 * s, 1, fi, kg);
                if (s + 1 < 4) b[(s + 1) & 1] = split_frag(ds[2 * s + 2], ds[2 * s + 3]);
                ak0 = split_mfma16(q0, b[s & 1], ak0);
                ak1 = split_mfma16(q1, b[s & 1], ak1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (key_ok) {
                const long o = ((long)seq * S + X0 + fi) * p.ldd + (long)head * 32 + 4 * kg;
                *reinterpret_cast<f32x4*>(p.dv + o) = av0;
                *reinterpret_cast<f32x4*>(p.dv + o + 16) = av1;
                *reinterpret_cast<f32x4*>(p.dk + o) = ak0;
                *reinterpret_cast<f32x4*>(p.dk + o + 16) = ak1;
            }
        }
        lds_barrier();                                // the images are free for the next problem
    }
}

}  // namespace

// LIME_PP_NOT_APPLICABLE: the caller takes token_attn_bwd_kernel (shapes / alignments outside this build)
int lime_token_attention_bwd_sp(const float* q, const float* k, const float* v, long ld, const float* dout, long ldo, float* dq, float* dk,
                                float* dv, long ldd, int n_seq, int S, int n_head, int head_dim, int head_stride, float scale,
                                const LimeDropout& drop, hipStream_t s) {
    if (!(S > 64 && S <= SPB && S % 4 == 0 && head_stride == 32 && head_dim <= 32 && head_dim % 2 == 0)) return LIME_PP_NOT_APPLICABLE;
    if (ld % 4 != 0 || ldd % 4 != 0 || ldo % 2 != 0) return LIME_PP_NOT_APPLICABLE;
    if (((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) != 0 || (((uintptr_t)dout) & 7) != 0)
        return LIME_PP_NOT_APPLICABLE;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
    static bool configured = false;
    static int n_cu = 256;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_sp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_sp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_bwd_f32: cannot reserve %d bytes of LDS: %s", LDS_BYTES,
                     hipGetErrorString(e));
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n_cu = cus;
        configured = true;
    }
    const long n_prob = (long)n_seq * n_head;
    BwdSpP p{q, k, v, ld, dout, ldo, dq, dk, dv, ldd, n_seq, S, n_head, head_dim, scale, drop};
    const unsigned grid = (unsigned)(n_prob < n_cu ? n_prob : n_cu);             // persistent: one workgroup per CU
    if (S == SPB) attn_bwd_sp_kernel<true><<<grid, 512, LDS_BYTES, s>>>(p);
    else attn_bwd_sp_kernel<false><<<grid, 512, LDS_BYTES, s>>>(p);
    return lime_check_launch("attn_bwd_sp_kernel");
}
