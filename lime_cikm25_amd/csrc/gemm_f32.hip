// lime_linear_f32: C = epilogue(A . W^T + bias) on the exact-fp32 matrix cores of gfx950.
//
// Every nn.Linear of the scoring path goes through this kernel (see include/lime_hip.h for the list
// of reference call sites).  Design, MI355X-first:
//   * v_mfma_f32_32x32x2_f32 (64 cycles / SIMD, bit-exact fp32 fma chain) -- the path's tolerance is
//     1e-3 against an fp32 CPU forward, and gfx950 has no xf32/TF32 mode, so the fp32 roof is the
//     157 TFLOP/s matrix rate, 16x below bf16.  At that rate one ds_read_b128 feeds four MFMAs and
//     LDS / L2 bandwidth are far from binding; what binds is keeping the MFMA pipe issued.
//   * the K loops of this path are short (K = 300 ... 512: 10 ... 16 chunks of 32), so a one-tile-per-
//     workgroup kernel spends a third of its life in prologue / epilogue latency (measured: 65 TF).
//     The kernel is therefore PERSISTENT: a workgroup walks output tiles t, t + G, t + 2G ... as one
//     continuous stream of K chunks -- the global->register prefetch of the next chunk (which may
//     belong to the next tile) is always issued before the MFMAs of the current one, the epilogue's
//     stores retire under the next tile's MFMAs, and the residual operand is loaded straight into
//     the accumulators at the tile boundary instead of being fetched row by row in the epilogue.
//   * both operands are K-contiguous row-major ([rows][K]: activations and nn.Linear weights as
//     stored), staged as [rows][36] fp32 so that a wave's ds_read_b128 (16-lane groups, 64 banks) is
//     conflict free; lane (i, h) fetches k = 8s+4h .. 8s+4h+3 and the four MFMAs of step s pair the
//     same k on both operands (the MFMA's k index is only a summation label).  Fragments of step
//     s + 1 are read while the MFMAs of step s execute.
//   * the word-embedding gather + positional table is fused into the A fetch (and into the residual
//     of the out_proj epilogue), bias / ReLU / tanh / residual / gated residual / LayerNorm into the
//     epilogue, so the encoder layer's activations cross HBM once per GEMM, not once per op.
//   * XCD-aware bijective remap of the workgroup id: an XCD walks whole row panels, so the N-tiles that
//     re-read one A panel hit the same 4 MiB L2.
#include "common.h"
#include "gemm_pp.h"
#include <stdlib.h>

#ifdef LIME_STAMPS
// Diagnostic build only (tools/gemm_stamps.py): per-wave s_memtime sums of the main-loop segments.  The stamp values go to
// a buffer nothing else reads; no output depends on them.  Never compiled into liblime_hip.so.
static unsigned long long* g_stamp_buf = nullptr;
extern "C" void lime_debug_set_stamp_buffer(unsigned long long* p) { g_stamp_buf = p; }
#define LIME_NSEG 8
#define STAMP(i)                                                            \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 \
        tsum[i] += t_ - tlast;                                              \
        tlast = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
    }
#else
#define STAMP(i)
#endif

namespace {

constexpr int BK = 32;    // K depth of one staged chunk
constexpr int LDK = 36;   // LDS row pitch in floats: 144 B, conflict-free for ds_read_b128, 16-B aligned

struct GemmP {
    const float* a; long lda; const int* a_ids; const float* a_pe; long lda_pe; int a_period;
    const float* w; long ldw; const float* bias;
    const float* res; long ldr; int res_div; int res_mod; const int* res_ids; const float* res_pe; long ldr_pe; int res_period;
    const float* ln_g; const float* ln_b; float ln_eps; float* ln_rstd;
    float* c; long ldc; int M, N, K; int act;
    int n_row_blocks, n_col_blocks;
#ifdef LIME_STAMPS
    unsigned long long* stamps;
#endif
    int res_in_acc;      // residual is loaded into the accumulators at the tile boundary (plain add, no act, no gate)
};

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T vzero() {
    typename VecT<VEC>::T z;
    if constexpr (VEC == 1) z = 0.f; else { for (int i = 0; i < VEC; ++i) z[i] = 0.f; }
    return z;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == LIME_ACT_RELU) return fmaxf(v, 0.f);
    if (act == LIME_ACT_TANH) return tanhf(v);
    if (act == LIME_ACT_SIGMOID) return lime_sigmoid(v);
    return v;
}

// ---- buffer addressing ----------------------------------------------------------------------------------
// Every global access goes through a buffer descriptor: a wave-uniform base (SGPRs) plus a 32-bit lane offset,
// so a row costs one VGPR instead of a 64-bit pointer pair, and an out-of-range offset reads as zero / drops
// the store in hardware -- rows beyond M, columns beyond N and k beyond K need no select and no branch.
constexpr unsigned OOB = 0x80000000u;          // >= num_records of every descriptor below
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    if constexpr (VEC == 4) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    else if constexpr (VEC == 2) return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// TM x TN MFMA tiles (32x32) per wave, WM x WN waves per workgroup (4 or 8 waves).
// GENERIC (small 64x64 tiles): activation and residual placement are run-time switches; output rows sit in the
//      accumulator registers (D = A W^T as the MFMA's C layout has it).
// Big tiles (!GENERIC) compute the TRANSPOSED product D^T = W A^T (weights as the MFMA A operand): an output ROW is
//      then a lane, and accumulator registers 4g .. 4g+3 are four consecutive COLUMNS of it -- residual loads and result
//      stores are 16-byte vector accesses (20 instead of 80 per lane and tile), and the LayerNorm statistics are a sum
//      over a lane's own registers plus one cross-half shuffle.
//   LN:  a tile spans every output column (n_col_blocks == 1) and the epilogue applies LayerNorm; the statistics of
//        the WN waves that share a row are combined through a small LDS exchange.  The residual (if any) is loaded
//        straight into the accumulators at the tile boundary.
//   ACT: compile-time activation of the non-LN big tiles (none / ReLU).
//   VIO: 16-byte residual loads / result stores (needs N, ldc, ldr multiples of 4 and aligned bases).
// PE:  the A operand is a gather with a positional table added (a_ids and a_pe both set).
template <int TM, int TN, int WM, int WN, int VEC, bool LN, bool PE, int ACT, bool GENERIC, bool VIO>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && TM * TN <= 4) ? 2 : 1) void gemm_f32_kernel(const GemmP p) {
    constexpr int NT = WM * WN * 64;       // threads per workgroup
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int TPR = BK / VEC;          // threads per staged row
    constexpr int RPP = NT / TPR;          // rows per staging pass
    constexpr int APASS = BM / RPP, WPASS = BN / RPP;
    static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "staging passes must tile the block");
    typedef typename VecT<VEC>::T vec_t;

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDK];
    __shared__ float red_sum[LN ? WN * BM : 1];                  // LayerNorm partials of the WN waves sharing a row
    __shared__ float red_sq[LN ? WN * BM : 1];
    __shared__ __attribute__((aligned(16))) float Bs[GENERIC ? 4 : BN];          // bias of the current tile's columns
    __shared__ __attribute__((aligned(16))) float Gs[LN ? BN : 4], Es[LN ? BN : 4];   // LayerNorm gamma / beta

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.n_row_blocks * p.n_col_blocks;
    const int nwg = gridDim.x;
    const int first = xcd_remap(blockIdx.x, nwg);
    if (first >= ntiles) return;
    const int nchunk = (p.K + BK - 1) / BK;
    const int tail_steps = ((p.K - (nchunk - 1) * BK) + 7) / 8;      // k8 steps of a tile's last chunk (1..4)
    const int lda4 = (int)p.lda * 4, ldw4 = (int)p.ldw * 4, ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4;
    const bool gather = p.a_ids != nullptr;

    auto tile_rc = [&](int tile, long& row0, int& col0) {
        const int rb = tile / p.n_col_blocks;
        row0 = (long)rb * BM;
        col0 = (tile - rb * p.n_col_blocks) * BN;
    };

    // ---- loader state: the tile whose chunks are being fetched -----------------------------------
    // Loads carry no arithmetic, so they stay in flight under the MFMAs; the positional add happens when the
    // registers are committed to LDS.
    const int srow = tid / TPR, sk = (tid % TPR) * VEC;
    __amdgpu_buffer_rsrc_t rs_a, rs_w;
    const __amdgpu_buffer_rsrc_t rs_pe = make_rsrc(p.a_pe ? p.a_pe : p.a);
    unsigned a_voff[APASS];                // byte offset of this lane's row (pass i) inside rs_a, k = sk; OOB if the row is out of range
    unsigned pe_voff[PE ? APASS : 1];
    unsigned w_voff[WPASS];
    int ids_next[APASS];                   // gather ids of the tile after the loader's, fetched one tile ahead
    auto prefetch_ids = [&](int tile) {
        if (!gather) return;
        long row0; int col0;
        tile_rc(tile < ntiles ? tile : first, row0, col0);
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const long r = row0 + srow + i * RPP;
            ids_next[i] = p.a_ids[r < p.M ? r : 0];
        }
    };
    auto loader_set_tile = [&](int tile) {          // uses ids_next for the gather rows
        long row0; int col0;
        tile_rc(tile, row0, col0);
        rs_a = make_rsrc(gather ? p.a : p.a + row0 * p.lda);
        rs_w = make_rsrc(p.w + (long)col0 * p.ldw);
        const long rows_left = p.M - row0;
        const int cols_left = p.N - col0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int lr = srow + i * RPP;
            const bool ok = lr < rows_left;
            const unsigned off = gather ? (unsigned)ids_next[i] * (unsigned)lda4 : (unsigned)(lr * lda4);
            a_voff[i] = ok ? off + sk * 4 : OOB;
            if constexpr (PE) pe_voff[i] = ok ? (unsigned)((row0 + lr) % p.a_period) * (unsigned)((int)p.lda_pe * 4) + sk * 4 : OOB;
        }
#pragma unroll
        for (int i = 0; i < WPASS; ++i) {
            const int ln = srow + i * RPP;
            w_voff[i] = (ln < cols_left) ? (unsigned)(ln * ldw4) + sk * 4 : OOB;
        }
    };

    vec_t areg[APASS], wreg[WPASS], pereg[PE ? APASS : 1];
    // Loads of one chunk; `part` < 0 issues all of them.  The main loop issues them in three parts in front of the first
    // three k8 steps of the current chunk -- part 0: the activation rows (and positional rows), which come from HBM / the
    // Infinity Cache and need the longest cover; parts 1, 2: the weight rows (L2 hits).  A wave that issues all 6..9 loads
    // back to back sits 1.7..3 k cycles in VMEM issue (measured with s_memtime stamps: 17..24 % of its life) while the
    // MFMA pipe drains; spread out, each stall hides under the 16..20 MFMAs just issued.
    auto issue = [&](int k0, int part) {
        const bool kin = (k0 + sk) < p.K;           // only the tail chunk of a tile has lanes beyond K
        const int soff = k0 * 4;
        if (part < 0 || part == 0) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                areg[i] = buf_load<VEC>(rs_a, kin ? a_voff[i] : OOB, soff);
                if constexpr (PE) pereg[i] = buf_load<VEC>(rs_pe, kin ? pe_voff[i] : OOB, soff);
            }
        }
#pragma unroll
        for (int i = 0; i < WPASS; ++i)
            if (part < 0 || part == 1 + (i & 1)) wreg[i] = buf_load<VEC>(rs_w, kin ? w_voff[i] : OOB, soff);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            vec_t v = areg[i];
            if constexpr (PE) v += pereg[i];
            *reinterpret_cast<vec_t*>(&As[buf][(srow + i * RPP) * LDK + sk]) = v;
        }
#pragma unroll
        for (int i = 0; i < WPASS; ++i) *reinterpret_cast<vec_t*>(&Ws[buf][(srow + i * RPP) * LDK + sk]) = wreg[i];
    };

    // ---- compute state -------------------------------------------------------------------------------
    const int wrow0 = (wave / WN) * TM * 32, wcol0 = (wave % WN) * TN * 32;
    const int fi = lane & 31, fh = lane >> 5;
    const int a_off = (wrow0 + fi) * LDK + fh * 4;
    const int w_off = (wcol0 + fi) * LDK + fh * 4;
    f32x16 acc[TM][TN];

    // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const __amdgpu_buffer_rsrc_t rs_rpe = make_rsrc(p.res_pe ? p.res_pe : p.a);
    auto residual_rsrc = [&](long row0) {
        return make_rsrc(p.res == nullptr ? p.a : ((p.res_ids || p.res_mod > 0) ? p.res : p.res + (row0 / p.res_div) * p.ldr));
    };
    // byte offset of output row `row` inside the residual operand (OOB when there is none), and inside the positional table
    auto residual_row = [&](long row0, long row, unsigned& ro, unsigned& po) {
        ro = OOB; po = 0;
        if (p.res != nullptr && row < p.M) {
            if (p.res_ids) {
                ro = (unsigned)p.res_ids[row] * (unsigned)ldr4;
                if (p.res_pe) po = (unsigned)(row % p.res_period) * (unsigned)((int)p.ldr_pe * 4);
            } else if (p.res_mod > 0) {
                ro = (unsigned)(((row / p.res_div) % p.res_mod) * ldr4);              // a periodic [res_mod, N] table
            } else {
                ro = (unsigned)((row / p.res_div - row0 / p.res_div) * ldr4);
            }
        }
    };
    // one element of the residual: column `col` of the tile that starts at col0 (GENERIC layout)
    auto residual1 = [&](int col0, int col, __amdgpu_buffer_rsrc_t rs_res, unsigned ro, unsigned po) {
        const bool ok = (col0 + col < p.N) && ro != OOB;
        float x = buf_load1(rs_res, ok ? ro + col * 4 : OOB, col0 * 4);
        if (p.res_pe) x += buf_load1(rs_rpe, ok ? po + col * 4 : OOB, col0 * 4);
        return x;
    };

    // Accumulator init of a tile: zero, or the residual operand itself (plain residual add; no operand -> every
    // offset is out of range and the loads return zero, so there is no branch around the accumulators).
    auto acc_init = [&](int tile) {
        long row0; int col0;
        tile_rc(tile, row0, col0);
        if constexpr (!GENERIC) {
            if (tid < BN) Bs[tid] = (p.bias && col0 + tid < p.N) ? p.bias[col0 + tid] : 0.f;     // read after a barrier
        }
        if constexpr (GENERIC) {
            const bool use_res = p.res_in_acc != 0;
            const __amdgpu_buffer_rsrc_t rs_res = residual_rsrc(row0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned ro = OOB, po = 0;
                    if (use_res) residual_row(row0, row0 + wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, ro, po);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] = residual1(col0, wcol0 + j * 32 + fi, rs_res, ro, po);
                }
            }
        } else if constexpr (!LN) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        } else {
            const __amdgpu_buffer_rsrc_t rs_res = residual_rsrc(row0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                unsigned ro, po;
                residual_row(row0, row0 + wrow0 + i * 32 + fi, ro, po);          // this lane's output row
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wcol0 + j * 32 + 8 * g + 4 * fh;            // first of 4 consecutive columns
                        if constexpr (VIO) {
                            const bool ok = (col0 + nl < p.N) && ro != OOB;
                            f32x4 x = buf_load<4>(rs_res, ok ? ro + nl * 4 : OOB, col0 * 4);
                            if (p.res_pe) x += buf_load<4>(rs_rpe, ok ? po + nl * 4 : OOB, col0 * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = x[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = residual1(col0, nl + e, rs_res, ro, po);
                        }
                    }
                }
            }
        }
    };

    // one k8 step: TM + TN fragment reads, then 4 * TM * TN MFMAs
    auto k8_step = [&](const float* Ab, const float* Wb, int s) {
        f32x4 af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + s * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const f32x4*>(Wb + j * 32 * LDK + s * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (GENERIC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], wf[j][u], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][u], af[i][u], acc[i][j], 0, 0, 0);   // D^T = W A^T
                }
    };

    // bias / act / LayerNorm / store of the finished tile.  Rows / columns outside the matrix are neutralised by a
    // select (LayerNorm sums) or by the buffer range check (loads, stores), never by a per-element branch.
    auto epilogue = [&](int tile) {
        long row0; int col0;
        tile_rc(tile, row0, col0);
        const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(p.c + row0 * p.ldc);
        const long rows_left = p.M - row0;
        if constexpr (GENERIC) {
            // row by row: activation (+ late residual) and store
            float bias[TN];
            bool cin[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wcol0 + j * 32 + fi;
                cin[j] = col < p.N;
                bias[j] = (p.bias && cin[j]) ? p.bias[col] : 0.f;
            }
            const unsigned c_lane = (unsigned)((wrow0 + 4 * fh) * ldc4 + (wcol0 + fi) * 4);
            const bool late_res = p.res != nullptr && !p.res_in_acc;
            const __amdgpu_buffer_rsrc_t rs_res = residual_rsrc(row0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool rin = (wrow0 + 4 * fh + rr) < rows_left;
                    unsigned ro = OOB, po = 0;
                    if (late_res) residual_row(row0, row0 + wrow0 + 4 * fh + rr, ro, po);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float v = apply_act(acc[i][j][r] + bias[j], p.act);
                        if (late_res) v += residual1(col0, wcol0 + j * 32 + fi, rs_res, ro, po);
                        buf_store1(v, rs_c, (rin && cin[j]) ? c_lane : OOB, rr * ldc4 + (col0 + j * 32) * 4);
                    }
                }
            }
        } else {
            lds_barrier();                               // Bs (written in acc_init) is visible even when a tile has one chunk
            const int wn = wave % WN;
            float mean[TM], rstd[TM];
            if constexpr (LN) {
                const float inv_n = 1.0f / (float)p.N;
                const float lo = (p.act == LIME_ACT_RELU) ? 0.f : -INFINITY;      // ReLU as a branch-free clamp
                // pass 1: v = acc + bias, row sum over this lane's registers, the other half-wave, the other waves
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nl = wcol0 + j * 32 + 8 * g + 4 * fh;
                            const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[nl]);
                            // columns beyond N hold exact zeros already: W rows beyond N stage as zeros, the residual
                            // load is out of range there and Bs is zero-filled -- no select needed in this pass
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = fmaxf(acc[i][j][4 * g + e] + b[e], lo);
                                acc[i][j][4 * g + e] = v;
                                sum += v;
                            }
                        }
                    }
                    sum += __shfl_xor(sum, 32);
                    if (fh == 0) red_sum[wn * BM + wrow0 + i * 32 + fi] = sum;
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float tot = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) tot += red_sum[w * BM + wrow0 + i * 32 + fi];
                    mean[i] = tot * inv_n;
                    float q = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nl = wcol0 + j * 32 + 8 * g + 4 * fh;
                            const bool gin = nl < p.N;                 // VIO: N % 4 == 0, a group is in or out as a whole
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool in = VIO ? gin : (nl + e < p.N);
                                const float d = in ? acc[i][j][4 * g + e] - mean[i] : 0.f;
                                acc[i][j][4 * g + e] = d;
                                q += d * d;
                            }
                        }
                    }
                    q += __shfl_xor(q, 32);
                    if (fh == 0) red_sq[wn * BM + wrow0 + i * 32 + fi] = q;
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float tot = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) tot += red_sq[w * BM + wrow0 + i * 32 + fi];
                    rstd[i] = 1.0f / sqrtf(tot * inv_n + p.ln_eps);
                    if (p.ln_rstd != nullptr && wn == 0 && fh == 0 && (wrow0 + i * 32 + fi) < rows_left)
                        p.ln_rstd[row0 + wrow0 + i * 32 + fi] = rstd[i];
                }
            }
            // result: 4 consecutive columns per register group -> one 16-byte store
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool rin = (wrow0 + i * 32 + fi) < rows_left;
                const unsigned c_lane = (unsigned)((wrow0 + i * 32 + fi) * ldc4 + (wcol0 + 4 * fh) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wcol0 + j * 32 + 8 * g + 4 * fh;
                        f32x4 v;
                        if constexpr (LN) {
                            const f32x4 ga = *reinterpret_cast<const f32x4*>(&Gs[nl]);
                            const f32x4 be = *reinterpret_cast<const f32x4*>(&Es[nl]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * rstd[i] * ga[e] + be[e];
                        } else {
                            const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[nl]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = acc[i][j][4 * g + e] + b[e];
                                if constexpr (ACT == LIME_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                            }
                        }
                        const int soff = (col0 + j * 32 + 8 * g) * 4;
                        if constexpr (VIO) {
                            buf_store4(v, rs_c, (rin && col0 + nl < p.N) ? c_lane : OOB, soff);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                buf_store1(v[e], rs_c, (rin && col0 + nl + e < p.N) ? c_lane + e * 4 : OOB, soff);
                        }
                    }
                }
            }
        }
    };

    // ---- the chunk stream ----------------------------------------------------------------------------
    // Outer loop over this workgroup's tiles, inner loop over a tile's full chunks, the (possibly partial) last chunk
    // peeled: the accumulators only ever flow through straight-line code and plain loops -- any if-merge that
    // touches them makes hipcc copy the whole 64..80-register array at the merge point.
    if constexpr (LN) {
        if (tid < BN) {
            Gs[tid] = tid < p.N ? p.ln_g[tid] : 0.f;
            Es[tid] = tid < p.N ? p.ln_b[tid] : 0.f;
        }
    }
    prefetch_ids(first);
    loader_set_tile(first);
    prefetch_ids(first + nwg);
    issue(0, -1);
    commit(0);
    lds_barrier();
    int buf = 0;
#ifdef LIME_STAMPS
    unsigned long long tsum[LIME_NSEG] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (int tile = first; tile < ntiles; tile += nwg) {
        const bool more = tile + nwg < ntiles;
        acc_init(tile);
        STAMP(0)                                      // 0: accumulator init (residual loads issued)
        for (int t = 0; t + 1 < nchunk; ++t) {
            STAMP(1)
            const float* Ab = &As[buf][a_off];
            const float* Wb = &Ws[buf][w_off];
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                if (s < 3) issue((t + 1) * BK, s);    // a third of the next chunk's loads in front of each of the first k8 steps
                __builtin_amdgcn_sched_barrier(0);    // pinned: hipcc otherwise sinks every load down to its use
                k8_step(Ab, Wb, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP(2)                                  // 2: prefetch issue + fragment reads + MFMA issue of a full chunk
            commit(buf ^ 1);
            STAMP(3)                                  // 3: vmcnt wait + LDS writes
            lds_barrier();
            STAMP(4)                                  // 4: barrier
            buf ^= 1;
        }
        // last chunk of the tile: the loader moves on to the next tile first
        if (more) {
            loader_set_tile(tile + nwg);
            prefetch_ids(tile + 2 * nwg);
            issue(0, -1);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* Ab = &As[buf][a_off];
            const float* Wb = &Ws[buf][w_off];
            for (int s = 0; s < tail_steps; ++s) k8_step(Ab, Wb, s);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5)                                      // 5: loader switch + tail chunk
        epilogue(tile);
        STAMP(6)                                      // 6: epilogue
        if (more) commit(buf ^ 1);
        lds_barrier();
        STAMP(7)                                      // 7: first commit of the next tile + barrier
        buf ^= 1;
    }
#ifdef LIME_STAMPS
    if (p.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < LIME_NSEG; ++i) p.stamps[((long)blockIdx.x * (NT / 64) + wave) * LIME_NSEG + i] = tsum[i];
    }
#endif
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

template <int TM, int TN, int WM, int WN, int VEC, bool LN, bool PE, int ACT, bool GENERIC, bool VIO>
int launch_one(const GemmP& p0, int wg_per_cu, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    GemmP p = p0;
    p.n_row_blocks = (p.M + BM - 1) / BM;
    p.n_col_blocks = (p.N + BN - 1) / BN;
    const long ntiles = (long)p.n_row_blocks * p.n_col_blocks;
    long nwg = (long)num_cus() * wg_per_cu;
    if (nwg > ntiles) nwg = ntiles;
#ifdef LIME_STAMPS
    p.stamps = g_stamp_buf;
#endif
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WM, WN, VEC, LN, PE, ACT, GENERIC, VIO>), dim3((unsigned)nwg), dim3(WM * WN * 64), 0,
                       stream, p);
    lime_set_last_linear_kernel("gemm_f32_kernel<%d, %d, %d, %d, %d, %s, %s, %d, %s, %s>", TM, TN, WM, WN, VEC, LN ? "true" : "false",
                                PE ? "true" : "false", ACT, GENERIC ? "true" : "false", VIO ? "true" : "false");
    return lime_check_launch("lime_linear_f32");
}

// small tiles (four waves, 64 x 64): everything at run time
int launch_small(const GemmP& p, int vec, hipStream_t s) {
    const bool pe = p.a_pe != nullptr;
    if (vec == 4) return pe ? launch_one<1, 1, 2, 2, 4, false, true, 0, true, false>(p, 4, s) : launch_one<1, 1, 2, 2, 4, false, false, 0, true, false>(p, 4, s);
    if (vec == 2) return pe ? launch_one<1, 1, 2, 2, 2, false, true, 0, true, false>(p, 4, s) : launch_one<1, 1, 2, 2, 2, false, false, 0, true, false>(p, 4, s);
    return pe ? launch_one<1, 1, 2, 2, 1, false, true, 0, true, false>(p, 4, s) : launch_one<1, 1, 2, 2, 1, false, false, 0, true, false>(p, 4, s);
}

inline bool aligned(const void* ptr, long ld, int vec) {
    return ptr == nullptr || (((uintptr_t)ptr % (vec * sizeof(float))) == 0 && (ld % vec) == 0);
}

}  // namespace

extern "C" int lime_relu_bwd_f32(float* dh, int64_t lddh, const float* h, int64_t ldh, int64_t rows, int32_t cols, float scale,
                                 void* stream);          // backward_f32.hip
extern "C" int lime_dropout_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, float p, uint64_t seed,
                                uint32_t site, void* stream);           // dropout_f32.hip

extern "C" int lime_linear_f32(const lime_linear_args* a, void* stream) {
    LIME_REQUIRE(a != nullptr, LIME_ERR_BAD_ARG, "lime_linear_f32: args is NULL");
    LIME_REQUIRE(a->a && a->w && a->c, LIME_ERR_BAD_ARG, "lime_linear_f32: a, w and c must be non-NULL");
    LIME_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, LIME_ERR_BAD_ARG, "lime_linear_f32: bad dims M=%d N=%d K=%d", a->M, a->N,
                 a->K);
    LIME_REQUIRE(a->ldw >= a->K && a->ldc >= a->N && a->lda >= a->K, LIME_ERR_BAD_ARG,
                 "lime_linear_f32: leading dimension smaller than the row (lda=%ld ldw=%ld ldc=%ld)", (long)a->lda,
                 (long)a->ldw, (long)a->ldc);
    LIME_REQUIRE(!a->a_pe || (a->a_ids && a->a_period > 0 && a->lda_pe >= a->K), LIME_ERR_BAD_ARG,
                 "lime_linear_f32: a_pe needs a_ids, a_period > 0 and lda_pe >= K");
    LIME_REQUIRE(!a->res || a->res_ids || a->res_div >= 1, LIME_ERR_BAD_ARG, "lime_linear_f32: res_div must be >= 1");
    LIME_REQUIRE(!a->res || a->ldr >= a->N, LIME_ERR_BAD_ARG, "lime_linear_f32: ldr smaller than N");
    LIME_REQUIRE(!a->res_pe || (a->res_ids && a->res_period > 0 && a->ldr_pe >= a->N), LIME_ERR_BAD_ARG,
                 "lime_linear_f32: res_pe needs res_ids, res_period > 0 and ldr_pe >= N");
    LIME_REQUIRE(!a->ln_gamma || a->ln_beta, LIME_ERR_BAD_ARG, "lime_linear_f32: ln_gamma without ln_beta");
    LIME_REQUIRE(a->act >= LIME_ACT_NONE && a->act <= LIME_ACT_RELU_GRAD, LIME_ERR_BAD_ARG, "lime_linear_f32: bad act %d", a->act);
    LIME_REQUIRE(a->res_mod >= 0 && (a->pool32 == 0 || a->pool32 == 1), LIME_ERR_BAD_ARG, "lime_linear_f32: res_mod < 0 or pool32 not 0 / 1");
    LIME_REQUIRE(!a->ln_rstd || (a->ln_gamma && !a->pool32), LIME_ERR_BAD_ARG, "lime_linear_f32: ln_rstd needs the LayerNorm epilogue without pool32");
    if (a->M == 0) return LIME_OK;

    if (a->act == LIME_ACT_RELU_GRAD) {               // the ReLU gradient as an epilogue: fused in the split-product kernel, else two launches
        LIME_REQUIRE(a->res && !a->res_ids && a->res_mod == 0 && a->res_div <= 1 && !a->ln_gamma && !a->pool32 && !a->m_dev && !a->c_ids,
                     LIME_ERR_BAD_ARG, "lime_linear_f32: LIME_ACT_RELU_GRAD takes res = the forward activation (dense rows) and no other epilogue");
        const int sp = lime_linear_sp(a, (hipStream_t)stream);
        if (sp != LIME_PP_NOT_APPLICABLE) return sp;
        lime_linear_args plain = *a;
        plain.act = LIME_ACT_NONE;
        plain.res = nullptr;
        const int st = lime_linear_f32(&plain, stream);
        if (st != LIME_OK) return st;
        return lime_relu_bwd_f32(a->c, a->ldc, a->res, a->ldr, a->M, a->N, a->act_scale, stream);
    }

    if (a->dropout_p != 0.f) {                        // dropout behind the activation: fused in the split-product ReLU kernel, else a pass of its own
        LIME_REQUIRE(a->dropout_p > 0.f && a->dropout_p < 1.f, LIME_ERR_BAD_ARG, "lime_linear_f32: dropout_p outside [0, 1)");
        LIME_REQUIRE((a->act == LIME_ACT_NONE || a->act == LIME_ACT_RELU) && !a->res && !a->ln_gamma && !a->pool32 && !a->c_ids && !a->m_dev,
                     LIME_ERR_BAD_ARG, "lime_linear_f32: dropout_p goes with act none / ReLU and no other epilogue");
        const int sp = lime_linear_sp(a, (hipStream_t)stream);
        if (sp != LIME_PP_NOT_APPLICABLE) return sp;
        lime_linear_args plain = *a;
        plain.dropout_p = 0.f;
        const int st = lime_linear_f32(&plain, stream);
        if (st != LIME_OK) return st;
        return lime_dropout_f32(a->c, a->ldc, a->c, a->ldc, a->M, a->N, a->dropout_p, a->dropout_seed, a->dropout_site, stream);
    }

    // 4096 <= M with few 128-row tiles (M = 6400, N = 400: 100 tiles on 512 workgroup slots took 60 us on the big-M kernel, 32 us in
    // 64 x 64 tiles): the mid-M kernel first, for the problems it takes (no LayerNorm / pooling / scatter)
    if (a->M >= 4096 && !a->ln_gamma && !a->pool32 && !a->c_ids && !a->a_pe && !(lime_split_mode() & 4) &&       // (bit 2: tests pin kernels)
        (((long)a->M + 127) / 128) * (((long)a->N + 255) / 256) < 160) {
        const int st = lime_linear_mid(a, (hipStream_t)stream);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }

    // big M, 16-byte friendly operands: two four-wave workgroups per CU with LDS-DMA staging (gemm_pp_f32.hip)
    static const bool pp_off = getenv("LIME_GEMM_NO_PP") != nullptr;          // A/B switch for tools/, not a product option
    if (!pp_off || a->pool32) {
        // first choice: fp32-level split products on the bf16 matrix cores (gemm_sp_f32.hip; lime_set_split_gemm(0) turns it off)
        const int sp = lime_linear_sp(a, (hipStream_t)stream);
        if (sp != LIME_PP_NOT_APPLICABLE) return sp;
        const int st = lime_linear_pp(a, (hipStream_t)stream);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    LIME_REQUIRE(!a->c_ids, LIME_ERR_UNSUPPORTED,
                 "lime_linear_f32: c_ids needs the big-M kernel (M >= 4096, 16-byte operands, periodic residual, no LayerNorm, act none)");
    if (a->m_dev) {                                   // a device-side row count outside the big-M kernel: the mid-M kernel honours it
        const int st = lime_linear_mid(a, (hipStream_t)stream);
        LIME_REQUIRE(st != LIME_PP_NOT_APPLICABLE, LIME_ERR_UNSUPPORTED,
                     "lime_linear_f32: m_dev needs 16-byte friendly operands (K, N multiples of 4, aligned rows) and no LayerNorm / pooling / a_pe");
        return st;
    }
    LIME_REQUIRE(!a->pool32, LIME_ERR_UNSUPPORTED,
                 "lime_linear_f32: pool32 needs the big-M kernel (M >= 4096 and a multiple of 32, LayerNorm + dense residual, 16-byte operands)");

    GemmP p;
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids; p.a_pe = a->a_pe; p.lda_pe = a->lda_pe; p.a_period = a->a_period;
    p.w = a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = a->res; p.ldr = a->ldr; p.res_div = a->res_div > 0 ? a->res_div : 1; p.res_ids = a->res_ids;
    p.res_mod = (a->res && !a->res_ids && a->res_mod > 0) ? a->res_mod : 0;
    p.res_pe = a->res_pe; p.ldr_pe = a->ldr_pe; p.res_period = a->res_period;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_rstd = a->ln_rstd;
    p.c = a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act;
    p.n_row_blocks = p.n_col_blocks = 0;
    // acc + bias + res (the documented order with no activation) == (res + acc) + bias up to fp32 rounding
    p.res_in_acc = (a->res != nullptr && a->act == LIME_ACT_NONE) ? 1 : 0;

    int vec = 4;
    while (vec > 1 && !((a->K % vec) == 0 && aligned(a->a, a->lda, vec) && aligned(a->w, a->ldw, vec) &&
                        aligned(a->a_pe, a->lda_pe, vec)))
        vec >>= 1;
    hipStream_t s = (hipStream_t)stream;
    // Tile selection.
    //   LayerNorm epilogue: one eight-wave 128 x 256 / 128 x 320 tile spans the row (N <= 320); the residual (if any) is
    //   preloaded into the accumulators, so it must be a plain add.  Fast instantiation: 16-byte operand staging and
    //   16-byte residual / result accesses; anything misaligned takes the scalar-IO instantiation.
    //   Big M, no residual, activation none / ReLU, everything 16-byte friendly: the same eight-wave tiles, the width
    //   (256 / 320) that pads N least.  Everything else: 64 x 64 tiles with run-time epilogue.
    const bool has_res = a->res != nullptr;
    const bool vio = vec == 4 && (a->N % 4 == 0) && aligned(a->c, a->ldc, 4) && aligned(a->res, a->ldr, 4) &&
                     aligned(a->res_pe, a->ldr_pe, 4);
    if (a->ln_gamma) {
        LIME_REQUIRE(a->N <= 320, LIME_ERR_UNSUPPORTED, "lime_linear_f32: LayerNorm epilogue needs N <= 320 (N=%d)", a->N);
        LIME_REQUIRE(a->act == LIME_ACT_NONE || (a->act == LIME_ACT_RELU && !has_res), LIME_ERR_UNSUPPORTED,
                     "lime_linear_f32: LayerNorm epilogue supports act none (+ residual) or ReLU (no residual)");
        LIME_REQUIRE(a->a_pe == nullptr, LIME_ERR_UNSUPPORTED, "lime_linear_f32: LayerNorm epilogue with a positional A operand");
        if (a->N <= 256) return vio ? launch_one<1, 4, 4, 2, 4, true, false, 0, false, true>(p, 1, s)
                                    : launch_one<1, 4, 4, 2, 1, true, false, 0, false, false>(p, 1, s);
        return vio ? launch_one<1, 5, 4, 2, 4, true, false, 0, false, true>(p, 1, s)
                   : launch_one<1, 5, 4, 2, 1, true, false, 0, false, false>(p, 1, s);
    }
    const bool simple = !has_res && (a->act == LIME_ACT_NONE || a->act == LIME_ACT_RELU);
    if (a->M >= 4096 && simple && vio) {
        const int pad5 = (a->N + 319) / 320 * 320 - a->N, pad4 = (a->N + 255) / 256 * 256 - a->N;
        const bool relu = a->act == LIME_ACT_RELU, pe = a->a_pe != nullptr;
        if (pad5 < pad4) {
            if (pe) return relu ? launch_one<1, 5, 4, 2, 4, false, true, 1, false, true>(p, 1, s) : launch_one<1, 5, 4, 2, 4, false, true, 0, false, true>(p, 1, s);
            return relu ? launch_one<1, 5, 4, 2, 4, false, false, 1, false, true>(p, 1, s) : launch_one<1, 5, 4, 2, 4, false, false, 0, false, true>(p, 1, s);
        }
        if (pe) return relu ? launch_one<1, 4, 4, 2, 4, false, true, 1, false, true>(p, 1, s) : launch_one<1, 4, 4, 2, 4, false, true, 0, false, true>(p, 1, s);
        return relu ? launch_one<1, 4, 4, 2, 4, false, false, 1, false, true>(p, 1, s) : launch_one<1, 4, 4, 2, 4, false, false, 0, false, true>(p, 1, s);
    }
    {   // small / mid M (and anything the big-tile kernels above do not take): the deep-prefetch LDS-DMA kernel (gemm_mid_f32.hip)
        const int st = lime_linear_mid(a, s);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    return launch_small(p, vec, s);
}
