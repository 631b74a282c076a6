// lime_linear_f32, small / mid-M instantiation: the GEMMs around the token encoders (M = 32 ... ~10 k rows: intent layers,
// intent attention, LIME.project, gate_proj, SAGEConv, K / Q, the positional table through in_proj; K = 300 ... 1800).
//
// These launches are LATENCY bound, not throughput bound: a 64 x 64 tile of a K = 400 problem is 3 MFLOP (5 us at one CU's
// MFMA rate) and there are fewer tiles than workgroup slots, so the time of a launch is the time of ONE tile's k loop.  The
// register-staged kernel of gemm_f32.hip keeps one 32-deep chunk in flight (0.4 us of MFMAs against > 1 us of L2 / HBM load
// latency): 17 us for [1600, 400] x [400, 400], 54 us for K = 1800.  Here the operands go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write) into a ring of SIX 16-deep stages, five chunks ahead of
// the MFMAs (1 us of cover), one barrier per chunk; a workgroup computes one 64 x 64 tile and exits (48 KB LDS, three per
// CU).  Same operand image, swizzle and transposed product as gemm_pp_f32.hip (D^T = W A^T on v_mfma_f32_16x16x4_f32: an
// output row's 4 consecutive columns sit in one lane -> 16-byte bias / residual loads and result stores).
//
// Epilogue: out = act(acc + bias) + residual, act in {none, ReLU, tanh, sigmoid}; residual row of output row r:
// res_ids[r] (gathered), (r / res_div) % res_mod (periodic table) or r / res_div (dense / broadcast rows).  A rows may be
// gathered (a_ids).  No LayerNorm / pooling here (those shapes are the big-M kernel's).
#include "common.h"
#include "gemm_pp.h"

namespace {

constexpr int MB = 64, NB = 64, KB = 16, NS = 6;       // MB x NB: the largest tile (LDS is sized for it); KB-deep chunks, NS stages
constexpr unsigned OOB = 0x80000000u;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct MidP {
    const float* a; long lda; const int* a_ids;
    const float* w; long ldw; const float* bias;
    const float* res; long ldr; int res_div, res_mod; const int* res_ids;
    float* c; long ldc; int M, N, K, act, n_col_blocks;
    int shape;           // tile shape of this problem: 0 = 64 x 64, 1 = 32 x 64, 2 = 32 x 32 (mid_params)
    const int* m_dev;    // optional device-side row count (min(*m_dev, M) rows; workgroups of tiles beyond it exit)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds_base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_base, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }        // see gemm_pp_f32.hip

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == LIME_ACT_RELU) return fmaxf(v, 0.f);
    if (act == LIME_ACT_TANH) return tanhf(v);
    if (act == LIME_ACT_SIGMOID) return lime_sigmoid(v);
    return v;
}

constexpr int STAGE_MAX = (MB + NB) * KB;                            // floats

// One TM x TN tile of problem p (tile index inside the problem); lds: the workgroup's NS * STAGE_MAX floats.  The tile shape is chosen
// per problem (MidP.shape): a launch whose 64 x 64 tiles leave workgroup slots empty is bound by ONE tile's k loop -- 16 MFMAs of 32
// cycles per wave and chunk, one wave per SIMD -- so the same problem in 32 x 64 or 32 x 32 tiles (8 / 4 MFMAs per wave and chunk,
// two or four times the workgroups) finishes sooner.  Every output element sees the same k order in all three: the results are
// bit-identical.  Waves: TM / 16 row groups x 4 / (TM / 16) column groups of TN / (column groups) columns.
template <int TM, int TN>
__device__ __forceinline__ void mid_tile(const MidP& p, const int tile, float* const lds) {
    constexpr int WR = TM / 16, WC = 4 / WR, NT = TN / (16 * WC);     // row groups, column groups, 16-column MFMA tiles per wave
    constexpr int NA = TM / 16, NW = TN / 16;                          // 16-row DMA groups of A and of W per chunk
    constexpr int DPW = (NA + NW + 3) / 4;                             // DMA instructions per wave and chunk (2, 2, 1)
    constexpr int A_ST = TM * KB, STAGE = (TM + TN) * KB;
    static_assert(WR * WC == 4 && NT >= 1 && STAGE <= STAGE_MAX, "tile shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % WR, wc = wave / WR;
    const int fi = lane & 15, kg = lane >> 4;
    const int rb = tile / p.n_col_blocks;
    const int row0 = rb * TM, col0 = (tile - rb * p.n_col_blocks) * TN;
    int M = p.M;
    if (p.m_dev) {
        const int m = __builtin_amdgcn_readfirstlane(*p.m_dev);
        M = m < M ? m : M;
    }
    if (row0 >= M) return;                           // (uniform per workgroup: nothing was issued yet)

    // ---- loader: the NA + NW 16-row groups of a chunk (1 KB DMA each) are dealt to the waves, DPW per wave (a wave without a group
    // in a round issues an out-of-bounds piece: the counted waits below assume DPW instructions per wave and chunk) ---------------
    const int srow = lane >> 2;
    const int lseg = (lane & 3) ^ swz4((lane >> 4) & 3);
    const int lda4 = (int)p.lda * 4, ldw4 = (int)p.ldw * 4;
    const bool gather = p.a_ids != nullptr;
    const __amdgpu_buffer_rsrc_t rs_a = mk_rsrc(gather ? (const char*)p.a : (const char*)p.a + (long)row0 * p.lda * 4);
    const __amdgpu_buffer_rsrc_t rs_w = mk_rsrc((const char*)p.w + (long)col0 * p.ldw * 4);
    unsigned g_voff[DPW];                           // group wave + 4 d: A group (< NA), W group (- NA) or none
    int g_lds[DPW], g_kind[DPW];                    // kind 0: A, 1: W, 2: none (its piece is zeros into a dump area behind the stages)
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
        const int g = wave + 4 * d;
        unsigned vo = OOB;
        if (g < NA) {
            const int rl = 16 * g + srow;
            unsigned rowsel = (unsigned)rl;
            if (gather && row0 + rl < M) rowsel = (unsigned)p.a_ids[row0 + rl];
            if (row0 + rl < M) vo = rowsel * (unsigned)lda4 + (unsigned)lseg * 16u;
            g_lds[d] = g * 256; g_kind[d] = 0;
        } else if (g < NA + NW) {
            const int rl = 16 * (g - NA) + srow;
            if (col0 + rl < p.N) vo = (unsigned)rl * (unsigned)ldw4 + (unsigned)lseg * 16u;
            g_lds[d] = A_ST + (g - NA) * 256; g_kind[d] = 1;
        } else {
            static_assert(NS * STAGE + 4 * 256 <= NS * STAGE_MAX || (NA + NW) % 4 == 0, "room for the dump area");
            g_lds[d] = NS * STAGE + wave * 256; g_kind[d] = 2;
        }
        g_voff[d] = vo;
    }
    const int nchunk = (p.K + KB - 1) / KB;
    auto issue = [&](int c) {                       // chunk c -> stage c % NS; beyond K (or beyond the last chunk): zeros
        const bool kin = c * KB + lseg * 4 < p.K;
        float* const sb = lds + (c % NS) * STAGE;
#pragma unroll
        for (int d = 0; d < DPW; ++d)               // exactly DPW instructions per wave and chunk: the counted waits below rely on it
            dma16(g_kind[d] == 0 ? rs_a : rs_w, g_kind[d] == 2 ? lds + g_lds[d] : sb + g_lds[d], kin ? g_voff[d] : OOB, c * 64);
    };

    // ---- compute: wave (wr, wc) owns output rows 16 wr .. + 15 x columns 16 NT wc .. of the tile (NT MFMA column tiles) ---------
    const int pseg = (kg ^ swz4((fi >> 2) & 3)) * 4;
    const int a_off = (16 * wr + fi) * KB + pseg, w_off = A_ST + (16 * NT * wc + fi) * KB + pseg;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int c = 0; c < NS - 1; ++c) issue(c);
    // Software pipeline by one chunk: iteration c waits for chunk c + 1, reads ITS fragments into registers and issues the
    // MFMAs of chunk c behind those reads, so the LDS round trip of a chunk's ds_read_b128 hides under the previous chunk's
    // MFMAs (the exposed read + barrier + issue were ~0.2 us of every 0.4 us chunk).
    auto read_frags = [&](int c, f32x4& af, f32x4 (&wf)[NT]) {
        const float* sb = lds + (c % NS) * STAGE;
        af = *reinterpret_cast<const f32x4*>(sb + a_off);
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const f32x4*>(sb + w_off + t * 16 * KB);
    };
    f32x4 af, wf[NT], afn, wfn[NT];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (NS - 2)) : "memory");    // chunk 0: this wave's pieces ...
    lds_barrier();                                                            // ... and everyone's
    read_frags(0, af, wf);
    for (int c = 0; c < nchunk; ++c) {
        // chunk c + 1 has landed when at most the DPW (NS - 3) younger DMA instructions of this wave are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (NS - 3)) : "memory");
        lds_barrier();                              // everyone's pieces of chunk c + 1; and every wave has READ stage c % NS
        __builtin_amdgcn_sched_barrier(0);
        issue(c + NS - 1);                          // into stage (c - 1) % NS, whose fragments were read an iteration ago
        __builtin_amdgcn_sched_barrier(0);
        read_frags(c + 1, afn, wfn);                // (past the last chunk: a zero-filled stage, never used)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][q], af[q], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        af = afn;
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = wfn[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued past the last chunk must land before the LDS is released

    // ---- epilogue: lane (fi, kg) holds row 16 wr + fi, columns 16 (NT wc + t) + 4 kg .. + 3 ---------------------------------------
    const int row = row0 + 16 * wr + fi;
    if (row >= M) return;
    long rrow = -1;
    if (p.res) {
        if (p.res_ids) rrow = p.res_ids[row];
        else if (p.res_mod > 0) rrow = (row / p.res_div) % p.res_mod;
        else rrow = row / p.res_div;
    }
    float* const crow = p.c + (long)row * p.ldc;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = col0 + 16 * (NT * wc + t) + 4 * kg;
        if (col >= p.N) continue;                   // N % 4 == 0: a group of four is in or out as a whole
        f32x4 v = acc[t];
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += p.bias[col + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_fn(v[j], p.act);
        if (rrow >= 0) v += *reinterpret_cast<const f32x4*>(p.res + rrow * p.ldr + col);
        *reinterpret_cast<f32x4*>(crow + col) = v;
    }
}

// shape 0: 64 x 64, 1: 32 x 64, 2: 32 x 32 (uniform per workgroup)
__device__ __forceinline__ void mid_tile_any(const MidP& p, const int tile, float* const lds) {
    if (p.shape == 2) mid_tile<32, 32>(p, tile, lds);
    else if (p.shape == 1) mid_tile<32, 64>(p, tile, lds);
    else mid_tile<64, 64>(p, tile, lds);
}

__global__ __launch_bounds__(256, 3) void gemm_mid_kernel(const MidP p) {
    __shared__ __attribute__((aligned(16))) float lds[NS * STAGE_MAX];
    mid_tile_any(p, (int)blockIdx.x, lds);
}

// Several INDEPENDENT problems in one launch: the launches around the encoders are latency bound (one tile's k loop, >= 10 us each
// however small), so two GEMMs that do not depend on each other cost one launch's time side by side instead of two in a row.
constexpr int MAX_GROUP = 8;
struct MidGroup {
    MidP p[MAX_GROUP];
    int first[MAX_GROUP + 1];       // first[k]: the first workgroup of problem k; first[n] = the grid
    int n;
};

__global__ __launch_bounds__(256, 3) void gemm_mid_group_kernel(const MidGroup g) {
    __shared__ __attribute__((aligned(16))) float lds[NS * STAGE_MAX];
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.first[k + 1]) ++k;        // uniform: scalar loads and compares
    mid_tile_any(g.p[k], (int)blockIdx.x - g.first[k], lds);
}

int mid_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n = cus;
        else n = 256;
    }
    return n;
}

inline bool al16(const void* ptr, long ld) { return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld % 4) == 0); }

}  // namespace

// fills p and the tile count; false: the problem is outside this kernel
static bool mid_params(const lime_linear_args* a, MidP& p, long& ntiles) {
    if (a->ln_gamma || a->pool32 || a->a_pe || a->ln_rstd || a->c_ids || a->res_pe) return false;
    if (a->K % 4 || a->N % 4 || a->K < 16) return false;
    if (!al16(a->a, a->lda) || !al16(a->w, a->ldw) || !al16(a->c, a->ldc) || !al16(a->res, a->ldr)) return false;
    const long lim = 0x7FFFFFF0L;
    if (64L * a->lda * 4 >= lim || 64L * a->ldw * 4 >= lim) return false;
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids;
    p.w = a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = a->res; p.ldr = a->ldr; p.res_div = a->res_div > 0 ? a->res_div : 1; p.res_ids = a->res_ids;
    p.res_mod = (a->res && !a->res_ids && a->res_mod > 0) ? a->res_mod : 0;
    p.c = a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act; p.m_dev = a->m_dev;
    // the smallest tiles whose workgroups still run in ONE round (three workgroups per CU): the launch is then one tile's k loop long
    static const long slots = 3L * mid_cus();
    const long t64 = (long)((a->M + 63) / 64) * ((a->N + 63) / 64), t3264 = (long)((a->M + 31) / 32) * ((a->N + 63) / 64),
               t32 = (long)((a->M + 31) / 32) * ((a->N + 31) / 32);
    static const char* const force = getenv("LIME_MID_SHAPE");               // A/B switch for tools/, not a product option
    p.shape = force ? (atoi(force) == 2 ? 2 : (atoi(force) == 1 ? 1 : 0)) : (t32 <= slots ? 2 : (t3264 <= slots ? 1 : 0));
    const int tn = p.shape == 2 ? 32 : 64;
    p.n_col_blocks = (a->N + tn - 1) / tn;
    ntiles = p.shape == 2 ? t32 : (p.shape == 1 ? t3264 : t64);
    return ntiles <= 0x3FFFFFFFL;
}

// LIME_OK / error: launched; LIME_PP_NOT_APPLICABLE: the caller takes the general kernel.
int lime_linear_mid(const lime_linear_args* a, hipStream_t s) {
    static const bool off = getenv("LIME_GEMM_NO_MID") != nullptr;           // A/B switch for tools/, not a product option
    if (off) return LIME_PP_NOT_APPLICABLE;
    MidP p;
    long ntiles = 0;
    if (!mid_params(a, p, ntiles)) return LIME_PP_NOT_APPLICABLE;
    hipLaunchKernelGGL(gemm_mid_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, p);
    lime_set_last_linear_kernel("gemm_mid_kernel");
    return lime_check_launch("lime_linear_f32");
}

extern "C" int lime_linear_group_f32(const lime_linear_args* args, int32_t n, void* stream) {
    LIME_REQUIRE(args != nullptr && n >= 1 && n <= MAX_GROUP, LIME_ERR_BAD_ARG, "lime_linear_group_f32: args is NULL or n outside 1 .. %d", MAX_GROUP);
    MidGroup g;
    g.n = 0;
    long total = 0;
    for (int k = 0; k < n; ++k) {
        const lime_linear_args* a = &args[k];
        LIME_REQUIRE(a->a && a->w && a->c, LIME_ERR_BAD_ARG, "lime_linear_group_f32: problem %d: a, w and c must be non-NULL", k);
        LIME_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0 && a->ldw >= a->K && a->ldc >= a->N && a->lda >= a->K, LIME_ERR_BAD_ARG,
                     "lime_linear_group_f32: problem %d: bad dims / leading dimensions", k);
        LIME_REQUIRE(a->act >= LIME_ACT_NONE && a->act <= LIME_ACT_SIGMOID && a->res_mod >= 0 && (!a->res || a->res_ids || a->res_div >= 1) &&
                     (!a->res || a->ldr >= a->N), LIME_ERR_BAD_ARG, "lime_linear_group_f32: problem %d: bad act / residual arguments", k);
        if (a->M == 0) continue;
        long ntiles = 0;
        LIME_REQUIRE(mid_params(a, g.p[g.n], ntiles), LIME_ERR_UNSUPPORTED,
                     "lime_linear_group_f32: problem %d is outside the mid-M kernel (16-byte friendly operands, K >= 16, no LayerNorm / pooling / "
                     "a_pe / c_ids / res_pe): launch it with lime_linear_f32", k);
        g.first[g.n] = (int)total;
        total += ntiles;
        LIME_REQUIRE(total <= 0x3FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_linear_group_f32: too many tiles");
        ++g.n;
    }
    if (g.n == 0) return LIME_OK;
    g.first[g.n] = (int)total;
    hipLaunchKernelGGL(gemm_mid_group_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    lime_set_last_linear_kernel("gemm_mid_group_kernel");
    return lime_check_launch("lime_linear_group_f32");
}
