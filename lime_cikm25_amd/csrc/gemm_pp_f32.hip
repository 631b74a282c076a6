// lime_linear_f32, big-M instantiations: two independent four-wave workgroups per CU ("ping-pong"), operands staged
// global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), exact-fp32 MFMA.
//
// Why a second GEMM kernel (the eight-wave persistent kernel of gemm_f32.hip stays for every other shape):
// s_memtime stamps of that kernel on the encoder shapes (K = 300 ... 512, i.e. 10 ... 16 chunks per tile) showed the MFMA
// pipe issued only ~60 % of a wave's life: 8-18 % went to vmcnt + ds_write of the register-staged prefetch, 5-15 % to the
// workgroup barrier, 13-30 % to the tile boundary (residual loads into the accumulators, LayerNorm epilogue, stores) --
// and with ONE workgroup per CU every wave of the CU sits in those phases at the same time, so nothing fills the pipe.
// Here:
//   * a workgroup is 4 waves (one per SIMD) and owns a 128 x 320 (or 128 x 256) output tile; TWO workgroups are resident
//     per CU (2 x 61 KB LDS, <= 256 VGPRs), each SIMD holds one wave of either.  They run unsynchronised, so one
//     workgroup's barrier / load wait / epilogue is the other's MFMA time.  A lone wave can keep its SIMD's MFMA pipe full
//     (10 independent accumulator tiles), so a CU with one workgroup left is not slower per tile.
//   * operands go global -> LDS without passing registers: no staging VGPRs (the budget goes to a 32 x 320 accumulator
//     slab per wave), no ds_write, no vmcnt wait in front of a ds_write.  K chunks are 16 deep, two LDS stages; the DMA of
//     chunk c + 1 is in flight under the MFMAs of chunk c (a wave issues its two A instructions in front of them and its
//     weight instructions in the middle, staggered by wave, so that the workgroup's 28 instructions do not queue up in
//     the CU's address path all at once); one barrier per chunk.
//   * an LDS-DMA instruction writes 1 KB linearly (lane l -> base + 16 l), so rows cannot be padded; the image is
//     [row][16 floats] with the 16-byte segment index XOR-swizzled by swz4((row >> 2) & 3) -- applied to the per-lane
//     SOURCE address when staging and to the ds_read_b128 address when reading fragments (see swz4: the permutation is
//     chosen for the hardware's non-contiguous 16-lane read groups; measured SQ_LDS_BANK_CONFLICT = 0).
//   * the MFMA is v_mfma_f32_16x16x4_f32, not 32x32x2: same nominal rate, same LDS traffic per flop here, but half the
//     accumulator-register traffic per flop.  (A registers-only MFMA loop on random operands is POWER limited with
//     32x32x2 -- tools/probes/mfma_probe sustains 122 TFLOP/s against 154 with 16x16x4, both 155 on zeros; this kernel is
//     not: tools/power_probe.py, same time on zeros and random data at 2.39 GHz.)
//   * a wave owns 32 output rows x all tile columns of the transposed product D^T = W A^T: an output row lives in the
//     four lanes (i, i + 16, i + 32, i + 48), each holding 4 of every 16 columns as consecutive registers -- residual
//     loads / result stores are 16-byte accesses, and LayerNorm needs no cross-wave exchange at all (in-lane sums +
//     two shuffles).
//   * k beyond K, rows beyond M and weight rows beyond N are out-of-range buffer offsets: the DMA writes zeros.
#include "common.h"
#include "gemm_pp.h"

#ifdef LIME_STAMPS
// Diagnostic build only (tools/gemm_stamps.py): per-wave s_memtime sums of the main-loop segments; never in liblime_hip.so.
static unsigned long long* g_pp_stamp_buf = nullptr;
extern "C" void lime_debug_set_pp_stamp_buffer(unsigned long long* p) { g_pp_stamp_buf = p; }
#define PSTAMP(i)                                                           \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 \
        tsum[i] += t_ - tlast;                                              \
        tlast = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
    }
#else
#define PSTAMP(i)
#endif

namespace {

constexpr int BM = 128;   // output rows per tile (4 waves x 32)
constexpr int BK = 16;    // k depth of one LDS stage
constexpr unsigned OOB = 0x80000000u;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int buf_load_i32(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}
// 16 bytes per lane global -> LDS (lane l lands at lds_base + 16 l), out-of-range offsets write zeros.
// (The builtin only exists in the device pass; inside a kernel TEMPLATE it makes the host pass drop the launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds_base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_base, 16, voff, soff, 0, 0);
#endif
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 <-> fp32 (round to nearest even; inputs are finite on this path)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {       // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 v) {
    f32x4 r;
    r[0] = __builtin_bit_cast(float, v[0] << 16);
    r[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    r[2] = __builtin_bit_cast(float, v[1] << 16);
    r[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return r;
}
__device__ __forceinline__ f32x4 buf_load4_bf16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {     // 4 bf16 -> 4 floats
    return unpack_bf16x4(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)));
}
__device__ __forceinline__ void buf_store4_bf16(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    u32x2 o;
    o[0] = pack_bf16(v[0], v[1]);
    o[1] = pack_bf16(v[2], v[3]);
    __builtin_amdgcn_raw_buffer_store_b64(o, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// Swizzle term of image row r: segment XOR swz4((r >> 2) & 3).  ds_read_b128 is serviced in four NON-contiguous 16-lane
// groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32 for the other two; MI355X_MICROARCH.md, LDS): with the MFMA
// 16x16 lane layout (row = lane & 15, k slot = lane >> 4) a group mixes rows 0-3 / 12-15 of one k slot with rows 4-11 of
// another, and the plain XOR with (r >> 2) & 3 put two rows of every group on each 16-byte slot (SQ_LDS_BANK_CONFLICT =
// 49 % of the LDS cycles).  The permutation {0, 2, 3, 1} of (r >> 2) & 3 makes all four groups conflict free.
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

// Sum over the 16 lanes of a DPP row (the lanes that share a k slot, i.e. the 16 tokens of an MFMA tile): butterfly with
// quad_perm (xor 1, xor 2) and row rotations by 4 and 8; every lane ends up with the total.
__device__ __forceinline__ float row16_sum(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, true));       // row_ror:4
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true));       // row_ror:8
    return v;
}

// NTL: 32-column MFMA tiles per wave (tile width 32 * NTL).  LN: LayerNorm epilogue (one column block spans N).
// RES: 0 none, 1 dense fp32 residual rows (r, or r % res_mod), 2 gathered rows + fp32 positional table,
//      3 (BF only) dense bf16 residual rows.
// BF:  bf16 operands (A rows, gathered table rows, W) and bf16 output, v_mfma_f32_16x16x32_bf16, fp32 accumulation and
//      epilogue (lime_linear_bf16).  The LDS image is byte-identical: a row's chunk is 64 bytes = 16 floats or 32 bf16.
// POOL: (LayerNorm epilogue) instead of the [M, N] result, row r of C is the mean of result rows 32 r .. 32 r + 31 -- a
//      wave's 32 output rows -- so the mean pooling over the tokens of a news (newsEncoders.py:317,321) never sees the
//      activations in HBM: S = 32 sequences are finished here, longer ones by a mean over their S / 32 block rows.
// CID: the A rows are a COMPACTED list (the live tokens of the batch): result row r is stored at row c_ids[r] of C, and a
//      periodic residual (RES == 1 with res_mod) is indexed by c_ids[r] % res_mod -- in_proj over the non-padding tokens only.
// p.m_dev (any instantiation): the row count is read from device memory (min(*m_dev, M)), so a launch captured into a HIP
//      graph follows the batch's live-row count without a host round trip.
// TRIM: 16-column MFMA tiles left out at the end of a wave's slab: the N = 300 GEMMs (out_proj, linear2) run 19 tiles = 304
//      columns instead of 20 (the loader keeps its 320-row geometry: the 20th weight piece is an out-of-range, zero-fill DMA).
template <int NTL, bool LN, bool RELU, int RES, bool BF, bool POOL = false, bool RSTD = false, bool CID = false, int TRIM = 0>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(const PPParams p) {
    constexpr int ES = BF ? 2 : 4;                 // operand element size
    constexpr int EPS = 16 / ES;                   // elements per 16-byte segment
    constexpr int BKE = 64 / ES;                   // elements per chunk
    constexpr int BN = NTL * 32;                   // loader / LDS geometry
    constexpr int BNE = BN - 16 * TRIM;            // columns a tile really computes
    constexpr int A_ST = BM * BK, W_ST = BN * BK, STAGE = A_ST + W_ST;     // floats
    constexpr int NWI = BN / 64;                                            // weight DMA instructions per wave and chunk
    // ONE __shared__ object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + 4 * BN];
    float* const Bs = lds + 2 * STAGE;             // bias of the tile's columns, double-buffered by tile parity
    float* const Gs = Bs + 2 * BN;                 // LayerNorm gamma / beta (column block 0 only)
    float* const Es = Gs + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kg = lane >> 4;          // MFMA 16x16x4: lane (i, kg) supplies row i, k slot kg
    int M = p.M, n_row_blocks = p.n_row_blocks;
    if (p.m_dev) {                                     // device-side row count (uniform: one scalar load)
        const int m = __builtin_amdgcn_readfirstlane(*p.m_dev);
        M = m < M ? (m > 0 ? m : 0) : M;
        n_row_blocks = (M + BM - 1) / BM;
    }
    const int ntiles = n_row_blocks * p.n_col_blocks;
    // Tile -> workgroup assignment.  Workgroups go to the 8 XCDs round-robin by blockIdx, and (measured,
    // tools/probes/wg_map_probe) blockIdx b and b + 256 of a 512-workgroup launch share a CU.  Each XCD walks ONE contiguous
    // range of tiles (neighbouring row panels share its L2); inside the range tiles are dealt round-robin to the XCD's
    // workgroups in blockIdx order, so the workgroups that get one tile more than the rest are the first ones -- first
    // residency slots, one per CU.  (A plain `tile = remapped id + k * grid` puts every such extra tile on the first XCDs:
    // 8 tiles on some CUs, 6 on others, 14 % of the launch spent waiting for them.)
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;                  // this workgroup's XCD and its index there
    const int nw_x = ((int)gridDim.x - xcd + 7) >> 3;                        // workgroups on this XCD
    const int qt = ntiles >> 3, rt = ntiles & 7;
    const int tbase = xcd * qt + (xcd < rt ? xcd : rt);                     // first tile of this XCD's range
    const int tcount = qt + (xcd < rt ? 1 : 0);                             // tiles in the range

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w);
    const bool gather_a = p.a_ids != nullptr;
    const int lda4 = (int)p.lda * ES, ldw4 = (int)p.ldw * ES, ldc4 = (int)p.ldc * ES;         // row pitches in bytes
    const int ldr4 = (int)p.ldr * ((RES == 3 || (RES == 2 && BF)) ? 2 : 4);

    // ---- loader state: per-lane byte offsets of the rows this lane stages ---------------------------------------------
    // DMA instruction `idx` of an operand covers image rows 16 idx .. 16 idx + 15; lane l fills row 16 idx + (l >> 2),
    // physical segment l & 3, which holds logical segment (l & 3) ^ swz4((row >> 2) & 3) = (l & 3) ^ swz4((l >> 4) & 3).
    const int srow = lane >> 2;
    const int lseg = (lane & 3) ^ swz4((lane >> 4) & 3);
    unsigned a_voff[2], w_voff[NWI];
    int aid_next[2];                                   // gathered row ids of the NEXT tile (loaded a tile ahead)
    int rid_next[2] = {0, 0};                          // RES == 2: residual row ids of this lane's two output rows, next tile
    int cid_next[2] = {0, 0};                          // CID: output rows of this lane's two result rows, next tile
    __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a);

    auto tile_rc = [&](int tile, int& row0, int& col0) {
        const int rb = tile / p.n_col_blocks;
        row0 = rb * BM;
        col0 = (tile - rb * p.n_col_blocks) * BNE;
    };
    auto prefetch_ids = [&](int tile) {                // ids of `tile` -> registers (any tile index: out of range reads 0)
        int row0, col0;
        tile_rc(tile, row0, col0);
        const bool live = tile >= 0;
        const __amdgpu_buffer_rsrc_t rs_aids = make_rsrc(p.a_ids ? (const void*)p.a_ids : (const void*)p.w);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = row0 + 16 * (wave * 2 + j) + srow;
            aid_next[j] = buf_load_i32(rs_aids, (gather_a && live && row < M) ? (unsigned)row * 4u : OOB);
        }
        if constexpr (RES == 2) {
            const __amdgpu_buffer_rsrc_t rs_rids = make_rsrc(p.res_ids ? (const void*)p.res_ids : (const void*)p.w);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int row = row0 + 32 * wave + 16 * tt + fi;
                rid_next[tt] = buf_load_i32(rs_rids, (live && row < M) ? (unsigned)row * 4u : OOB);
            }
        }
        if constexpr (CID) {
            const __amdgpu_buffer_rsrc_t rs_cids = make_rsrc(p.c_ids);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int row = row0 + 32 * wave + 16 * tt + fi;
                cid_next[tt] = buf_load_i32(rs_cids, (live && row < M) ? (unsigned)row * 4u : OOB);
            }
        }
    };
    auto loader_set_tile = [&](int tile) {             // consumes aid_next
        int row0, col0;
        tile_rc(tile, row0, col0);
        // dense A: the descriptor base moves to the tile's first row, offsets stay small; gather: base = the table
        rs_a = make_rsrc(gather_a ? (const char*)p.a : (const char*)p.a + (long)row0 * p.lda * ES);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = 16 * (wave * 2 + j) + srow;
            const unsigned rowsel = gather_a ? (unsigned)aid_next[j] : (unsigned)rl;
            a_voff[j] = (row0 + rl < M) ? rowsel * (unsigned)lda4 + (unsigned)lseg * 16u : OOB;
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            const int n = col0 + 16 * (wave * NWI + j) + srow;
            w_voff[j] = (n < p.N) ? (unsigned)n * (unsigned)ldw4 + (unsigned)lseg * 16u : OOB;
        }
    };
    // DMA of chunk c (elements c BKE .. c BKE + BKE - 1) into `stage`, in two parts: every wave issues its two A
    // instructions in front of the chunk's MFMAs, and its weight instructions in the MIDDLE of them -- wave w behind its
    // MFMA group w (see compute).  Issued all at once, the workgroup's 28 instructions queue up in the CU's one address
    // path and every wave sits ~3k cycles in "DMA issue" (a quarter of its time by s_memtime stamps); staggered, a wave
    // meets an idle path.  The weight panel is L2 resident, so the latest batch still lands before the chunk's barrier.
    auto issue_a = [&](int stage, int c) {
#if defined(LIME_PP_ABLATE) && LIME_PP_ABLATE == 2       // tools/pp_ablate.py: no operand traffic (results are garbage)
        return;
#endif
        const bool kin = c * BKE + lseg * EPS < p.K;   // K % EPS == 0: a segment is valid or not as a whole
        float* const sb = lds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dma16(rs_a, sb + (wave * 2 + j) * 256, kin ? a_voff[j] : OOB, c * 64);
    };
    auto issue_w = [&](int stage, int c) {
#if defined(LIME_PP_ABLATE) && LIME_PP_ABLATE == 2
        return;
#endif
        const bool kin = c * BKE + lseg * EPS < p.K;
        float* const sb = lds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NWI; ++j)
            dma16(rs_w, sb + A_ST + (wave * NWI + j) * 256, kin ? w_voff[j] : OOB, c * 64);
    };

    // ---- compute state ---------------------------------------------------------------------------------------------
    // v_mfma_f32_16x16x4_f32: lane (i, kg) supplies A[i][k = kg] and B[k = kg][j = i]; the MFMA's k index is only a
    // summation label, so a lane reads ONE b128 = k 4 kg .. 4 kg + 3 of its row (logical segment kg) per operand tile and
    // chunk, and the four MFMAs q = 0..3 of a tile pair element q of both fragments: together they cover the chunk's 16 k.
    constexpr int NT16 = 2 * NTL - TRIM;                           // 16-column tiles per wave
    const int pseg = (kg ^ swz4((fi >> 2) & 3)) * 4;
    const int a_off = (32 * wave + fi) * BK + pseg, w_off = A_ST + fi * BK + pseg;
    f32x4 acc[2][NT16];

    // MFMAs of the chunk in `stage`; nstage >= 0: this wave's weight DMAs of chunk nc go out behind its MFMA group `wave`.
    // The weight fragments of group g + 1 are read BEFORE the MFMAs of group g are issued (the sched_barriers that pin
    // the DMA issue would otherwise also keep hipcc from hoisting those reads, and every group would start with an
    // exposed LDS round trip).
    constexpr int GT = 4;                                          // tiles per group
    constexpr int NG = (NT16 + GT - 1) / GT;                       // groups: 5 (320 / 304 columns) or 4 (256); the last may be short
    auto compute = [&](int stage, int nstage, int nc) {
#if defined(LIME_PP_ABLATE) && LIME_PP_ABLATE == 1       // tools/pp_ablate.py: no fragment reads, no MFMAs (DMA issued up front)
        if (nstage >= 0) issue_w(nstage, nc);
        return;
#endif
        const float* sb = lds + stage * STAGE;
        f32x4 af[2], wf[2][GT];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) af[tt] = *reinterpret_cast<const f32x4*>(sb + a_off + tt * 16 * BK);
#pragma unroll
        for (int t = 0; t < GT; ++t) wf[0][t] = *reinterpret_cast<const f32x4*>(sb + w_off + t * 16 * BK);
#pragma unroll
        for (int gb = 0; gb < NG; ++gb) {
            if (gb + 1 < NG) {
#pragma unroll
                for (int t = 0; t < GT; ++t)
                    if ((gb + 1) * GT + t < NT16)
                        wf[(gb + 1) & 1][t] = *reinterpret_cast<const f32x4*>(sb + w_off + ((gb + 1) * GT + t) * 16 * BK);
            }
            if constexpr (BF) {
                // one v_mfma_f32_16x16x32_bf16 per tile: the b128 IS the lane's fragment (8 bf16 = k 8 kg .. 8 kg + 7)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int t = 0; t < GT; ++t)
                        if (gb * GT + t < NT16)
                            acc[tt][gb * GT + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wf[gb & 1][t]), __builtin_bit_cast(bf16x8, af[tt]), acc[tt][gb * GT + t], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int t = 0; t < GT; ++t)
                            if (gb * GT + t < NT16)
                                acc[tt][gb * GT + t] =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gb & 1][t][q], af[tt][q], acc[tt][gb * GT + t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);         // pins the DMA issue between the MFMA groups
            if (nstage >= 0 && gb == wave) issue_w(nstage, nc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // C layout of D^T (16x16 tile): lane (i, kg) holds output row (token) 16 tt + i of the wave's 32, columns
    // 16 t + 4 kg + r in acc[tt][t][r].
    auto acc_init = [&](int tile, int par, const int* rid, const int* cid) {
        int row0, col0;
        tile_rc(tile, row0, col0);
        float* const bs = Bs + (par ? BN : 0);
        for (int c = tid; c < BN; c += 256) bs[c] = (p.bias && col0 + c < p.N) ? p.bias[col0 + c] : 0.f;
        if constexpr (RES == 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < NT16; ++t) acc[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            constexpr bool RBF = RES == 3 || (RES == 2 && BF);                 // residual rows are bf16
            constexpr int RB = RBF ? 2 : 4;
            const __amdgpu_buffer_rsrc_t rs_rpe = make_rsrc(p.res_pe ? p.res_pe : p.w);
            __amdgpu_buffer_rsrc_t rs_res;
            if ((RES == 1 && p.res_mod <= 0) || RES == 3) rs_res = make_rsrc((const char*)p.res + (long)row0 * p.ldr * RB);
            else rs_res = make_rsrc(p.res);
            // Column validity is only tested in the last 64 columns (the dispatcher guarantees N - col0 >= BN - 64):
            // elsewhere the offset is `row offset + literal`, which hipcc cannot hoist out of the tile loop (hoisted
            // offsets and lane masks cost 100+ spilled registers).
            unsigned rof[2], pof[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int rl = 32 * wave + 16 * tt + fi, row = row0 + rl;
                unsigned ro = OOB, po = OOB;
                if (row < M) {
                    if constexpr (RES == 1) {
                        if constexpr (CID) ro = (unsigned)(cid[tt] % p.res_mod) * (unsigned)ldr4;      // dispatcher: res_mod > 0
                        else ro = (p.res_mod > 0 ? (unsigned)(row % p.res_mod) : (unsigned)rl) * (unsigned)ldr4;
                    } else if constexpr (RES == 3) {
                        ro = (unsigned)rl * (unsigned)ldr4;
                    } else {
                        ro = (unsigned)rid[tt] * (unsigned)ldr4;
                        if (p.res_pe) po = (unsigned)(row % p.res_period) * (unsigned)((int)p.ldr_pe * 4);
                    }
                }
                rof[tt] = ro == OOB ? OOB : ro + (unsigned)kg * (4u * RB);
                pof[tt] = po == OOB ? OOB : po + (unsigned)kg * 16u;
            }
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    unsigned o1 = rof[tt] + (unsigned)t * (16u * RB), o2 = pof[tt] + (unsigned)t * 64u;
                    if (t >= NT16 - 4) {
                        const bool ok = col0 + 16 * t + 4 * kg < p.N;
                        o1 = ok ? o1 : OOB;
                        o2 = ok ? o2 : OOB;
                    }
                    f32x4 x;
                    if constexpr (RBF) x = buf_load4_bf16(rs_res, o1, col0 * RB);
                    else x = buf_load4(rs_res, o1, col0 * RB);
                    if constexpr (RES == 2) x += buf_load4(rs_rpe, o2, col0 * 4);
                    acc[tt][t] = x;
                }
                if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto epilogue = [&](int tile, int par, const int* cid) {
        int row0, col0;
        tile_rc(tile, row0, col0);
        const float* const bs = Bs + (par ? BN : 0) + 4 * kg;
        const __amdgpu_buffer_rsrc_t rs_c = make_rsrc((char*)p.c + ((CID ? 0L : (long)row0 * p.ldc) + col0) * ES);
        float sum[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bs + 16 * t);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 v = acc[tt][t] + b;
                if constexpr (RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                acc[tt][t] = v;
                if constexpr (LN) {                    // columns beyond N are exact zeros (zero weights, zero bias)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sum[tt] += v[j]; sq[tt] += v[j] * v[j]; }
                }
            }
        }
        float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
        if constexpr (LN) {
            const float inv_n = 1.0f / (float)p.ln_count;        // the real columns (zero-padded ones add nothing to the sums)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float s1 = sum[tt], s2 = sq[tt];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                mean[tt] = s1 * inv_n;
                rstd[tt] = rsqrtf(fmaxf(s2 * inv_n - mean[tt] * mean[tt], 0.f) + p.ln_eps);

            }
        }
        unsigned cof[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int rl = 32 * wave + 16 * tt + fi;
            cof[tt] = (row0 + rl < M) ? (unsigned)(CID ? cid[tt] : rl) * (unsigned)ldc4 + (unsigned)kg * (4u * ES) : OOB;
        }
        if constexpr (RSTD) {                          // the training forward keeps 1 / sqrt(var + eps) for lime_layernorm_bwd_f32
            const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.ln_rstd + row0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int rl = 32 * wave + 16 * tt + fi;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd[tt]), rs_r,
                                                      (kg == 0 && row0 + rl < M) ? (unsigned)rl * 4u : OOB, 0, 0);
            }
        }
        const float* const gs = Gs + 4 * kg;
        const float* const es = Es + 4 * kg;
        if constexpr (POOL) {
            // block row (row0 + 32 wave) / 32 of C: the column means over this wave's 32 output rows (all valid or all beyond M)
            const int rl0 = 32 * wave;
            const __amdgpu_buffer_rsrc_t rs_p = make_rsrc((char*)p.c + ((long)((row0 + rl0) >> 5) * p.ldc + col0) * 4);
            const bool rows_ok = row0 + rl0 < M;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gs + 16 * t);
                const f32x4 be = *reinterpret_cast<const f32x4*>(es + 16 * t);
                f32x4 y = (acc[0][t] - mean[0]) * rstd[0] * ga + be;
                y += (acc[1][t] - mean[1]) * rstd[1] * ga + be;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = row16_sum(y[j]) * (1.0f / 32.0f);
                const bool ok = rows_ok && fi == 0 && (col0 + 16 * t + 4 * kg < p.N);
                buf_store4(y, rs_p, ok ? (unsigned)(16 * t + 4 * kg) * 4u : OOB, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                f32x4 ga, be;
                if constexpr (LN) {
                    ga = *reinterpret_cast<const f32x4*>(gs + 16 * t);
                    be = *reinterpret_cast<const f32x4*>(es + 16 * t);
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x4 y = acc[tt][t];
                    if constexpr (LN) y = (y - mean[tt]) * rstd[tt] * ga + be;
                    unsigned o = cof[tt] + (unsigned)t * (16u * ES);
                    if (t >= NT16 - 4) o = (col0 + 16 * t + 4 * kg < p.N) ? o : OOB;
                    if constexpr (BF) buf_store4_bf16(y, rs_c, o, 0);
                    else buf_store4(y, rs_c, o, 0);
                }
            }
        }
    };

    // ---- main: tiles wg, wg + nwg, ... as one stream of chunks ---------------------------------------------------------
    if constexpr (LN) {
        for (int c = tid; c < BN; c += 256) {
            Gs[c] = c < p.N ? p.ln_g[c] : 0.f;
            Es[c] = c < p.N ? p.ln_b[c] : 0.f;
        }
    }
    const int nchunk = (p.K + BKE - 1) / BKE;
    int ti = wl;                                       // index inside the XCD's range; tile = tbase + ti
    auto tile_at = [&](int i) { return i < tcount ? tbase + i : -1; };
    int tile = tile_at(ti);
    if (tile < 0) return;
    prefetch_ids(tile);
    int rid_cur[2] = {rid_next[0], rid_next[1]};
    int cid_cur[2] = {cid_next[0], cid_next[1]};
    loader_set_tile(tile);
    prefetch_ids(tile_at(ti + nw_x));
    issue_a(0, 0);
    issue_w(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int stage = 0, par = 0;
#ifdef LIME_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (; tile >= 0; ti += nw_x, tile = tile_at(ti), par ^= 1) {
        const bool more = ti + nw_x < tcount;
        const int cid_epi[2] = {cid_cur[0], cid_cur[1]};       // the loader moves on before this tile's epilogue
        acc_init(tile, par, rid_cur, cid_cur);
        PSTAMP(0)                                     // 0: accumulator init (residual loads issued)
        for (int c = 0; c + 1 < nchunk; ++c) {
            issue_a(stage ^ 1, c + 1);
            PSTAMP(1)                                 // 1: DMA issue
            compute(stage, stage ^ 1, c + 1);
            __builtin_amdgcn_sched_barrier(0);        // MFMAs touch no memory: hipcc otherwise sinks them below the wait + barrier
            PSTAMP(2)                                 // 2: fragment reads + MFMA issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PSTAMP(3)                                 // 3: DMA landed
            lds_barrier();
            PSTAMP(4)                                 // 4: barrier
            stage ^= 1;
        }
        // last chunk of the tile: the loader moves on to the next tile first
        if (more) {
            rid_cur[0] = rid_next[0];
            rid_cur[1] = rid_next[1];
            cid_cur[0] = cid_next[0];
            cid_cur[1] = cid_next[1];
            loader_set_tile(tbase + ti + nw_x);
            prefetch_ids(tile_at(ti + 2 * nw_x));
            issue_a(stage ^ 1, 0);
        }
        PSTAMP(5)                                     // 5: loader switch
        compute(stage, more ? (stage ^ 1) : -1, 0);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PSTAMP(3)
        lds_barrier();
        PSTAMP(4)
        stage ^= 1;
        // the stores retire under the next tile's first chunk; the bias image this reads is double-buffered by tile parity
        // (the next tile's acc_init rewrites the other half)
        epilogue(tile, par, cid_epi);
        PSTAMP(6)                                     // 6: epilogue
    }
#ifdef LIME_STAMPS
    if (p.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.stamps[((long)blockIdx.x * 4 + wave) * 8 + i] = tsum[i];
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

template <int NTL, bool LN, bool RELU, int RES, bool BF = false, bool POOL = false, bool RSTD = false, bool CID = false, int TRIM = 0>
int launch(const PPParams& p0, hipStream_t stream) {
    PPParams p = p0;
    p.n_row_blocks = (p.M + BM - 1) / BM;
    p.n_col_blocks = (p.N + NTL * 32 - 16 * TRIM - 1) / (NTL * 32 - 16 * TRIM);
    const long ntiles = (long)p.n_row_blocks * p.n_col_blocks;
    long nwg = 2L * num_cus();
    if (nwg > ntiles) nwg = ntiles;
#ifdef LIME_STAMPS
    p.stamps = g_pp_stamp_buf;
#endif
    hipLaunchKernelGGL((gemm_pp_kernel<NTL, LN, RELU, RES, BF, POOL, RSTD, CID, TRIM>), dim3((unsigned)nwg), dim3(256), 0, stream, p);
    lime_set_last_linear_kernel("gemm_pp_kernel<%d, %s, %s, %d, %s, %s, %s, %s, %d>", NTL, LN ? "true" : "false", RELU ? "true" : "false", RES,
                                BF ? "true" : "false", POOL ? "true" : "false", RSTD ? "true" : "false", CID ? "true" : "false", TRIM);   // as rocprofv3 prints it
    return lime_check_launch("lime_linear_f32");
}

inline bool al16(const void* ptr, long ld) { return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld % 4) == 0); }

}  // namespace

// LIME_OK / error: launched (or failed); LIME_PP_NOT_APPLICABLE: the caller takes the general kernel.
int lime_linear_pp(const lime_linear_args* a, hipStream_t s) {
    const bool has_res = a->res != nullptr, ln = a->ln_gamma != nullptr;
    const bool relu = a->act == LIME_ACT_RELU;
    if (a->M < 4096 || a->a_pe != nullptr) return LIME_PP_NOT_APPLICABLE;
    if (a->c_ids && !(has_res && !a->res_ids && a->res_mod > 0 && !ln && a->act == LIME_ACT_NONE)) return LIME_PP_NOT_APPLICABLE;
    if (a->c_ids && (long)a->M * a->ldc * 4 >= 0x7FFFFFF0L) return LIME_PP_NOT_APPLICABLE;    // scattered rows: offsets from C's base
    if (!(a->act == LIME_ACT_NONE || (relu && !has_res))) return LIME_PP_NOT_APPLICABLE;
    if (a->K % 4 || a->N % 4 || a->K < 2 * 16) return LIME_PP_NOT_APPLICABLE;      // >= 2 chunks: a barrier between the
                                                                                     // bias image's write and its read
    if (!al16(a->a, a->lda) || !al16(a->w, a->ldw) || !al16(a->c, a->ldc) || !al16(a->res, a->ldr) || !al16(a->res_pe, a->ldr_pe))
        return LIME_PP_NOT_APPLICABLE;
    if (a->bias && (uintptr_t)a->bias % 4) return LIME_PP_NOT_APPLICABLE;
    // 32-bit byte offsets: within one 128-row block of a dense operand, within the whole of a gathered / periodic one
    const long lim = 0x7FFFFFF0L;
    if (128L * a->lda * 4 >= lim || (long)a->N * a->ldw * 4 >= lim || 128L * a->ldc * 4 >= lim || 128L * a->ldr * 4 >= lim)
        return LIME_PP_NOT_APPLICABLE;
    if ((long)a->M * 4 >= lim) return LIME_PP_NOT_APPLICABLE;
    int res = 0;
    if (has_res) {
        if (a->res_ids) res = 2;
        else if (a->res_div <= 1) res = 1;
        else return LIME_PP_NOT_APPLICABLE;
        if (res == 1 && a->res_mod > 0 && (long)a->res_mod * a->ldr * 4 >= lim) return LIME_PP_NOT_APPLICABLE;
    }
    if (ln && (a->N > 320 || relu)) return LIME_PP_NOT_APPLICABLE;
    if (a->pool32 && !(ln && has_res && !a->res_ids && a->res_div <= 1 && a->M % 32 == 0)) return LIME_PP_NOT_APPLICABLE;
    // column validity is tested in the last two 32-column tiles of a block only: the last block must not be narrower
    auto tail_ok = [&](int bn) { const int last = a->N - (a->N - 1) / bn * bn; return last >= bn - 64; };

    PPParams p;
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids;
    p.w = a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = a->res; p.ldr = a->ldr; p.res_mod = a->res_mod; p.res_ids = a->res_ids;
    p.res_pe = a->res_pe; p.ldr_pe = a->ldr_pe; p.res_period = a->res_period > 0 ? a->res_period : 1;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_rstd = a->ln_rstd;
    p.c = a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.ln_count = a->N;
    p.n_row_blocks = p.n_col_blocks = 0;
    p.m_dev = a->m_dev; p.c_ids = a->c_ids;
    if (a->c_ids) {                                    // compacted in_proj: periodic residual, rows scattered by c_ids
        const int pad5c = (a->N + 319) / 320 * 320 - a->N, pad4c = (a->N + 255) / 256 * 256 - a->N;
        if (!(pad5c < pad4c) || !tail_ok(320)) return LIME_PP_NOT_APPLICABLE;
        return launch<10, false, false, 1, false, false, false, true>(p, s);
    }
    if (ln) {
        if (!tail_ok(320)) return LIME_PP_NOT_APPLICABLE;
        if (a->ln_rstd) {                              // training forward: residual + LayerNorm, rstd kept
            if (res == 0 || a->pool32) return LIME_PP_NOT_APPLICABLE;
            if (a->N <= 304 && a->N >= 304 - 64)
                return res == 1 ? launch<10, true, false, 1, false, false, true, false, 1>(p, s)
                                : launch<10, true, false, 2, false, false, true, false, 1>(p, s);
            return res == 1 ? launch<10, true, false, 1, false, false, true>(p, s) : launch<10, true, false, 2, false, false, true>(p, s);
        }
        if (a->N <= 304 && a->N >= 304 - 64 && res != 0) {   // 19 column tiles (304 columns) cover N = 300: 5 % less MFMA work, 8 registers less
            if (res == 1) return a->pool32 ? launch<10, true, false, 1, false, true, false, false, 1>(p, s)
                                           : launch<10, true, false, 1, false, false, false, false, 1>(p, s);
            return launch<10, true, false, 2, false, false, false, false, 1>(p, s);
        }
        if (res == 0) return launch<10, true, false, 0>(p, s);
        if (res == 1) return a->pool32 ? launch<10, true, false, 1, false, true>(p, s) : launch<10, true, false, 1>(p, s);
        return launch<10, true, false, 2>(p, s);
    }
    if (res == 2) return LIME_PP_NOT_APPLICABLE;
    // the tile width (256 / 320) that pads N least
    const int pad5 = (a->N + 319) / 320 * 320 - a->N, pad4 = (a->N + 255) / 256 * 256 - a->N;
    if (tail_ok(pad5 < pad4 ? 320 : 256)) {
        if (pad5 < pad4) {
            if (res == 1) return launch<10, false, false, 1>(p, s);
            return relu ? launch<10, false, true, 0>(p, s) : launch<10, false, false, 0>(p, s);
        }
        if (res == 1) return launch<8, false, false, 1>(p, s);
        return relu ? launch<8, false, true, 0>(p, s) : launch<8, false, false, 0>(p, s);
    }
    // N whose last 256 / 320-column block would be too narrow (N = 400, 1200, 200 of the layers around the encoders at large
    // batch): 19-tile (304) or 13-tile (208) slabs of the same loaders -- the trimmed tiles are simply not computed
    const int pad304 = (a->N + 303) / 304 * 304 - a->N, pad208 = (a->N + 207) / 208 * 208 - a->N;
    const bool ok304 = tail_ok(304), ok208 = tail_ok(208);
    if (ok304 && (!ok208 || pad304 <= pad208)) {
        if (res == 1) return launch<10, false, false, 1, false, false, false, false, 1>(p, s);
        return relu ? launch<10, false, true, 0, false, false, false, false, 1>(p, s) : launch<10, false, false, 0, false, false, false, false, 1>(p, s);
    }
    if (ok208) {
        if (res == 1) return launch<8, false, false, 1, false, false, false, false, 3>(p, s);
        return relu ? launch<8, false, true, 0, false, false, false, false, 3>(p, s) : launch<8, false, false, 0, false, false, false, false, 3>(p, s);
    }
    return LIME_PP_NOT_APPLICABLE;
}

// ---- bf16 operands (BASELINE config 3: bf16 MFMA, fp32 accumulate / LayerNorm) ------------------------------------------
extern "C" int lime_linear_bf16(const lime_linear_bf16_args* a, void* stream) {
    LIME_REQUIRE(a != nullptr, LIME_ERR_BAD_ARG, "lime_linear_bf16: args is NULL");
    LIME_REQUIRE(a->a && a->w && a->c, LIME_ERR_BAD_ARG, "lime_linear_bf16: a, w and c must be non-NULL");
    LIME_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, LIME_ERR_BAD_ARG, "lime_linear_bf16: bad dims M=%d N=%d K=%d", a->M, a->N, a->K);
    LIME_REQUIRE(a->K % 8 == 0 && a->K >= 64 && a->N % 4 == 0, LIME_ERR_UNSUPPORTED,
                 "lime_linear_bf16: K must be a multiple of 8 and >= 64, N a multiple of 4 (pad with zero columns / rows)");
    LIME_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, LIME_ERR_BAD_ARG, "lime_linear_bf16: leading dimension < row");
    auto al = [](const void* ptr, long ld, int elem, int bytes) { return ptr == nullptr || ((uintptr_t)ptr % bytes == 0 && (ld * elem) % bytes == 0); };
    LIME_REQUIRE(al(a->a, a->lda, 2, 16) && al(a->w, a->ldw, 2, 16), LIME_ERR_BAD_ARG,
                 "lime_linear_bf16: a / w rows must be 16-byte aligned (lda, ldw multiples of 8)");
    LIME_REQUIRE(a->reserved == 0 && (a->pool32 == 0 || a->pool32 == 1), LIME_ERR_BAD_ARG, "lime_linear_bf16: pool32 must be 0 / 1, reserved 0");
    LIME_REQUIRE(a->pool32 ? al(a->c, a->ldc, 4, 16) : al(a->c, a->ldc, 2, 8), LIME_ERR_BAD_ARG,
                 "lime_linear_bf16: c rows must be 8-byte aligned (ldc multiple of 4); pool32: 16-byte aligned fp32 rows");
    LIME_REQUIRE(!a->pool32 || (a->ln_gamma && a->res_kind == 3 && a->M % 32 == 0), LIME_ERR_UNSUPPORTED,
                 "lime_linear_bf16: pool32 needs the LayerNorm epilogue with a bf16 residual and M %% 32 == 0");
    LIME_REQUIRE(a->res_kind >= 0 && a->res_kind <= 3 && (a->res_kind == 0) == (a->res == nullptr), LIME_ERR_BAD_ARG,
                 "lime_linear_bf16: res_kind %d does not match res", a->res_kind);
    LIME_REQUIRE(a->act == LIME_ACT_NONE || (a->act == LIME_ACT_RELU && a->res_kind == 0), LIME_ERR_UNSUPPORTED,
                 "lime_linear_bf16: activation none, or ReLU without residual");
    const bool ln = a->ln_gamma != nullptr, relu = a->act == LIME_ACT_RELU;
    LIME_REQUIRE(!ln || (a->ln_beta && a->N <= 320 && !relu && a->ln_count > 0 && a->ln_count <= a->N), LIME_ERR_UNSUPPORTED,
                 "lime_linear_bf16: LayerNorm needs beta, N <= 320, no activation, 0 < ln_count <= N");
    if (a->res_kind == 1) LIME_REQUIRE(al(a->res, a->ldr, 4, 16) && a->ldr >= a->N, LIME_ERR_BAD_ARG, "lime_linear_bf16: fp32 residual misaligned");
    if (a->res_kind == 2)
        LIME_REQUIRE(a->res_ids && al(a->res, a->ldr, 2, 8) && a->ldr >= a->N && (!a->res_pe || (al(a->res_pe, a->ldr_pe, 4, 16) &&
                     a->res_period > 0 && a->ldr_pe >= a->N)), LIME_ERR_BAD_ARG, "lime_linear_bf16: gathered residual needs ids, aligned rows, res_period");
    if (a->res_kind == 3) LIME_REQUIRE(al(a->res, a->ldr, 2, 8) && a->ldr >= a->N, LIME_ERR_BAD_ARG, "lime_linear_bf16: bf16 residual misaligned");
    if (a->M == 0) return LIME_OK;
    const long lim = 0x7FFFFFF0L;
    LIME_REQUIRE(128L * a->lda * 2 < lim && (long)a->N * a->ldw * 2 < lim && 128L * a->ldc * 4 < lim && 128L * a->ldr * 4 < lim &&
                 (long)a->M * 4 < lim, LIME_ERR_UNSUPPORTED, "lime_linear_bf16: operand too large for 32-bit offsets");
    const bool wide = ((a->N + 319) / 320 * 320 - a->N) < ((a->N + 255) / 256 * 256 - a->N);
    const int bn = (ln || wide) ? 320 : 256;
    LIME_REQUIRE(a->N - (a->N - 1) / bn * bn >= bn - 64, LIME_ERR_UNSUPPORTED,
                 "lime_linear_bf16: the last %d-column block of N=%d is narrower than %d columns (pad N)", bn, a->N, bn - 64);

    PPParams p;
    p.a = (const float*)a->a; p.lda = a->lda; p.a_ids = a->a_ids;
    p.w = (const float*)a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = (const float*)a->res; p.ldr = a->ldr; p.res_mod = a->res_mod; p.res_ids = a->res_ids;
    p.res_pe = a->res_pe; p.ldr_pe = a->ldr_pe; p.res_period = a->res_period > 0 ? a->res_period : 1;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_rstd = nullptr;
    p.c = (float*)a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.ln_count = ln ? a->ln_count : a->N;
    p.n_row_blocks = p.n_col_blocks = 0;
    p.m_dev = a->m_dev; p.c_ids = a->c_ids;
    if (a->c_ids) {
        LIME_REQUIRE(a->res_kind == 1 && a->res_mod > 0 && !ln && !relu && bn == 320 && (long)a->M * a->ldc * 2 < lim, LIME_ERR_UNSUPPORTED,
                     "lime_linear_bf16: c_ids needs the fp32 periodic residual (res_kind 1, res_mod > 0), no LayerNorm / activation, N in "
                     "320-column blocks and M * ldc * 2 < 2 GB");
        return launch<10, false, false, 1, true, false, false, true>(p, (hipStream_t)stream);
    }
    hipStream_t s = (hipStream_t)stream;
    if (ln) {
        switch (a->res_kind) {
            case 0: return launch<10, true, false, 0, true>(p, s);
            case 2: return (a->N <= 304 && a->N >= 240) ? launch<10, true, false, 2, true, false, false, false, 1>(p, s)
                                                         : launch<10, true, false, 2, true>(p, s);
            case 3:
                if (a->N <= 304 && a->N >= 240)
                    return a->pool32 ? launch<10, true, false, 3, true, true, false, false, 1>(p, s)
                                     : launch<10, true, false, 3, true, false, false, false, 1>(p, s);
                return a->pool32 ? launch<10, true, false, 3, true, true>(p, s) : launch<10, true, false, 3, true>(p, s);
            default: break;
        }
        LIME_REQUIRE(false, LIME_ERR_UNSUPPORTED, "lime_linear_bf16: LayerNorm with an fp32 residual is not built");
    }
    LIME_REQUIRE(a->res_kind <= 1, LIME_ERR_UNSUPPORTED, "lime_linear_bf16: bf16 / gathered residuals are built with LayerNorm only");
    if (bn == 320) {
        if (a->res_kind == 1) return launch<10, false, false, 1, true>(p, s);
        return relu ? launch<10, false, true, 0, true>(p, s) : launch<10, false, false, 0, true>(p, s);
    }
    if (a->res_kind == 1) return launch<8, false, false, 1, true>(p, s);
    return relu ? launch<8, false, true, 0, true>(p, s) : launch<8, false, false, 0, true>(p, s);
}
