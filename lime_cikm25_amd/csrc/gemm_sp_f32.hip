// lime_linear_f32, big-M instantiations on the bf16 matrix cores with fp32-level arithmetic ("split product").
//
// gfx950 multiplies fp32 operands on the matrix cores at 1/16 of the bf16 rate (v_mfma_f32_16x16x4_f32: 157.3 TFLOP/s, no
// xf32 / tf32 form), and the four GEMMs of an encoder layer are bound by exactly that pipe (gemm_pp_f32.hip: 0.55-0.7 of it).
// This kernel keeps fp32 operands in memory and in LDS and splits each fragment IN REGISTERS into three bf16 terms
//       x = hi + mid + lo,    hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)     (the subtractions are exact)
// -- 8 + 8 + 8 mantissa bits with round-to-nearest residuals: |x - (hi + mid + lo)| <= 2^-25 |x| -- and forms
//       a w  ~=  hi hi + hi mid + mid hi + mid mid + hi lo + lo hi
// with six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block (fp32 accumulation, the small terms first).  The dropped terms
// (mid lo, lo mid, lo lo) are below 2^-24 |a w|: the product carries the error of ONE fp32 rounding, which is what the fp32 MFMA
// (an fma chain) commits per product too.  Measured against an fp64 product (tools/probes/split_probe.hip, K = 32 .. 4096):
// max |c - ref| / sum |a w| = 0.4 .. 1.8e-7 for this scheme, 1.0 .. 1.4e-7 for v_mfma_f32_16x16x4_f32.  bf16 has fp32's exponent
// range, so nothing can overflow or flush that would not in fp32 (an fp16 split would need half the products but not survive
// arbitrary checkpoints).  Six bf16 MFMAs cost 96 cycles per block against 256 for the eight fp32 ones.
//
// Structure (one eight-wave workgroup per CU, 157 KB of LDS):
//   * tile 256 rows x 320 (CT = 10) or 256 (CT = 8) columns; wave w owns rows 64 (w & 3) .. + 63 and the column half w >> 2
//     (4 x CT accumulator tiles of 16 x 16 = 160 registers): a weight fragment's split (~40 VALU instructions) serves four row
//     tiles = 24 MFMAs, an activation fragment's CT column tiles.  The two waves of a SIMD (w, w + 4) share rows.
//   * K chunks are 32 deep (one bf16 MFMA's k), two LDS stages filled by LDS-DMA (buffer_load_dwordx4 ... lds) while the
//     previous chunk is multiplied, a wave's nine 1 KB pieces issued two per column tile in the first half of the chunk (they
//     have the second half to land); one barrier per chunk.
//   * LDS image: [row][128 bytes] -- eight 16-byte segments, a lane's fragment is the 32-byte pair kg (k = 8 kg .. 8 kg + 7).
//     A DMA piece writes 1 KB linearly, so rows cannot be padded: the pair index is XOR-swizzled by (row >> 1) & 3 (on the DMA's
//     per-lane SOURCE address and on the read address), and lanes with odd kg read the two halves of their pair in the opposite
//     order.  That makes every ds_read_b128 conflict free over the hardware's 16-lane groups ({0-3, 12-15, 20-27}, ...); the
//     k order inside a lane group is only a summation label, and both operands use the same one.
//   * epilogue as in gemm_pp_f32.hip (transposed product D^T = W A^T: a lane holds four consecutive columns of one row):
//     bias / ReLU / residual in the accumulators / LayerNorm (row sums of the two column halves meet through 4 KB of LDS) /
//     32-row block means (pool32) / compacted-row scatter (c_ids) / device-side row count (m_dev).
//   * gathered A rows, residual rows and scatter rows: the ids of the NEXT tile are fetched during the first chunk of the
//     current one and parked in LDS (no per-lane id registers: the accumulators and the split fragments use 220 of the 256).
// A variant with two independent four-wave workgroups per CU (128-row tiles, the weight stage refilled in two halves so that two
// workgroups' LDS fit) measured slower end to end: two barriers per chunk and half a chunk for a DMA piece to land cost more than
// the overlapped epilogues returned (profiles/r03_notes.md).
#include "common.h"
#include "gemm_pp.h"
#include "lds_dma.h"
#include "split_mfma.h"

using namespace lime_dev;

#ifdef LIME_STAMPS
// Diagnostic build only (tools/gemm_stamps.py): per-wave s_memtime sums of the main-loop segments; never in liblime_hip.so.
static unsigned long long* g_sp_stamp_buf = nullptr;
extern "C" void lime_debug_set_sp_stamp_buffer(unsigned long long* p) { g_sp_stamp_buf = p; }
#define SSTAMP(i)                                                           \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 \
        tsum[i] += t_ - tlast;                                              \
        tlast = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
    }
#else
#define SSTAMP(i)
#endif

namespace {

constexpr int BM = 256;                    // rows per tile: 4 row waves x 64
constexpr int ROWB = 128;                  // bytes per image row = one 32-deep fp32 chunk
constexpr int A_BYTES = BM * ROWB;

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int buf_load_i32(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}

// the three-term split and the six-MFMA product: split_mfma.h
using Split = SplitFrag;
__device__ __forceinline__ Split split8(const f32x4 x0, const f32x4 x1) { return split_frag(x0, x1); }
__device__ __forceinline__ f32x4 mfma6(const Split& w, const Split& a, f32x4 c) { return split_mfma16(w, a, c); }

// CT: 16-column tiles per wave (tile width 32 CT).  LN / RELU / RES / POOL / RSTD / CID as in gemm_pp_kernel:
// RES 0 none, 1 dense fp32 residual rows (r, or r % res_mod; with CID: c_ids[r] % res_mod), 2 rows gathered by res_ids + the fp32
// positional table, 3 no residual: `res` holds the forward ReLU output whose sign gates the result (LIME_ACT_RELU_GRAD).
template <int CT, bool LN, bool RELU, int RES, bool POOL, bool RSTD, bool CID>
__global__ __launch_bounds__(512, 2) void gemm_sp_kernel(const PPParams p) {
    constexpr int BN = 32 * CT;
    constexpr int W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int NWP = BN / 64;                                   // weight DMA pieces per wave and chunk (8 rows each): 5 / 4
    constexpr int NPIECE = 4 + NWP;
    constexpr int O_BS = 2 * STAGE, O_GS = O_BS + 2 * BN * 4, O_ES = O_GS + BN * 4, O_LNP = O_ES + BN * 4;
    constexpr int O_AROW = O_LNP + (LN ? BM * 2 * 8 : 0), O_IDS = O_AROW + 2 * BM * 4;
    constexpr int TOTAL = O_IDS + ((RES == 2 || CID) ? 2 * BM * 4 : 0);
    static_assert(TOTAL <= 163840, "LDS budget");
    // ONE __shared__ object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) unsigned char lds[TOTAL];
    float* const Bs = reinterpret_cast<float*>(lds + O_BS);        // bias of the tile's columns, double-buffered by tile parity
    float* const Gs = reinterpret_cast<float*>(lds + O_GS);
    float* const Es = reinterpret_cast<float*>(lds + O_ES);
    f32x2* const lnp = reinterpret_cast<f32x2*>(lds + O_LNP);      // [row][column half] partial (sum, sum of squares)
    unsigned* const arow = reinterpret_cast<unsigned*>(lds + O_AROW);   // [2][256] byte offset of each tile row's A row (or OOB)
    int* const ids = reinterpret_cast<int*>(lds + O_IDS);          // [2][256] residual row ids (RES == 2) or output rows (CID)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;
    const int fi = lane & 15, kg = lane >> 4;
    int M = p.M, n_row_blocks = p.n_row_blocks;
    if (p.m_dev) {
        const int m = __builtin_amdgcn_readfirstlane(*p.m_dev);
        M = m < M ? (m > 0 ? m : 0) : M;
        n_row_blocks = (M + BM - 1) / BM;
    }
    const int ntiles = n_row_blocks * p.n_col_blocks;
    // tile -> workgroup: each XCD walks one contiguous tile range (see gemm_pp_f32.hip), tiles dealt round-robin inside it
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int nw_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int qt = ntiles >> 3, rt = ntiles & 7;
    const int tbase = xcd * qt + (xcd < rt ? xcd : rt);
    const int tcount = qt + (xcd < rt ? 1 : 0);

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w);
    const bool gather_a = p.a_ids != nullptr;
    const int lda4 = (int)p.lda * 4, ldw4 = (int)p.ldw * 4, ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4;

    auto tile_rc = [&](int tile, int& row0, int& col0) {
        const int rb = tile / p.n_col_blocks;
        row0 = rb * BM;
        col0 = (tile - rb * p.n_col_blocks) * BN;
    };

    // ---- per-tile row lists in LDS -----------------------------------------------------------------------------------
    // threads 0 .. 255: A row offsets; threads 256 .. 511: residual / scatter ids.  `load_rows` issues the global loads,
    // `park_rows` writes them to half `sel` of the lists (the caller separates the two by a chunk of MFMAs).
    int row_val = 0;
    auto load_rows = [&](int tile) {
        int row0, col0;
        tile_rc(tile, row0, col0);
        const int rl = tid & 255, row = row0 + rl;
        const bool ok = tile >= 0 && row < M;
        if (tid < 256) {
            if (gather_a) {
                const __amdgpu_buffer_rsrc_t rs_ids = make_rsrc(p.a_ids);
                row_val = buf_load_i32(rs_ids, ok ? (unsigned)row * 4u : OOB);
            } else {
                row_val = rl;
            }
            if (!ok) row_val = -1;
        } else if constexpr (RES == 2 || CID) {
            const __amdgpu_buffer_rsrc_t rs_ids = make_rsrc(RES == 2 ? (const void*)p.res_ids : (const void*)p.c_ids);
            row_val = buf_load_i32(rs_ids, ok ? (unsigned)row * 4u : OOB);
        }
    };
    auto park_rows = [&](int sel) {
        const int rl = tid & 255;
        if (tid < 256) arow[sel * BM + rl] = row_val < 0 ? OOB : (unsigned)row_val * (unsigned)lda4;
        else if constexpr (RES == 2 || CID) ids[sel * BM + rl] = row_val;
    };

    // ---- loader --------------------------------------------------------------------------------------------------------
    // piece idx of an operand covers image rows 8 idx .. 8 idx + 7: lane l fills row 8 idx + (l >> 3), physical segment l & 7 =
    // pair (l & 7) >> 1, half l & 1; the pair holds logical pair ((l & 7) >> 1) ^ ((row >> 1) & 3), and (row >> 1) & 3 = (l >> 4) & 3.
    const int srow = lane >> 3;
    const int lseg = 2 * ((((lane & 7) >> 1) ^ ((lane >> 4) & 3))) + (lane & 1);
    int l_row0 = 0, l_col0 = 0;                        // the tile the loader is on
    auto loader_set_tile = [&](int tile) { tile_rc(tile, l_row0, l_col0); };
    auto issue_piece = [&](int q, int stage, int c, int sel) {          // q: 0 .. 3 A pieces, 4 .. NPIECE - 1 weight pieces
        const bool kin = c * 32 + lseg * 4 < p.K;      // K % 4 == 0: a segment is valid or not as a whole
        unsigned char* const sb = lds + stage * STAGE;
        if (q < 4) {
            const int idx = wave * 4 + q;
            const unsigned ro = arow[sel * BM + 8 * idx + srow];
            // dense A: the descriptor base moves to the tile's first row, offsets stay small; gather: base = the table
            const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(gather_a ? (const char*)p.a : (const char*)p.a + (long)l_row0 * p.lda * 4);
            dma16(rs_a, sb + idx * 1024, kin ? ro + (unsigned)lseg * 16u : OOB, c * ROWB);          // OOB + 16 lseg stays out of range
        } else {
            const int idx = wave * NWP + (q - 4);
            const int n = l_col0 + 8 * idx + srow;
            dma16(rs_w, sb + A_BYTES + idx * 1024, (kin && n < p.N) ? (unsigned)n * (unsigned)ldw4 + (unsigned)lseg * 16u : OOB, c * ROWB);
        }
    };

    // ---- compute -------------------------------------------------------------------------------------------------------
    // lane (fi, kg) of a 16-row tile reads pair kg of row fi: two b128, halves in the order (kg & 1, 1 - (kg & 1)).
    const int pair_off = ((kg ^ ((fi >> 1) & 3)) * 32);
    const int h0 = (kg & 1) * 16, h1 = 16 - h0;
    const int a_off = (64 * wr + fi) * ROWB + pair_off;
    const int w_off = A_BYTES + (16 * CT * wc + fi) * ROWB + pair_off;
    f32x4 acc[4][CT];
    int nct = CT;                                      // column tiles of this wave that hold real columns (set per tile)
    int crow[4] = {0, 0, 0, 0};                        // CID: output rows (RES == 2: residual rows) of this lane's four result rows (read in acc_init, used in
                                                       // the epilogue: the row lists' half is re-parked for the next tile in between)

    auto compute = [&](int stage, int nstage, int nc, int sel) {        // nstage >= 0: the DMA pieces of chunk nc go out between the tiles
        const unsigned char* const sb = lds + stage * STAGE;
        Split a[4];
        // the first column tile is multiplied while the activation fragments are still being split: row tile i's six MFMAs
        // go out behind split i (all four splits first would leave the matrix pipe idle for ~190 VALU instructions per chunk)
        f32x4 r0 = *reinterpret_cast<const f32x4*>(sb + w_off + h0), r1 = *reinterpret_cast<const f32x4*>(sb + w_off + h1);
        f32x4 x0[4], x1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x0[i] = *reinterpret_cast<const f32x4*>(sb + a_off + i * 16 * ROWB + h0);
            x1[i] = *reinterpret_cast<const f32x4*>(sb + a_off + i * 16 * ROWB + h1);
        }
        // software pipeline over the column tiles: tile j + 1's weight fragment is read and split in the same basic block as tile j's 24
        // MFMAs (hipcc keeps the MFMAs together and the split behind them -- sched_group_barrier requests for a 1 : 2 interleave were
        // not honoured; the partner wave of the SIMD has the pipe meanwhile, and the loop sits at 0.89 of the bf16 pipe's sustained rate
        // either way: profiles/r03_notes.md)
        Split w = split8(r0, r1);
        if (CT > 1) {
            r0 = *reinterpret_cast<const f32x4*>(sb + w_off + 16 * ROWB + h0);
            r1 = *reinterpret_cast<const f32x4*>(sb + w_off + 16 * ROWB + h1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = split8(x0[i], x1[i]);
            if (0 < nct) acc[i][0] = mfma6(w, a[i], acc[i][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nstage >= 0) {
            issue_piece(0, nstage, nc, sel);
            if (1 < NPIECE) issue_piece(1, nstage, nc, sel);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (CT > 1) w = split8(r0, r1);
#pragma unroll
        for (int j = 1; j < CT; ++j) {
            Split wn = w;
            if (j < nct) {                             // (one basic block: the reads, this tile's MFMAs and the next tile's split)
                if (j + 1 < CT) {                      // the next tile's fragment: read, and split under this tile's MFMAs
                    r0 = *reinterpret_cast<const f32x4*>(sb + w_off + (j + 1) * 16 * ROWB + h0);
                    r1 = *reinterpret_cast<const f32x4*>(sb + w_off + (j + 1) * 16 * ROWB + h1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma6(w, a[i], acc[i][j]);
                if (j + 1 < CT) wn = split8(r0, r1);
            }
            __builtin_amdgcn_sched_barrier(0);         // pins the DMA issue between the column tiles
            if (nstage >= 0) {                         // two pieces per column tile: all out in the first half of the chunk
                if (2 * j < NPIECE) issue_piece(2 * j, nstage, nc, sel);
                if (2 * j + 1 < NPIECE) issue_piece(2 * j + 1, nstage, nc, sel);
            }
            __builtin_amdgcn_sched_barrier(0);
            w = wn;
        }
    };

    // C layout (16 x 16 tile of D^T): lane (fi, kg) holds row 16 i + fi of the wave's 64, columns 16 t + 4 kg + r in acc[i][t][r]
    const int cw0 = 16 * CT * wc;                      // first column of this wave inside the tile
    auto acc_init = [&](int tile, int par, int sel) {
        int row0, col0;
        tile_rc(tile, row0, col0);
        {
            const int left = p.N - col0 - cw0;
            nct = __builtin_amdgcn_readfirstlane(left <= 0 ? 0 : (left >= 16 * CT ? CT : (left + 15) >> 4));
        }
        float* const bs = Bs + (par ? BN : 0);
        for (int c = tid; c < BN; c += 512) bs[c] = (p.bias && col0 + c < p.N) ? p.bias[col0 + c] : 0.f;
        if constexpr (RES == 2) {                   // the residual row ids of this lane's four rows: kept in registers, the lists'
#pragma unroll                                   // half is re-parked for the next tile before the epilogue reads them
            for (int i = 0; i < 4; ++i) crow[i] = ids[sel * BM + 64 * wr + 16 * i + fi];
        }
        if constexpr (RES == 0 || RES == 2 || RES == 3) {       // RES == 2: the gathered rows + positional rows are added in the epilogue (two
#pragma unroll                                   // loads per element into the accumulators here spilled 186 registers)
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const __amdgpu_buffer_rsrc_t rs_rpe = make_rsrc(p.res_pe ? p.res_pe : p.w);
            __amdgpu_buffer_rsrc_t rs_res;
            if (RES == 1 && p.res_mod <= 0 && p.res_div <= 1) rs_res = make_rsrc((const char*)p.res + (long)row0 * p.ldr * 4);
            else rs_res = make_rsrc(p.res);
            unsigned rof[4], pof[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 64 * wr + 16 * i + fi, row = row0 + rl;
                unsigned ro = OOB, po = OOB;
                if (row < M) {
                    if constexpr (RES == 1) {
                        if constexpr (CID) {
                            crow[i] = ids[sel * BM + rl];
                            ro = (unsigned)(crow[i] % p.res_mod) * (unsigned)ldr4;       // dispatcher: res_mod > 0
                        } else ro = (p.res_mod > 0 ? (unsigned)(row % p.res_mod) : (p.res_div > 1 ? (unsigned)(row / p.res_div) : (unsigned)rl)) * (unsigned)ldr4;
                    } else {
                        ro = (unsigned)ids[sel * BM + rl] * (unsigned)ldr4;
                        if (p.res_pe) po = (unsigned)(row % p.res_period) * (unsigned)((int)p.ldr_pe * 4);
                    }
                }
                rof[i] = ro == OOB ? OOB : ro + (unsigned)(cw0 + 4 * kg) * 4u;
                pof[i] = po == OOB ? OOB : po + (unsigned)(cw0 + 4 * kg) * 4u;
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const bool ok = col0 + cw0 + 16 * t + 4 * kg < p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 x = buf_load4(rs_res, ok ? rof[i] + (unsigned)t * 64u : OOB, col0 * 4);
                    if constexpr (RES == 2) x += buf_load4(rs_rpe, ok ? pof[i] + (unsigned)t * 64u : OOB, col0 * 4);
                    acc[i][t] = x;
                }
                if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto epilogue = [&](int tile, int par, int sel) {
        int row0, col0;
        tile_rc(tile, row0, col0);
        const float* const bs = Bs + (par ? BN : 0) + cw0 + 4 * kg;
        float sum[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned rof[4] = {OOB, OOB, OOB, OOB}, pof[4] = {OOB, OOB, OOB, OOB};
        const __amdgpu_buffer_rsrc_t rs_res2 = make_rsrc(p.res ? (const void*)p.res : (const void*)p.w);
        const __amdgpu_buffer_rsrc_t rs_rpe2 = make_rsrc(p.res_pe ? (const void*)p.res_pe : (const void*)p.w);
        const __amdgpu_buffer_rsrc_t rs_c = make_rsrc((char*)p.c + ((CID ? 0L : (long)row0 * p.ldc) + col0) * 4);
        unsigned cof[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = 64 * wr + 16 * i + fi;
            cof[i] = (row0 + rl < M) ? (unsigned)(CID ? crow[i] : rl) * (unsigned)ldc4 + (unsigned)(cw0 + 4 * kg) * 4u : OOB;
        }
        if constexpr (RES == 3) {                   // the forward activation h of this tile: rows row0 + rl of p.res
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 64 * wr + 16 * i + fi;
                if (row0 + rl < M) rof[i] = (unsigned)(row0 + rl) * (unsigned)ldr4 + (unsigned)(cw0 + 4 * kg) * 4u;
            }
        }
        if constexpr (RES == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 64 * wr + 16 * i + fi, row = row0 + rl;
                if (row < M) {
                    rof[i] = (unsigned)crow[i] * (unsigned)ldr4 + (unsigned)(cw0 + 4 * kg) * 4u;
                    if (p.res_pe) pof[i] = (unsigned)(row % p.res_period) * (unsigned)((int)p.ldr_pe * 4) + (unsigned)(cw0 + 4 * kg) * 4u;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bs + 16 * t);
            f32x4 rr[4];
            if constexpr (RES == 2) {
                const bool okc = col0 + cw0 + 16 * t + 4 * kg < p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rr[i] = buf_load4(rs_res2, okc ? rof[i] + (unsigned)t * 64u : OOB, col0 * 4) +
                            buf_load4(rs_rpe2, okc ? pof[i] + (unsigned)t * 64u : OOB, col0 * 4);
            }
            if constexpr (RES == 3) {
                const bool okc = col0 + cw0 + 16 * t + 4 * kg < p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = buf_load4(rs_res2, okc ? rof[i] + (unsigned)t * 64u : OOB, col0 * 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][t] + b;
                if constexpr (RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    if constexpr (!LN && RES == 0) {
                        if (p.drop.thresh != 0) {      // nn.Dropout behind the ReLU (training mode): one hash per four consecutive columns
                            const unsigned keep = lime_keep4(p.drop, ((uint64_t)(row0 + 64 * wr + 16 * i + fi) * (uint64_t)p.N +
                                                                      (uint64_t)(col0 + cw0 + 16 * t + 4 * kg)) >> 2);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = (keep >> j) & 1u ? v[j] * p.drop.scale : 0.f;
                        }
                    }
                } else if constexpr (!LN && RES != 1) {        // Attention.affine1 (tanh), gates (sigmoid): uniform run-time choice
                    if (p.act == LIME_ACT_TANH) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = tanhf(v[j]);
                    } else if (p.act == LIME_ACT_SIGMOID) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = lime_sigmoid(v[j]);
                    }
                }
                if constexpr (RES == 2) v += rr[i];               // act(acc + bias) + residual, the documented order
                if constexpr (RES == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rr[i][j] > 0.f ? v[j] * p.act_scale : 0.f;
                }
                if constexpr (!LN) {                   // nothing of the row is needed any more: store now (the accumulators die here)
                    buf_store4(v, rs_c, (col0 + cw0 + 16 * t + 4 * kg < p.N) ? cof[i] + (unsigned)t * 64u : OOB, 0);
                }
                acc[i][t] = v;
                if constexpr (LN) {                    // columns beyond N are exact zeros (zero weights, zero bias, no residual)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sum[i] += v[j]; sq[i] += v[j] * v[j]; }
                }
            }
            if constexpr (RES == 2 || RES == 3) __builtin_amdgcn_sched_barrier(0);      // one column tile's residual loads at a time
        }
        float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LN) {
            // a row's columns are spread over the two column halves: the halves' partial sums meet in LDS
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s1 = sum[i], s2 = sq[i];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                sum[i] = s1; sq[i] = s2;
                if (kg == 0) lnp[(64 * wr + 16 * i + fi) * 2 + wc] = f32x2{s1, s2};
            }
            lds_barrier();
            const float inv_n = 1.0f / (float)p.ln_count;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 o = lnp[(64 * wr + 16 * i + fi) * 2 + (wc ^ 1)];
                // the same association in both halves (half 0 + half 1): the two waves of a row agree bit for bit
                const float s1 = wc ? o[0] + sum[i] : sum[i] + o[0], s2 = wc ? o[1] + sq[i] : sq[i] + o[1];
                mean[i] = s1 * inv_n;
                rstd[i] = rsqrtf(fmaxf(s2 * inv_n - mean[i] * mean[i], 0.f) + p.ln_eps);
            }
        }
        if constexpr (RSTD) {
            const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.ln_rstd + row0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 64 * wr + 16 * i + fi;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd[i]), rs_r,
                                                      (kg == 0 && wc == 0 && row0 + rl < M) ? (unsigned)rl * 4u : OOB, 0, 0);
            }
        }
        const float* const gs = Gs + cw0 + 4 * kg;
        const float* const es = Es + cw0 + 4 * kg;
        if constexpr (POOL) {
            // block rows (row0 + 64 wr) / 32 and + 1 of C: the column means over 32 output rows (all valid or all beyond M)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int rl0 = 64 * wr + 32 * b;
                const __amdgpu_buffer_rsrc_t rs_p = make_rsrc((char*)p.c + ((long)((row0 + rl0) >> 5) * p.ldc + col0) * 4);
                const bool rows_ok = row0 + rl0 < M;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(gs + 16 * t);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(es + 16 * t);
                    f32x4 y = (acc[2 * b][t] - mean[2 * b]) * rstd[2 * b] * ga + be;
                    y += (acc[2 * b + 1][t] - mean[2 * b + 1]) * rstd[2 * b + 1] * ga + be;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = row16_sum(y[j]) * (1.0f / 32.0f);
                    const bool ok = rows_ok && fi == 0 && (col0 + cw0 + 16 * t + 4 * kg < p.N);
                    buf_store4(y, rs_p, ok ? (unsigned)(cw0 + 16 * t + 4 * kg) * 4u : OOB, 0);
                }
            }
        } else if constexpr (LN) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gs + 16 * t);
                const f32x4 be = *reinterpret_cast<const f32x4*>(es + 16 * t);
                const bool ok = col0 + cw0 + 16 * t + 4 * kg < p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 y = (acc[i][t] - mean[i]) * rstd[i] * ga + be;
                    buf_store4(y, rs_c, ok ? cof[i] + (unsigned)t * 64u : OOB, 0);
                }
            }
        }
    };

    // ---- main: tiles wl, wl + nw_x, ... of this XCD's range as one stream of chunks ------------------------------------
    if constexpr (LN) {
        for (int c = tid; c < BN; c += 512) {
            Gs[c] = c < p.N ? p.ln_g[c] : 0.f;
            Es[c] = c < p.N ? p.ln_b[c] : 0.f;
        }
    }
    const int nchunk = (p.K + 31) / 32;
    int ti = wl;
    auto tile_at = [&](int i) { return i < tcount ? tbase + i : -1; };
    int tile = tile_at(ti);
    if (tile < 0) return;
    load_rows(tile);
    park_rows(0);
    loader_set_tile(tile);
    lds_barrier();                                     // the row lists of the first tile are in LDS
    int sel = 0;                                       // which half of the row lists the CURRENT tile uses
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) issue_piece(q, 0, 0, sel);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int stage = 0, par = 0;
#ifdef LIME_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (; tile >= 0; ti += nw_x, tile = tile_at(ti), par ^= 1, sel ^= 1) {
        const bool more = ti + nw_x < tcount;
        acc_init(tile, par, sel);
        load_rows(tile_at(ti + nw_x));                 // the next tile's row lists: loaded now, parked after the first chunk
        SSTAMP(0)                                     // 0: accumulator init (residual loads issued)
        for (int c = 0; c + 1 < nchunk; ++c) {
            compute(stage, stage ^ 1, c + 1, sel);
            __builtin_amdgcn_sched_barrier(0);         // MFMAs touch no memory: hipcc otherwise sinks them below the wait + barrier
            SSTAMP(c == 0 ? 1 : 2)                    // 1: first chunk of a tile (waits for the residual loads), 2: reads + split + MFMA + DMA issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SSTAMP(c == 0 ? 3 : 4)                    // 3: DMA (and the previous tile's stores) landed, first chunk; 4: other chunks
            if (c == 0) park_rows(sel ^ 1);
            lds_barrier();
            SSTAMP(5)                                 // 5: barrier
            stage ^= 1;
        }
        // last chunk of the tile: the loader moves on to the next tile first (its row lists were parked >= one barrier ago:
        // the dispatcher guarantees nchunk >= 2)
        if (more) loader_set_tile(tbase + ti + nw_x);
        compute(stage, more ? (stage ^ 1) : -1, 0, sel ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        SSTAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SSTAMP(4)
        lds_barrier();
        SSTAMP(5)
        stage ^= 1;
        // the stores retire under the next tile's first chunk; the bias image is double-buffered by tile parity
        epilogue(tile, par, sel);
        SSTAMP(6)                                     // 6: epilogue (stores issued)
    }
#ifdef LIME_STAMPS
    if (p.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.stamps[((long)blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
    }
#endif
}

int sp_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int CT, bool LN, bool RELU, int RES, bool POOL = false, bool RSTD = false, bool CID = false>
int launch(const PPParams& p0, hipStream_t stream) {
    PPParams p = p0;
    p.n_row_blocks = (p.M + BM - 1) / BM;
    p.n_col_blocks = (p.N + 32 * CT - 1) / (32 * CT);
    const long ntiles = (long)p.n_row_blocks * p.n_col_blocks;
    long nwg = sp_num_cus();
    if (nwg > ntiles) nwg = ntiles;
#ifdef LIME_STAMPS
    p.stamps = g_sp_stamp_buf;
#endif
    hipLaunchKernelGGL((gemm_sp_kernel<CT, LN, RELU, RES, POOL, RSTD, CID>), dim3((unsigned)nwg), dim3(512), 0, stream, p);
    lime_set_last_linear_kernel("gemm_sp_kernel<%d, %s, %s, %d, %s, %s, %s>", CT, LN ? "true" : "false", RELU ? "true" : "false", RES,
                                POOL ? "true" : "false", RSTD ? "true" : "false", CID ? "true" : "false");    // as rocprofv3 prints it
    return lime_check_launch("lime_linear_f32");
}

inline bool al16(const void* ptr, long ld) { return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld % 4) == 0); }

int g_split_mode = -1;         // -1: not read yet; 0 off; 1 on

}  // namespace

// 1: lime_linear_f32 routes the big-M GEMMs through the split-product kernel (default); 0: through the fp32-MFMA kernels
// (gemm_pp_f32.hip).  Process-wide; returns the previous setting.  LIME_SPLIT_GEMM=0 in the environment sets the start value.
extern "C" int lime_set_split_gemm(int on) {
    if (g_split_mode < 0) {
        const char* e = getenv("LIME_SPLIT_GEMM");
        g_split_mode = (e && e[0] == '0') ? 0 : 1;
    }
    const int prev = g_split_mode;
    if (on >= 0 && on <= 7) g_split_mode = on;       // bit 1: also the gathered-residual LayerNorm GEMM, bit 2: ignore the fill and row-count rules (tests / A-B runs)
    return prev;
}

int lime_split_mode() {
    if (g_split_mode < 0) lime_set_split_gemm(-1);
    return g_split_mode;
}

// LIME_OK / error: launched (or failed); LIME_PP_NOT_APPLICABLE: the caller takes the fp32-MFMA kernels.
int lime_linear_sp(const lime_linear_args* a, hipStream_t s) {
    if (g_split_mode < 0) lime_set_split_gemm(-1);
    if (!(g_split_mode & 1)) return LIME_PP_NOT_APPLICABLE;
    const bool has_res = a->res != nullptr, ln = a->ln_gamma != nullptr;
    const bool relu = a->act == LIME_ACT_RELU;
    if (a->a_pe != nullptr) return LIME_PP_NOT_APPLICABLE;
    const bool act_rt = a->act == LIME_ACT_TANH || a->act == LIME_ACT_SIGMOID;       // applied at run time in the epilogue
    const bool relu_grad = a->act == LIME_ACT_RELU_GRAD;
    if (a->dropout_p > 0.f && !(relu && !has_res && !ln && !a->pool32 && !a->c_ids)) return LIME_PP_NOT_APPLICABLE;
    if (relu_grad && (!has_res || ln || a->res_ids || a->res_mod > 0 || a->res_div > 1 || a->c_ids || a->pool32)) return LIME_PP_NOT_APPLICABLE;
    if (!(a->act == LIME_ACT_NONE || relu_grad || (relu && !has_res) || (act_rt && !ln && (!has_res || a->res_ids)))) return LIME_PP_NOT_APPLICABLE;
    if (a->K % 4 || a->N % 4 || a->K < 64) return LIME_PP_NOT_APPLICABLE;          // >= 2 chunks (row lists, bias image)
    if (!al16(a->a, a->lda) || !al16(a->w, a->ldw) || !al16(a->c, a->ldc) || !al16(a->res, a->ldr) || !al16(a->res_pe, a->ldr_pe))
        return LIME_PP_NOT_APPLICABLE;
    if (a->bias && (uintptr_t)a->bias % 4) return LIME_PP_NOT_APPLICABLE;
    const long row_blocks = ((long)a->M + BM - 1) / BM;
    if (a->c_ids && !(has_res && !a->res_ids && a->res_mod > 0 && !ln && a->act == LIME_ACT_NONE)) return LIME_PP_NOT_APPLICABLE;
    const long lim = 0x7FFFFFF0L;
    if (a->c_ids && (long)a->M * a->ldc * 4 >= lim) return LIME_PP_NOT_APPLICABLE;
    if (256L * a->lda * 4 >= lim || (long)a->N * a->ldw * 4 >= lim || 256L * a->ldc * 4 >= lim || 256L * a->ldr * 4 >= lim ||
        (long)a->M * 4 >= lim)
        return LIME_PP_NOT_APPLICABLE;
    int res = 0;
    if (relu_grad) {
        if ((long)a->M * a->ldr * 4 >= lim) return LIME_PP_NOT_APPLICABLE;
        res = 3;
    } else if (has_res) {
        if (a->res_ids) res = 2;
        else res = 1;                                  // dense, periodic (res_mod) or broadcast (res_div > 1) rows
        if (res == 1 && a->res_mod > 0 && (long)a->res_mod * a->ldr * 4 >= lim) return LIME_PP_NOT_APPLICABLE;
        if (res == 1 && a->res_div > 1 && (a->res_mod > 0 || ln || ((long)a->M / a->res_div + 1) * a->ldr * 4 >= lim)) return LIME_PP_NOT_APPLICABLE;
    }
    // The epilogues the layers AROUND the encoders need (tanh / sigmoid: Attention.affine1, gates; a gathered residual without
    // LayerNorm: LIME.project over the freshness table; a broadcast residual: SAGEConv lin_r) exist here for the LARGE batches
    // (BASELINE configs[2] and [4]: 14k .. 150k rows, where the 64 x 64-tile mid-M kernel runs at a fifth of this kernel's rate);
    // below ~12k rows one round of 256-row tiles is slower than the mid-M kernel's launch.
    const bool extended = act_rt || (res == 2 && !ln) || (res == 1 && a->res_div > 1);
    if (extended && a->M < 12288 && !(g_split_mode & 4)) return LIME_PP_NOT_APPLICABLE;
    if (ln && (a->N > 320 || relu)) return LIME_PP_NOT_APPLICABLE;
    // out_proj (gathered residual + positional rows + LayerNorm): this kernel's instantiation adds the residual in its epilogue and
    // keeps 265 registers in scratch there; it measures 67-80 TFLOP/s against 100 of gemm_pp_f32.hip (with the residual loaded into
    // the accumulators at the tile start: 186 registers, 73 TFLOP/s) -- left to the fp32 kernel; lime_set_split_gemm(3) routes it here
    if (res == 2 && ln && !(g_split_mode & 2)) return LIME_PP_NOT_APPLICABLE;      // only with lime_set_split_gemm(3)
    if (a->pool32 && !(ln && has_res && !a->res_ids && a->res_div <= 1 && a->M % 32 == 0)) return LIME_PP_NOT_APPLICABLE;
    // the tile width (256 / 320) that pads N least
    const int pad5 = (a->N + 319) / 320 * 320 - a->N, pad4 = (a->N + 255) / 256 * 256 - a->N;
    const bool wide = ln || pad5 <= pad4;
    const long ntiles = row_blocks * ((a->N + (wide ? 319 : 255)) / (wide ? 320 : 256));
    // From the same M on as gemm_pp_f32.hip takes over from the mid-M kernel ...
    if (a->M < 4096) return LIME_PP_NOT_APPLICABLE;
    // ... and only where 256-row tiles on one workgroup per CU fill the chip: a launch is `rounds` passes of ncu tiles, and a tile
    // block that hangs over N computes dead columns.  Below ~0.45 of (tiles / (rounds x ncu)) x (N / covered columns) the 128-row /
    // 64-row tile kernels win: one round here takes 70 us at K = 400 however few tiles it has (tools/exp/sp_fill.py: M = 14k, N = 400 -- fill
    // 0.34 -- 70 us here, 53 there; M = 21k -- 0.50 -- 71 vs 85; tanh, N = 200: M = 28k -- 0.33 -- 70 vs 68, M = 42k -- 0.50 -- 75 vs 107).
    // A device-side row count (m_dev) hides the real M: those launches (the compacted encoder layers) always come here.  A caller
    // that cuts its rows into EQUAL passes (Model.score_impressions) gets one kernel family -- one rounding -- for every pass.
    if (!a->m_dev && !(g_split_mode & 4)) {
        const long ncu = sp_num_cus();
        const long rounds = (ntiles + ncu - 1) / ncu;
        const double fill = (double)ntiles / (double)(rounds * ncu) * (double)a->N / (double)(((a->N + (wide ? 319 : 255)) / (wide ? 320 : 256)) * (wide ? 320 : 256));
        if (fill < 0.45) return LIME_PP_NOT_APPLICABLE;
    }

    PPParams p;
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids;
    p.w = a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = a->res; p.ldr = a->ldr; p.res_mod = a->res_mod; p.res_ids = a->res_ids;
    p.res_pe = a->res_pe; p.ldr_pe = a->ldr_pe; p.res_period = a->res_period > 0 ? a->res_period : 1;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_rstd = a->ln_rstd;
    p.c = a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.ln_count = a->N;
    p.n_row_blocks = p.n_col_blocks = 0;
    p.m_dev = a->m_dev; p.c_ids = a->c_ids;
    p.act = act_rt ? a->act : 0;
    p.res_div = (res == 1 && a->res_div > 1) ? a->res_div : 1;
    p.act_scale = a->act_scale;
    if (a->dropout_p > 0.f) p.drop = lime_make_dropout(a->dropout_p, a->dropout_seed, a->dropout_site);
    {   // diagnostic: LIME_SP_MASK disables classes of instantiations (bit 0 c_ids, 1 LayerNorm + rstd, 2 LayerNorm, 3 residual,
        // 4 ReLU, 5 plain; bit 6: the 256-column tiles)
        static const int mask = getenv("LIME_SP_MASK") ? atoi(getenv("LIME_SP_MASK")) : 0;
        const int cls = a->c_ids ? 0 : (ln && a->ln_rstd) ? 1 : ln ? 2 : res ? 3 : relu ? 4 : 5;
        if (mask & (1 << cls)) return LIME_PP_NOT_APPLICABLE;
        if (!wide && (mask & 64)) return LIME_PP_NOT_APPLICABLE;
    }
    if (a->c_ids) return wide ? launch<10, false, false, 1, false, false, true>(p, s) : launch<8, false, false, 1, false, false, true>(p, s);
    if (ln) {
        if (a->ln_rstd) {                              // training forward: residual + LayerNorm, rstd kept
            if (res == 0 || a->pool32) return LIME_PP_NOT_APPLICABLE;
            return res == 1 ? launch<10, true, false, 1, false, true>(p, s) : launch<10, true, false, 2, false, true>(p, s);
        }
        if (res == 0) return LIME_PP_NOT_APPLICABLE;
        if (res == 1) return a->pool32 ? launch<10, true, false, 1, true>(p, s) : launch<10, true, false, 1>(p, s);
        return launch<10, true, false, 2>(p, s);
    }
    if (res == 3) {                                    // (the 320-column instantiation puts 33 registers in scratch: not built; linear1 is 512 wide)
        if (wide) return LIME_PP_NOT_APPLICABLE;
        return launch<8, false, false, 3>(p, s);
    }
    if (res == 2) return wide ? launch<10, false, false, 2>(p, s) : launch<8, false, false, 2>(p, s);
    if (wide) {
        if (res == 1) return launch<10, false, false, 1>(p, s);
        return relu ? launch<10, false, true, 0>(p, s) : launch<10, false, false, 0>(p, s);
    }
    if (res == 1) return launch<8, false, false, 1>(p, s);
    return relu ? launch<8, false, true, 0>(p, s) : launch<8, false, false, 0>(p, s);
}
