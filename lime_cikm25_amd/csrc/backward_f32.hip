// Backward kernels of the training step (reference trainer.py:131-148 drives
// loss.backward() through the encoder layers of newsEncoders.py:244-247,311-321).
//
//   wgrad_kernel           dW[n, k] = sum_m dY[m, n] X[m, k]      split over M, exact-fp32 MFMA, partials + fixed-order reduce
//   colsum_kernel          db[n] = sum_m dY[m, n]
//   layernorm_bwd_kernel   dZ of y = LayerNorm(z) from (dY, y, rstd) + the column sums for d gamma / d beta / bias
//   relu_bwd_kernel        dH *= (h > 0)
//   token_attn_bwd_kernel  dQ / dK / dV of softmax(scale Q K^T) V per (sequence, head), probabilities recomputed
//   embed_bwd_kernel       dTable[ids[r]] += dX[r]                (float atomics; embed_bwd_sorted.hip is the fixed-order replacement)
//   sumsq / clip / adam    clip_grad_norm_ + Adam over flat buffers, nll_softmax: the loss of trainer.py:71-73
//
// All dense reductions are fixed-order (partials in a caller workspace, then one summing pass).
#include "common.h"
#include "dropout.h"
#include "gemm_pp.h"
#include "split_mfma.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// buffer addressing (as in gemm_f32.hip): wave-uniform base + 32-bit lane offset, an out-of-range offset reads as zero
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ f32x4v buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    // v_mfma_f32_16x16x4_f32: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; c[r] = C[4 (l >> 4) + r][l & 15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float block_sum_4(float v, float* red) {        // 256 threads, fixed order
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// reduce_partials: out[r, c] (+)= sum_s ws[s * split_stride + r * ldw + c].  A workgroup owns 64 consecutive outputs; its
// four waves take the splits s = wave, wave + 4, ... and the four sums are added in a fixed order.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, long split_stride, int splits,
                                                               long ldw, float* __restrict__ out, long ldo, int rows, int cols,
                                                               int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    const bool ok = e < (long)rows * cols;
    const int r = ok ? (int)(e / cols) : 0, c = ok ? (int)(e - (long)r * cols) : 0;
    float s = 0.f;
    if (ok) {
        const float* p = ws + (long)r * ldw + c;
        for (int i = g; i < splits; i += 4) s += p[(long)i * split_stride];
    }
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && ok) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float* o = out + (long)r * ldo + c;
        *o = accumulate ? *o + t : t;
    }
}

// The same with four consecutive outputs per lane (16-byte accesses; cols % 4 == 0 so a group never leaves its row) and four
// splits of a wave in flight: the scalar version reads 256 B per wave and load, 0.7 TB/s on the 39 MB of in_proj's 32 partial
// tiles.  Same association as above (a wave sums its splits in order, the four waves' sums are added pairwise): same bits.
// `extra` (optional): column `cols` of the partial grid also holds a sum -- the ones column's bias gradient -- and goes to extra[r] in the
// same launch (a second launch per weight gradient for N floats was 19 launches of 6 us per training step).
__global__ __launch_bounds__(256) void reduce_partials_vec4_kernel(const float* __restrict__ ws, long split_stride, int splits,
                                                                    long ldw, float* __restrict__ out, long ldo, int rows, int cols,
                                                                    int accumulate, float* __restrict__ extra) {
    __shared__ f32x4v red4[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c4 = (cols >> 2) + (extra != nullptr ? 1 : 0);
    const long e = (long)blockIdx.x * 64 + lane;                 // group of four columns
    const bool ok = e < (long)rows * c4;
    const int r = ok ? (int)(e / c4) : 0, c = ok ? (int)(e - (long)r * c4) * 4 : 0;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* p = ws + (long)r * ldw + c;
        int i = g;
        for (; i + 12 < splits; i += 16) {
            f32x4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4v*>(p + (long)(i + 4 * u) * split_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; i < splits; i += 4) s += *reinterpret_cast<const f32x4v*>(p + (long)i * split_stride);
    }
    red4[g][lane] = s;
    __syncthreads();
    if (g == 0 && ok) {
        const f32x4v t = (red4[0][lane] + red4[1][lane]) + (red4[2][lane] + red4[3][lane]);
        if (c == cols) {                                   // the extra column (only with `extra`)
            extra[r] = accumulate ? extra[r] + t[0] : t[0];
        } else {
            f32x4v* o = reinterpret_cast<f32x4v*>(out + (long)r * ldo + c);
            *o = accumulate ? *o + t : t;
        }
    }
}

// The three column sums of layernorm_bwd (dgamma, dbeta, dzsum) out of its per-workgroup partials ws[blk][3][E] in ONE launch:
// a workgroup owns 16 groups of four consecutive floats of the 3 E, its 16 split lanes take the blocks s, s + 16, ... (up to 768
// blocks: 48 independent 16-byte loads per thread instead of 192 dependent-issue ones in two workgroups, 18 us per vector) and their
// sums meet in LDS in a fixed order.  E % 4 == 0.
__global__ __launch_bounds__(256) void reduce_ln3_kernel(const float* __restrict__ ws, int nblk, int E, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2, int accumulate) {
    __shared__ f32x4v red[16][16];
    const int cg = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int g = blockIdx.x * 16 + cg, ng = 3 * E / 4;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
    if (g < ng) {
        const float* p = ws + 4L * g;
        int i = sl;
        for (; i + 48 < nblk; i += 64) {
            f32x4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4v*>(p + (long)(i + 16 * u) * 3 * E);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; i < nblk; i += 16) s += *reinterpret_cast<const f32x4v*>(p + (long)i * 3 * E);
    }
    red[sl][cg] = s;
    __syncthreads();
    if (sl == 0 && g < ng) {
        f32x4v t = red[0][cg];
#pragma unroll
        for (int u = 1; u < 16; ++u) t += red[u][cg];
        const int c = 4 * g, k = c / E, cc = c - k * E;          // E % 4 == 0: a group never leaves its vector
        float* const o = k == 0 ? o0 : (k == 1 ? o1 : o2);
        if (o) {
            f32x4v* q = reinterpret_cast<f32x4v*>(o + cc);
            *q = accumulate ? *q + t : t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: workgroup = one 128 x TK tile of dW (TK = 64 NKT: the whole K of the encoder layers' 300-wide operands) over one
// slice of the M rows.  Eight waves in a 2 x 4 grid, each a 64 x 16 NKT patch (4 x NKT accumulator tiles of 16 x 16), two
// waves per SIMD so one wave's LDS waits sit under the other's MFMAs.  64-row chunks of dY and X go global -> registers
// -> LDS (one stage): the loads of chunk i + 1 are issued right after chunk i has been written to LDS and are in flight
// for the whole of its MFMA phase (56 KB per workgroup -- one 32-row chunk in flight left the kernel latency-bound).
// ---------------------------------------------------------------------------------------------------
constexpr int WG_TN = 128;
constexpr int WG_MC = 64;
constexpr int WG_THREADS = 512;

template <int NKT, bool VEC>       // NKT: 16-column accumulator tiles per wave along K (TK = 64 NKT); VEC: 16-byte loads
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(const float* __restrict__ dy, long ldy, const float* __restrict__ x,
                                                            long ldx, float* __restrict__ ws, int M, int N, int K, int n_tiles,
                                                            int k_tiles, int rows_per_split, int ones_col) {
    constexpr int TK = 64 * NKT;
    constexpr int LDA = WG_TN + 16;          // pitch % 32 == 16: the two row groups of a half-wave hit disjoint banks
    constexpr int LDB = TK + 16;
    constexpr int A4 = WG_MC * WG_TN / 4 / WG_THREADS;                    // float4 per thread per chunk (dY tile): 2
    constexpr int B4 = (WG_MC * TK / 4 + WG_THREADS - 1) / WG_THREADS;    // (X tile): 3, 4 or 5
    extern __shared__ float wg_smem[];
    float* const As = wg_smem;                                            // [WG_MC * LDA]
    float* const Bs = wg_smem + WG_MC * LDA;                              // [WG_MC * LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, kg = lane >> 4;
    const int ntile = n_tiles * k_tiles;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile = logical - split * ntile;
    const int n0 = (tile / k_tiles) * WG_TN, k0 = (tile % k_tiles) * TK;
    const long m_begin = (long)split * rows_per_split;
    const long m_end = min((long)M, m_begin + rows_per_split);
    const int wn = (wave >> 2) * 64, wk = (wave & 3) * (16 * NKT);

    f32x4 acc[4][NKT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Loads are branch-free buffer loads: rows beyond the slice and columns beyond N / K carry the OOB offset and read zeros.
    // Offsets are relative to the first row of the slice (the host checks that a slice spans < 2 GB).
    f32x4v ra[A4], rb[B4];
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(dy + m_begin * ldy + n0);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(x + m_begin * ldx + k0);
    const int rows_here = (int)(m_end - m_begin);
    unsigned a_off[A4], b_off[B4];
    int a_row[A4], b_row[B4];
    bool a_cin[A4][VEC ? 1 : 4], b_cin[B4][VEC ? 1 : 4];
    int b_one[B4];                                          // which element of this lane's X float4 is the ones column (-1: none)
#pragma unroll
    for (int j = 0; j < A4; ++j) {
        const int f = tid + WG_THREADS * j, r = f / (WG_TN / 4), c = (f % (WG_TN / 4)) * 4;
        a_row[j] = r;
        a_off[j] = (unsigned)r * (unsigned)(ldy * 4) + (unsigned)c * 4u;
        if constexpr (VEC) a_cin[j][0] = n0 + c < N;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) a_cin[j][e] = n0 + c + e < N;
    }
#pragma unroll
    for (int j = 0; j < B4; ++j) {
        const int f = tid + WG_THREADS * j, r = f / (TK / 4), c = (f % (TK / 4)) * 4;
        const bool in_tile = f < WG_MC * TK / 4;
        b_row[j] = in_tile ? r : (1 << 30);
        b_off[j] = (unsigned)r * (unsigned)(ldx * 4) + (unsigned)c * 4u;
        b_one[j] = (ones_col && K >= k0 + c && K < k0 + c + 4) ? K - (k0 + c) : -1;
        if constexpr (VEC) b_cin[j][0] = k0 + c < K;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) b_cin[j][e] = k0 + c + e < K;
    }
    auto load_chunk = [&](int mrel) {                       // mrel: first row of the chunk relative to the slice
        const int soff_a = mrel * (int)(ldy * 4), soff_b = mrel * (int)(ldx * 4);
#pragma unroll
        for (int j = 0; j < A4; ++j) {
            const bool rin = mrel + a_row[j] < rows_here;
            if constexpr (VEC) ra[j] = buf_load4(rs_a, rin && a_cin[j][0] ? a_off[j] : OOB, soff_a);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[j][e] = buf_load1(rs_a, rin && a_cin[j][e] ? a_off[j] + 4u * e : OOB, soff_a);
        }
#pragma unroll
        for (int j = 0; j < B4; ++j) {
            const bool rin = mrel + b_row[j] < rows_here;
            if constexpr (VEC) rb[j] = buf_load4(rs_b, rin && b_cin[j][0] ? b_off[j] : OOB, soff_b);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[j][e] = buf_load1(rs_b, rin && b_cin[j][e] ? b_off[j] + 4u * e : OOB, soff_b);
            // column K of X reads as 1 on the valid rows: column K of dW then holds sum_m dY[m, n], the bias gradient
            if (b_one[j] >= 0 && rin) rb[j][b_one[j]] = 1.0f;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < A4; ++j) {
            const int f = tid + WG_THREADS * j, r = f / (WG_TN / 4), c = (f % (WG_TN / 4)) * 4;
            *reinterpret_cast<f32x4v*>(&As[r * LDA + c]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B4; ++j) {
            const int f = tid + WG_THREADS * j, r = f / (TK / 4), c = (f % (TK / 4)) * 4;
            if (f < WG_MC * TK / 4) *reinterpret_cast<f32x4v*>(&Bs[r * LDB + c]) = rb[j];
        }
    };

    if (rows_here > 0) {
        load_chunk(0);
        const float* as = &As[kg * LDA + wn + fi];
        const float* bs = &Bs[kg * LDB + wk + fi];
        for (int m0 = 0; m0 < rows_here; m0 += WG_MC) {
            __syncthreads();                               // every wave is done reading the previous chunk
            store_chunk();
            __syncthreads();
            if (m0 + WG_MC < rows_here) load_chunk(m0 + WG_MC);
#pragma unroll
            for (int s = 0; s < WG_MC / 4; ++s) {
                float a[4], b[NKT];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = as[4 * s * LDA + 16 * i];
#pragma unroll
                for (int j = 0; j < NKT; ++j) b[j] = bs[4 * s * LDB + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NKT; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            }
        }
    }
    // partial tile -> ws[split][n][k] over the padded [n_tiles * 128, k_tiles * TK] grid
    const long ldw = (long)k_tiles * TK;
    float* o = ws + (long)split * ((long)n_tiles * WG_TN) * ldw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(long)(n0 + wn + 16 * i + 4 * kg + r) * ldw + k0 + wk + 16 * j + fi] = acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------------
// wgrad with LDS-DMA staging (16-byte aligned operands): the same tiling as wgrad_kernel, but 32-row chunks go global -> LDS
// directly (raw_ptr_buffer_load_lds, 1 KB per wave instruction, rows packed without padding), two stages, ONE barrier per
// chunk: chunk c + 1 is in flight while chunk c is multiplied; no staging registers, no LDS stores.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds_base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)          // (inside a kernel template the builtin makes the host pass drop the launch stub)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_base, 16, voff, soff, 0, 0);
#endif
}

constexpr int WD_MC = 32;          // rows per chunk: two stages of 32 rows = 112 KB at TK = 320 (16-row chunks with two workgroups
                                   // per CU measured the same kernel time and double the partial tiles to sum)

template <int NKT>
__global__ __launch_bounds__(WG_THREADS) void wgrad_dma_kernel(const float* __restrict__ dy, long ldy, const float* __restrict__ x,
                                                                long ldx, float* __restrict__ ws, int M, int N, int K, int n_tiles,
                                                                int k_tiles, int rows_per_split, int ones_col) {
    constexpr int TK = 64 * NKT;
    constexpr int A_FLOATS = WD_MC * WG_TN, B_FLOATS = WD_MC * TK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_P = A_FLOATS / 256, B_P = B_FLOATS / 256;     // 1 KB DMA pieces per chunk: 8 and 12 / 16 / 20
    constexpr int A_PW = (A_P + 7) / 8, B_PW = (B_P + 7) / 8;     // per wave (piece q = wave + 8 j, q < P)
    extern __shared__ float wg_smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, kg = lane >> 4;
    const int ntile = n_tiles * k_tiles;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile = logical - split * ntile;
    const int n0 = (tile / k_tiles) * WG_TN, k0 = (tile % k_tiles) * TK;
    const long m_begin = (long)split * rows_per_split;
    const int rows_here = (int)(min((long)M, m_begin + rows_per_split) - m_begin);
    const int wn = (wave >> 2) * 64, wk = (wave & 3) * (16 * NKT);

    f32x4 acc[4][NKT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rows_here > 0) {
        const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(dy + m_begin * ldy + n0);
        const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(x + m_begin * ldx + k0);
        // this lane's 16-byte pieces: piece q of the wave covers floats [256 q + 4 lane, + 4) of the packed [32][cols] image
        unsigned a_off[A_PW], b_off[B_PW];
        int a_row[A_PW], b_row[B_PW], b_one[B_PW], b_r[B_PW];
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int q = wave + 8 * j, f = q * 256 + 4 * lane, r = f / WG_TN, c = f % WG_TN;
            a_row[j] = (q < A_P && n0 + c < N) ? r : (1 << 30);
            a_off[j] = (unsigned)r * (unsigned)(ldy * 4) + (unsigned)c * 4u;
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const int q = wave + 8 * j, f = q * 256 + 4 * lane, r = f / TK, c = f % TK;
            b_row[j] = (q < B_P && k0 + c < K) ? r : (1 << 30);
            b_off[j] = (unsigned)r * (unsigned)(ldx * 4) + (unsigned)c * 4u;
            b_r[j] = r;
            // the all-ones column of X (bias gradient) falls into this lane's piece: the lane patches it after its DMA landed
            b_one[j] = (q < B_P && ones_col && K >= k0 + c && K < k0 + c + 4) ? K - k0 - c : -1;
        }
        auto issue = [&](int stage, int mrel) {
            float* const sb = wg_smem + stage * STAGE;
            const int soff_a = mrel * (int)(ldy * 4), soff_b = mrel * (int)(ldx * 4);
#pragma unroll
            for (int j = 0; j < A_PW; ++j)
                if (wave + 8 * j < A_P)
                    dma16(rs_a, sb + (wave + 8 * j) * 256, (mrel + a_row[j] < rows_here) ? a_off[j] : OOB, soff_a);
#pragma unroll
            for (int j = 0; j < B_PW; ++j)
                if (wave + 8 * j < B_P)
                    dma16(rs_b, sb + A_FLOATS + (wave + 8 * j) * 256, (mrel + b_row[j] < rows_here) ? b_off[j] : OOB, soff_b);
        };
        issue(0, 0);
        int stage = 0;
        for (int m0 = 0; m0 < rows_here; m0 += WD_MC) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of chunk m0 have landed
#pragma unroll
            for (int j = 0; j < B_PW; ++j)
                if (b_one[j] >= 0)
                    wg_smem[stage * STAGE + A_FLOATS + (wave + 8 * j) * 256 + 4 * lane + b_one[j]] = (m0 + b_r[j] < rows_here) ? 1.0f : 0.f;
            __syncthreads();                                           // everybody's have; the other stage is no longer read
            if (m0 + WD_MC < rows_here) issue(stage ^ 1, m0 + WD_MC);
            const float* as = wg_smem + stage * STAGE + kg * WG_TN + wn + fi;
            const float* bs = wg_smem + stage * STAGE + A_FLOATS + kg * TK + wk + fi;
#pragma unroll
            for (int s = 0; s < WD_MC / 4; ++s) {
                float a[4], b[NKT];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = as[4 * s * WG_TN + 16 * i];
#pragma unroll
                for (int j = 0; j < NKT; ++j) b[j] = bs[4 * s * TK + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NKT; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            }
            stage ^= 1;
        }
    }
    const long ldw = (long)k_tiles * TK;
    float* o = ws + (long)split * ((long)n_tiles * WG_TN) * ldw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(long)(n0 + wn + 16 * i + 4 * kg + r) * ldw + k0 + wk + 16 * j + fi] = acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------------
// colsum: partial[blk][c] = sum of x[r, c] over the block's rows
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ldx, int M, int N, int rows_per_block,
                                                      float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min((long)M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (long r = r0 + g; r < r1; r += 4) s += x[r * ldx + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < N) ws[(long)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = gamma * xhat + beta with xhat = (z - mean) * rstd; given dY, y and rstd:
//   g = dY * gamma;  dZ = rstd * (g - mean(g) - xhat * mean(g * xhat)),   xhat = (y - beta) / gamma
// dY row of output row r is dy[(r / dy_div)] * dy_scale (the mean-pool backward of newsEncoders.py:317,321 broadcasts one
// pooled-gradient row over the S tokens with 1 / S).  One wave per row, CPL columns per lane; partial column sums
// (d gamma, d beta, sum dZ) per workgroup in ws[blk][3][E].
// ---------------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, long lddy, int dy_div, float dy_scale,
                                                             const float* __restrict__ y, long ldy, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ rstd,
                                                             float* __restrict__ dz, long lddz, int M, int E,
                                                             float* __restrict__ ws) {
    __shared__ float red[4][3][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ga[CPL], be[CPL], inv_ga[CPL], sg[CPL], sb[CPL], sz[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = lane + 64 * j;
        ga[j] = c < E ? gamma[c] : 0.f;
        be[j] = c < E ? beta[c] : 0.f;
        inv_ga[j] = c < E ? 1.0f / ga[j] : 0.f;
        sg[j] = sb[j] = sz[j] = 0.f;
    }
    const float inv_e = 1.0f / (float)E;
    for (long r = (long)blockIdx.x * 4 + wave; r < M; r += (long)gridDim.x * 4) {
        const float* pdy = dy + (r / dy_div) * lddy;
        const float* py = y + r * ldy;
        float d[CPL], xh[CPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            d[j] = c < E ? pdy[c] * dy_scale : 0.f;
            xh[j] = c < E ? (py[c] - be[j]) * inv_ga[j] : 0.f;
            const float g = d[j] * ga[j];
            s1 += g;
            s2 += g * xh[j];
        }
        s1 = wave_sum(s1) * inv_e;
        s2 = wave_sum(s2) * inv_e;
        const float rs = rstd[r];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            const float v = rs * (d[j] * ga[j] - s1 - xh[j] * s2);
            if (c < E) dz[r * lddz + c] = v;
            sg[j] += d[j] * xh[j];
            sb[j] += d[j];
            sz[j] += c < E ? v : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        red[wave][0][lane + 64 * j] = sg[j];
        red[wave][1][lane + 64 * j] = sb[j];
        red[wave][2][lane + 64 * j] = sz[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * E; e += 256) {
        const int k = e / E, c = e - k * E;
        ws[((long)blockIdx.x * 3 + k) * E + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// The same with 16 lanes per row (four rows per wave at a time) and 16-byte accesses: E % 4 == 0, 16-byte aligned rows.
template <int V4>        // float4 per lane: ceil(E / 64)
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, long lddy, int dy_div, float dy_scale,
                                                                 const float* __restrict__ y, long ldy,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ rstd, float* __restrict__ dz, long lddz,
                                                                 int M, int E, float* __restrict__ ws,
                                                                 float* __restrict__ dz_drop, long lddd, LimeDropout drop) {
    __shared__ float red[4][3][64 * V4];
    __shared__ __attribute__((aligned(16))) float Gs[64 * V4], Bs[64 * V4], IGs[64 * V4];   // gamma, beta, 1 / gamma (registers are
                                                                                          // for the column sums: occupancy)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
    for (int c = threadIdx.x; c < 64 * V4; c += 256) {
        const float g = c < E ? gamma[c] : 0.f;
        Gs[c] = g;
        Bs[c] = c < E ? beta[c] : 0.f;
        IGs[c] = c < E ? 1.0f / g : 0.f;
    }
    __syncthreads();
    f32x4v sg[V4], sb[V4], sz[V4];
#pragma unroll
    for (int j = 0; j < V4; ++j) sg[j] = sb[j] = sz[j] = zero;
    const float inv_e = 1.0f / (float)E;
    for (long r4 = ((long)blockIdx.x * 4 + wave) * 4; r4 < M; r4 += (long)gridDim.x * 16) {
        const long r = r4 + rg;
        const bool rin = r < M;
        const float* pdy = dy + ((rin ? r : 0) / dy_div) * lddy;
        const float* py = y + (rin ? r : 0) * ldy;
        f32x4v d[V4], xh[V4];
        float s1 = 0.f, s2 = 0.f;
        // the three vectors are re-read from LDS for every row: hidden from the optimiser, which would otherwise hoist the
        // loop-invariant loads back into 60 registers
        const float *gs = Gs, *bs = Bs, *igs = IGs;
        asm volatile("" : "+v"(gs), "+v"(bs), "+v"(igs));
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            const bool ok = rin && c < E;
            const f32x4v ga = *reinterpret_cast<const f32x4v*>(&gs[c]), be = *reinterpret_cast<const f32x4v*>(&bs[c]);
            d[j] = ok ? *reinterpret_cast<const f32x4v*>(pdy + c) * dy_scale : zero;
            const f32x4v yv = ok ? *reinterpret_cast<const f32x4v*>(py + c) : be;
            xh[j] = (yv - be) * *reinterpret_cast<const f32x4v*>(&igs[c]);
            const f32x4v g = d[j] * ga;
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1 += g[e]; s2 += g[e] * xh[j][e]; }
        }
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
        s1 += __shfl_xor(s1, 4); s2 += __shfl_xor(s2, 4);
        s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8);
        s1 *= inv_e; s2 *= inv_e;
        const float rs = rin ? rstd[r] : 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            const bool ok = rin && c < E;
            f32x4v v = (d[j] * *reinterpret_cast<const f32x4v*>(&gs[c]) - s1 - xh[j] * s2) * rs;
            if (!ok) v = zero;
            if (ok) *reinterpret_cast<f32x4v*>(dz + r * lddz + c) = v;
            f32x4v t = v;
            if (dz_drop != nullptr && ok) {            // the gradient through the dropout in front of the residual add, written alongside
                const unsigned keep = lime_keep4(drop, ((uint64_t)r * (uint64_t)E + (uint64_t)c) >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (keep >> e) & 1u ? v[e] * drop.scale : 0.f;
                *reinterpret_cast<f32x4v*>(dz_drop + r * lddd + c) = t;
            }
            sg[j] += d[j] * xh[j];
            sb[j] += d[j];
            sz[j] += t;                                // column sums of what goes on to the linear in front: its bias gradient
        }
    }
    // the four row groups of the wave, then the four waves
#pragma unroll
    for (int j = 0; j < V4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = sg[j][e], b = sb[j][e], c = sz[j][e];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); c += __shfl_xor(c, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); c += __shfl_xor(c, 32);
            if (rg == 0) {
                const int col = 4 * (sub + 16 * j) + e;
                red[wave][0][col] = a; red[wave][1][col] = b; red[wave][2][col] = c;
            }
        }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * E; e += 256) {
        const int k = e / E, c = e - k * E;
        ws[((long)blockIdx.x * 3 + k) * E + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dh, long lddh, const float* __restrict__ h, long ldh,
                                                        long rows, int cols, float scale) {
    const long total = rows * cols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / cols;
        const int c = (int)(e - r * cols);
        const float g = dh[r * lddh + c];
        dh[r * lddh + c] = h[r * ldh + c] > 0.f ? g * scale : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------
// token attention backward (unmasked encoder-layer attention, newsEncoders.py:316,320).
// A problem = one (sequence, head); a wave owns 16 query rows (and, for dK / dV, the 16 key rows of the same numbers);
// a problem takes SP / 16 waves, an eight-wave workgroup 128 / SP problems.  Everything goes through
// v_mfma_f32_16x16x4_f32:
//   S = scale Q K^T -> P = softmax(S) (registers) -> LDS;   dP = dO V^T (registers);  delta = rowsum(P dP)
//   dV = P^T dO;   dS = scale P (dP - delta) -> LDS over P;   dQ = dS K;   dK = dS^T Q
// The grid is persistent (one workgroup per CU: the P / dS image leaves room for one): a workgroup walks its problems and
// holds the NEXT problem's Q / K / V / dO rows in registers while it computes the current one, so the global latency of
// the staging is off the critical path; two waves per SIMD cover each other's softmax and LDS phases.
// ---------------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;      // scores are taken to the log2 domain: p = exp2(s' - max') on v_exp_f32,
                                                  // as the forward kernel (token_attn_f32.hip) computes them
constexpr int AB_LD = 36;        // Q / K / V / dO rows: 32 columns + 4: 16-byte aligned rows for ds_read_b128 / ds_write_b128
#ifndef LIME_ATTN_BWD_ABLATE
#define LIME_ATTN_BWD_ABLATE 0   // tools/attn_bwd_ablate.py builds variants with phases removed (results garbage)
#endif

template <int SP>
__global__ __launch_bounds__(512) void token_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, long ld, const float* __restrict__ dout,
                                                              long ldo, float* __restrict__ dq, float* __restrict__ dk,
                                                              float* __restrict__ dv, long ldd, int n_seq, int S, int n_head,
                                                              int head_dim, int head_stride, float scale, int vec,
                                                              LimeDropout drop, const unsigned char* __restrict__ key_mask) {
    constexpr int NT = SP / 16;                 // 16-column score tiles per row
    constexpr int WPP = SP / 16;                // waves per problem
    constexpr int PPW = 8 / WPP;                // problems per workgroup
    constexpr int TPP = 64 * WPP;               // threads per problem
    constexpr int LDP = SP + 2;
    constexpr int PROB_FLOATS = 4 * SP * AB_LD + SP * LDP;
    constexpr int NV4 = SP * 8 / TPP;           // float4 per thread and operand (Q, K, V): 2
    constexpr int NV2 = SP * 16 / TPP;          // float2 per thread (dO): 4
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, kg = lane >> 4;
    const int pw = wave / WPP;                  // problem slot of this wave
    const int wr = wave % WPP;                  // which 16-row block of the problem
    const int lt = tid - pw * TPP;              // thread index inside the problem
    const long n_prob = (long)n_seq * n_head;
    float* Qs = smem + pw * PROB_FLOATS;
    float* Ks = Qs + SP * AB_LD;
    float* Vs = Ks + SP * AB_LD;
    float* Os = Vs + SP * AB_LD;                // dO
    float* Ps = Os + SP * AB_LD;                // P, then dS
    const int R0 = 16 * wr;
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};

    f32x4v rq[NV4], rk[NV4], rv[NV4];
    f32x2 ro[NV2];
    auto fetch = [&](long prob) {               // rows of problem `prob` -> registers (vec layout only)
        const bool live = prob < n_prob;
        const int seq = live ? (int)(prob / n_head) : 0, head = live ? (int)(prob % n_head) : 0;
        const long row_base = (long)seq * S;
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const int e = lt + TPP * u, r = e >> 3, c = (e & 7) * 4;
            const bool ok = live && r < S && !(LIME_ATTN_BWD_ABLATE & 1);
            const long g = (row_base + (ok ? r : 0)) * ld + (long)head * head_stride + c;
            rq[u] = ok ? *reinterpret_cast<const f32x4v*>(q + g) : z4;
            rk[u] = ok ? *reinterpret_cast<const f32x4v*>(k + g) : z4;
            rv[u] = ok ? *reinterpret_cast<const f32x4v*>(v + g) : z4;
        }
#pragma unroll
        for (int u = 0; u < NV2; ++u) {
            const int e = lt + TPP * u, r = e >> 4, c = (e & 15) * 2;
            const bool ok = live && r < S && c < head_dim && !(LIME_ATTN_BWD_ABLATE & 1);
            f32x2 d = {0.f, 0.f};
            if (ok) d = *reinterpret_cast<const f32x2*>(dout + (row_base + r) * ldo + (long)head * head_dim + c);
            ro[u] = d;
        }
    };
    auto commit = [&]() {                       // registers -> this problem's LDS images
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const int e = lt + TPP * u, r = e >> 3, c = (e & 7) * 4;
            *reinterpret_cast<f32x4v*>(&Qs[r * AB_LD + c]) = rq[u];
            *reinterpret_cast<f32x4v*>(&Ks[r * AB_LD + c]) = rk[u];
            *reinterpret_cast<f32x4v*>(&Vs[r * AB_LD + c]) = rv[u];
        }
#pragma unroll
        for (int u = 0; u < NV2; ++u) {
            const int e = lt + TPP * u, r = e >> 4, c = (e & 15) * 2;
            *reinterpret_cast<f32x2*>(&Os[r * AB_LD + c]) = ro[u];
        }
    };
    auto stage_scalar = [&](long prob) {        // any layout: straight into LDS (zero beyond S rows / head_dim columns)
        const bool live = prob < n_prob;
        const int seq = live ? (int)(prob / n_head) : 0, head = live ? (int)(prob % n_head) : 0;
        const long row_base = (long)seq * S;
        for (int e = lt; e < SP * 32; e += TPP) {
            const int r = e >> 5, c = e & 31;
            const bool ok = live && r < S && c < head_dim;
            const long g = (row_base + r) * ld + (long)head * head_stride + c;
            Qs[r * AB_LD + c] = ok ? q[g] : 0.f;
            Ks[r * AB_LD + c] = ok ? k[g] : 0.f;
            Vs[r * AB_LD + c] = ok ? v[g] : 0.f;
            Os[r * AB_LD + c] = ok ? dout[(row_base + r) * ldo + (long)head * head_dim + c] : 0.f;
        }
    };

    const long n_group = (n_prob + PPW - 1) / PPW;
    if (vec && (long)blockIdx.x < n_group) fetch((long)blockIdx.x * PPW + pw);
    for (long grp = blockIdx.x; grp < n_group; grp += gridDim.x) {
        const long prob = grp * PPW + pw;
        const bool live = prob < n_prob;
        const int seq = live ? (int)(prob / n_head) : 0, head = live ? (int)(prob % n_head) : 0;
        if (vec) commit(); else stage_scalar(prob);
        __syncthreads();
        if (vec && grp + gridDim.x < n_group) fetch((grp + gridDim.x) * PPW + pw);       // in flight during the compute below

        // ---- S tiles and dP tiles of this wave's 16 query rows ------------------------------------------------------
        f32x4 p[NT], dp[NT], pd[NT];
        {
            // the summation index d is only a label: lane group kg takes d = 8 kg .. 8 kg + 7 over the eight MFMA steps, so a
            // lane's eight operands are 32 consecutive bytes of its row -- two ds_read_b128 instead of eight ds_read_b32
            f32x4v qa[2], oa[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                qa[h] = *reinterpret_cast<const f32x4v*>(&Qs[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
                oa[h] = *reinterpret_cast<const f32x4v*>(&Os[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
            }
            // two score tiles at a time (four independent accumulator chains), the next pair's K / V fragments are requested
            // before this pair's MFMAs are issued
            f32x4v kb[2][2][2], vb[2][2][2];               // [buffer][tile of the pair][half]
            auto load_pair = [&](int buf, int ct) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr ((LIME_ATTN_BWD_ABLATE & 64) != 0) {      // no K / V fragment reads: MFMAs on register operands
                            kb[buf][t][h] = qa[h] + (float)ct;
                            vb[buf][t][h] = oa[h] + (float)ct;
                            continue;
                        }
                        kb[buf][t][h] = *reinterpret_cast<const f32x4v*>(&Ks[(16 * (ct + t) + fi) * AB_LD + 8 * kg + 4 * h]);
                        vb[buf][t][h] = *reinterpret_cast<const f32x4v*>(&Vs[(16 * (ct + t) + fi) * AB_LD + 8 * kg + 4 * h]);
                    }
            };
            load_pair(0, 0);
#pragma unroll
            for (int ct = 0; ct < NT; ct += 2) {
                const int buf = (ct >> 1) & 1;
                if (ct + 2 < NT) load_pair(buf ^ 1, ct + 2);
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, d0 = s0, s1 = s0, d1 = s0;
#pragma unroll
                for (int s = 0; s < ((LIME_ATTN_BWD_ABLATE & 2) ? 0 : 8); ++s) {
                    s0 = mfma16(qa[s >> 2][s & 3], kb[buf][0][s >> 2][s & 3], s0);
                    d0 = mfma16(oa[s >> 2][s & 3], vb[buf][0][s >> 2][s & 3], d0);
                    s1 = mfma16(qa[s >> 2][s & 3], kb[buf][1][s >> 2][s & 3], s1);
                    d1 = mfma16(oa[s >> 2][s & 3], vb[buf][1][s >> 2][s & 3], d1);
                }
                p[ct] = s0; dp[ct] = d0; p[ct + 1] = s1; dp[ct + 1] = d1;
            }
        }
        // key mask of layers.MultiHeadAttention (layers.py:227-232: masked_fill(mask == 0, -1e9) before the softmax): a masked
        // score is a constant, its dS is zero
        bool kmask[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
            kmask[ct] = key_mask != nullptr && live && 16 * ct + fi < S && key_mask[(long)seq * S + 16 * ct + fi] == 0;
        // softmax over the row (columns: tiles ct x the 16 lanes with the same kg), then delta and dS
#pragma unroll
        for (int r = 0; r < ((LIME_ATTN_BWD_ABLATE & 4) ? 0 : 4); ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const bool col_ok = 16 * ct + fi < S;
                const float sv = col_ok ? (kmask[ct] ? -1e9f * LOG2E : p[ct][r] * (scale * LOG2E)) : -INFINITY;
                p[ct][r] = sv;
                mx = fmaxf(mx, sv);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
            float sum = 0.f;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float e = __builtin_amdgcn_exp2f(p[ct][r] - mx);
                p[ct][r] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const float inv = 1.0f / sum;
            float dl = 0.f;
            // attention-probability dropout (nn.MultiheadAttention's dropout, newsEncoders.py:244-247): the forward used
            // keep * P / (1 - p); its mask is regenerated from the element's index.  kq[ct]: the factor on P for the dV product.
            const uint64_t mrow = ((uint64_t)prob * S + (uint64_t)(R0 + 4 * kg + r)) * (uint64_t)S;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                p[ct][r] *= inv;
                if (drop.thresh != 0) {
                    const float f = lime_keep(drop, mrow + (uint64_t)(16 * ct + fi)) ? drop.scale : 0.f;
                    dp[ct][r] *= f;
                    pd[ct][r] = p[ct][r] * f;
                }
                dl += p[ct][r] * dp[ct][r];
            }
            dl += __shfl_xor(dl, 1); dl += __shfl_xor(dl, 2); dl += __shfl_xor(dl, 4); dl += __shfl_xor(dl, 8);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) dp[ct][r] = kmask[ct] ? 0.f : scale * p[ct][r] * (dp[ct][r] - dl);      // dS
        }
        // P (as the forward multiplied it into V) -> LDS
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ps[(R0 + 4 * kg + r) * LDP + 16 * ct + fi] = drop.thresh != 0 ? pd[ct][r] : p[ct][r];
        __syncthreads();

        const long out_row0 = (long)seq * S + R0;
        auto store_tile = [&](float* dst, int dt, const f32x4& a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kg + r, col = 16 * dt + fi;
                if (live && R0 + row < S && col < head_stride && !((LIME_ATTN_BWD_ABLATE & 32) && a[r] != 12345.f))
                    dst[(out_row0 + row) * ldd + (long)head * head_stride + col] = a[r];
            }
        };
        // ---- dV[j, d] = sum_i P[i, j] dO[i, d] for the wave's key rows j ---------------------------------------------
        {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll 8
            for (int s = 0; s < ((LIME_ATTN_BWD_ABLATE & 8) ? 0 : SP / 4); ++s) {
                const int i = 4 * s + kg;
                const float a = Ps[i * LDP + R0 + fi];
                a0 = mfma16(a, Os[i * AB_LD + fi], a0);
                a1 = mfma16(a, Os[i * AB_LD + 16 + fi], a1);
            }
            store_tile(dv, 0, a0);
            store_tile(dv, 1, a1);
        }
        __syncthreads();
        // dS over P
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ps[(R0 + 4 * kg + r) * LDP + 16 * ct + fi] = dp[ct][r];
        __syncthreads();
        // ---- dQ[i, d] = sum_j dS[i, j] K[j, d];  dK[j, d] = sum_i dS[i, j] Q[i, d] -----------------------------------
        {
            f32x4 aq0 = {0.f, 0.f, 0.f, 0.f}, aq1 = aq0, ak0 = aq0, ak1 = aq0;
#pragma unroll 8
            for (int s = 0; s < ((LIME_ATTN_BWD_ABLATE & 16) ? 0 : SP / 4); ++s) {
                const int j = 4 * s + kg;
                const float ds_row = Ps[(R0 + fi) * LDP + j];          // dS[i = this wave's query row, j]
                aq0 = mfma16(ds_row, Ks[j * AB_LD + fi], aq0);
                aq1 = mfma16(ds_row, Ks[j * AB_LD + 16 + fi], aq1);
                const float ds_col = Ps[j * LDP + R0 + fi];            // dS[i = j, this wave's key row]
                ak0 = mfma16(ds_col, Qs[j * AB_LD + fi], ak0);
                ak1 = mfma16(ds_col, Qs[j * AB_LD + 16 + fi], ak1);
            }
            store_tile(dq, 0, aq0); store_tile(dq, 1, aq1);
            store_tile(dk, 0, ak0); store_tile(dk, 1, ak1);
        }
        __syncthreads();                        // the images are free for the next problem
    }
}

// ---------------------------------------------------------------------------------------------------
// Training-mode forward of the encoder attention WITH probability dropout: out = (keep * softmax(scale Q K^T) / (1 - p)) V.
// Eight waves, 16 query rows each, S <= 128; the scoring kernel in token_attn_f32.hip stays free of the mask arithmetic.
// The scores are computed TRANSPOSED (S^T = K Q^T): the MFMA result layout then holds, per lane, P^T[j = 4 kg + r][i = lane's
// query] -- exactly the B operand of the second product O^T = V^T P^T -- so the probabilities never leave the registers (no
// P image in LDS; K and V images only, 37 KB per workgroup); the softmax of a query is a reduction over the lane's own
// registers and its three kg partners; and the four r of an accumulator are four consecutive keys: one mask hash each.
// ---------------------------------------------------------------------------------------------------
template <int SP>
__global__ __launch_bounds__(512, 2) void token_attn_fwd_dropout_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                         const float* __restrict__ v, long ld, float* __restrict__ out,
                                                                         long ldo, int n_seq, int S, int n_head, int head_dim,
                                                                         int head_stride, float scale, LimeDropout drop, int vec) {
    constexpr int NT = SP / 16, WPP = SP / 16, PPW = 8 / WPP, TPP = 64 * WPP;
    constexpr int PROB_FLOATS = 2 * SP * AB_LD;            // K and V images; a lane's Q operand comes straight from global
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kg = lane >> 4;
    const int pw = wave / WPP, wr = wave % WPP, lt = tid - pw * TPP;
    const long n_prob = (long)n_seq * n_head;
    const long prob = (long)blockIdx.x * PPW + pw;
    const bool live = prob < n_prob;
    const int seq = live ? (int)(prob / n_head) : 0, head = live ? (int)(prob % n_head) : 0;
    float* Ks = smem + pw * PROB_FLOATS;
    float* Vs = Ks + SP * AB_LD;
    const int R0 = 16 * wr;
    const long row_base = (long)seq * S;
    // B operand of S^T = K Q^T: Q[i = R0 + fi][d = 8 kg .. 8 kg + 7] (k-permuted: 32 consecutive bytes of the lane's row)
    f32x4v qf[2];
    {
        const bool ok = live && R0 + fi < S;
        const float* qrow = q + (row_base + (ok ? R0 + fi : 0)) * ld + (long)head * head_stride + 8 * kg;
        if (vec) {
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            qf[0] = ok ? *reinterpret_cast<const f32x4v*>(qrow) : z;
            qf[1] = ok ? *reinterpret_cast<const f32x4v*>(qrow + 4) : z;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) qf[t >> 2][t & 3] = (ok && 8 * kg + t < head_dim) ? qrow[t] : 0.f;
        }
    }
    if (vec) {                  // 32-float head rows on 16-byte boundaries, zero padding columns
        for (int e = lt; e < SP * 8; e += TPP) {
            const int r = e >> 3, c = (e & 7) * 4;
            const bool ok = live && r < S;
            const long g = (row_base + (ok ? r : 0)) * ld + (long)head * head_stride + c;
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4v*>(&Ks[r * AB_LD + c]) = ok ? *reinterpret_cast<const f32x4v*>(k + g) : z;
            *reinterpret_cast<f32x4v*>(&Vs[r * AB_LD + c]) = ok ? *reinterpret_cast<const f32x4v*>(v + g) : z;
        }
    } else {
        for (int e = lt; e < SP * 32; e += TPP) {
            const int r = e >> 5, c = e & 31;
            const bool ok = live && r < S && c < head_dim;
            const long g = (row_base + r) * ld + (long)head * head_stride + c;
            Ks[r * AB_LD + c] = ok ? k[g] : 0.f;
            Vs[r * AB_LD + c] = ok ? v[g] : 0.f;
        }
    }
    __syncthreads();
    // S^T tiles: rows = keys 16 ct + 4 kg + r, column = this lane's query R0 + fi
    f32x4 p[NT];
    {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            f32x4v kf[2];                       // A operand: K[j = 16 ct + fi][d = 8 kg ..]
#pragma unroll
            for (int h = 0; h < 2; ++h) kf[h] = *reinterpret_cast<const f32x4v*>(&Ks[(16 * ct + fi) * AB_LD + 8 * kg + 4 * h]);
#pragma unroll
            for (int t = 0; t < 8; ++t) a = mfma16(kf[t >> 2][t & 3], qf[t >> 2][t & 3], a);
            p[ct] = a;
        }
    }
    // softmax over the keys of this lane's query: the lane's registers, then the kg partners (lanes +-16, +-32)
    float mx = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sv = (16 * ct + 4 * kg + r < S) ? p[ct][r] * (scale * LOG2E) : -INFINITY;
            p[ct][r] = sv;
            mx = fmaxf(mx, sv);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(p[ct][r] - mx);
            p[ct][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = drop.scale / sum;
    // mask element ((prob S + i) S + j): the four r of an accumulator are keys 16 ct + 4 kg .. + 3 -- one hash
    const uint64_t mrow = ((uint64_t)prob * S + (uint64_t)(R0 + fi)) * (uint64_t)S;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        unsigned m = 0xFu;
        if (drop.thresh != 0) {
            const uint64_t idx = mrow + (uint64_t)(16 * ct + 4 * kg);
            if ((idx & 3) == 0) m = lime_keep4(drop, idx >> 2);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) m = (m & ~(1u << r)) | ((lime_keep(drop, idx + r) ? 1u : 0u) << r);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) p[ct][r] = (m >> r) & 1u ? p[ct][r] * inv : 0.f;
    }
    // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]: A = V[j = 16 ct + 4 kg + t][d = 16 dt + fi] from LDS, B = p[ct][t] from registers
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * ct + 4 * kg + t;
            o0 = mfma16(Vs[j * AB_LD + fi], p[ct][t], o0);
            o1 = mfma16(Vs[j * AB_LD + 16 + fi], p[ct][t], o1);
        }
    // the lane holds O[i = R0 + fi][d = 4 kg + r] (o0) and [16 + 4 kg + r] (o1)
    if (live && R0 + fi < S) {
        float* o = out + (row_base + R0 + fi) * ldo + (long)head * head_dim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (4 * kg + r < head_dim) o[4 * kg + r] = o0[r];
            if (16 + 4 * kg + r < head_dim) o[16 + 4 * kg + r] = o1[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Sequences longer than 128 tokens (BASELINE config 4: body length 512): the same mathematics in 128 x 128 blocks.
//   attn_stats_kernel      per query row: lse = log2 sum_j exp(scale q.k_j) over ALL keys, delta = dO . O (O: the forward output)
//   attn_bwd_long_kernel   one workgroup per (sequence, head, key block): P = exp(scale S - lse) needs no row reduction any
//                          more; dK / dV of the block accumulate in registers over the query blocks, the dQ contributions of
//                          the key blocks are added with float atomics (dq is zeroed first)
// Eight waves, 16 rows each, as in token_attn_bwd_kernel.
// ---------------------------------------------------------------------------------------------------
constexpr int LB = 128;          // block edge

__device__ __forceinline__ void stage_rows(float* dst, const float* src, long ld, long row0, int rows_valid, int col0, int cols_valid,
                                           int tid) {
    // dst[r][c] (pitch AB_LD), r < 128, c < 32  <-  src[(row0 + r) * ld + col0 + c], zero outside the valid rows / columns
    for (int e = tid; e < LB * 32; e += 512) {
        const int r = e >> 5, c = e & 31;
        dst[r * AB_LD + c] = (r < rows_valid && c < cols_valid) ? src[(row0 + r) * ld + col0 + c] : 0.f;
    }
}

// dst: three bf16 images [128][SPLIT_PITCH] (split_mfma.h) of src rows row0 .. row0 + 127, columns col0 .. col0 + 31 (zero outside the
// valid rows / columns): the operand of the split-product Q K^T / dO V^T of the blocked kernels
__device__ __forceinline__ void stage_rows_split(unsigned short* dst, const float* src, long ld, long row0, int rows_valid, int col0,
                                                 int cols_valid, int tid) {
    for (int e = tid; e < LB * 16; e += 512) {
        const int r = e >> 4, c = (e & 15) * 2;
        const float* const p = src + (row0 + r) * ld + col0 + c;
        const float a = (r < rows_valid && c < cols_valid) ? p[0] : 0.f, b = (r < rows_valid && c + 1 < cols_valid) ? p[1] : 0.f;
        lime_dev::split_store2(dst, LB * lime_dev::SPLIT_PITCH, r, c, a, b);
    }
}

// SPX: the scores on the bf16 matrix cores as split products (K staged as three bf16 images, the wave's Q rows split in registers:
// six 16x16x32 MFMAs per 16 x 16 tile instead of eight 16x16x4 fp32 ones -- 96 cycles against 256)
template <bool SPX>
__global__ __launch_bounds__(512) void attn_stats_kernel(const float* __restrict__ q, const float* __restrict__ k, long ld,
                                                          const float* __restrict__ out, long ldout, const float* __restrict__ dout,
                                                          long ldo, float* __restrict__ stats, int S, int n_head, int head_dim,
                                                          int head_stride, float scale, int n_blk) {
    __shared__ float Qs[LB * AB_LD];
    __shared__ __attribute__((aligned(16))) float Ks[SPX ? (3 * LB * lime_dev::SPLIT_PITCH) / 2 : LB * AB_LD];
    unsigned short* const Kt = reinterpret_cast<unsigned short*>(Ks);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kg = lane >> 4;
    const int qb = blockIdx.x % n_blk;
    const long prob = blockIdx.x / n_blk;
    const int seq = (int)(prob / n_head), head = (int)(prob % n_head);
    const long row_base = (long)seq * S;
    const int q0 = qb * LB, q_valid = min(LB, S - q0);
    stage_rows(Qs, q, ld, row_base + q0, q_valid, head * head_stride, head_dim, tid);
    __syncthreads();
    const int R0 = 16 * wave;
    f32x4v qa[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) qa[h] = *reinterpret_cast<const f32x4v*>(&Qs[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
    lime_dev::SplitFrag qs;
    if constexpr (SPX) {
        const float x[8] = {qa[0][0], qa[0][1], qa[0][2], qa[0][3], qa[1][0], qa[1][1], qa[1][2], qa[1][3]};
        qs = lime_dev::split_frag(x);
    }
    float m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
    for (int kb = 0; kb < n_blk; ++kb) {
        const int k0 = kb * LB, k_valid = min(LB, S - k0);
        __syncthreads();
        if constexpr (SPX) stage_rows_split(Kt, k, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
        else stage_rows(Ks, k, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
        __syncthreads();
        f32x4 sc[LB / 16];
#pragma unroll
        for (int ct = 0; ct < LB / 16; ++ct) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SPX) {
                a = lime_dev::split_mfma16(qs, lime_dev::split_load(Kt, LB * lime_dev::SPLIT_PITCH, 16 * ct + fi, kg), a);
            } else {
                f32x4v kf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) kf[h] = *reinterpret_cast<const f32x4v*>(&Ks[(16 * ct + fi) * AB_LD + 8 * kg + 4 * h]);
#pragma unroll
                for (int t = 0; t < 8; ++t) a = mfma16(qa[t >> 2][t & 3], kf[t >> 2][t & 3], a);
            }
            sc[ct] = a;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int ct = 0; ct < LB / 16; ++ct) {
                const float sv = (16 * ct + fi < k_valid) ? sc[ct][r] * (scale * LOG2E) : -INFINITY;
                sc[ct][r] = sv;
                mx = fmaxf(mx, sv);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
            const float mn = fmaxf(m[r], mx);
            float sum = 0.f;
#pragma unroll
            for (int ct = 0; ct < LB / 16; ++ct) sum += __builtin_amdgcn_exp2f(sc[ct][r] - mn);
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            l[r] = l[r] * __builtin_amdgcn_exp2f(m[r] - mn) + sum;
            m[r] = mn;
        }
    }
    // lse of this wave's rows 4 kg + r (the 16 lanes of a kg group hold the same value), delta by one thread per row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = R0 + 4 * kg + r;
        if (fi == 0 && row < q_valid) stats[((row_base + q0 + row) * n_head + head) * 2] = m[r] + log2f(l[r]);   // log2 domain
    }
    if (tid < q_valid && out != nullptr) {
        const float* po = out + (row_base + q0 + tid) * ldout + (long)head * head_dim;
        const float* pd = dout + (row_base + q0 + tid) * ldo + (long)head * head_dim;
        float d = 0.f;
        for (int c = 0; c < head_dim; ++c) d += po[c] * pd[c];
        stats[((row_base + q0 + tid) * n_head + head) * 2 + 1] = d;
    }
}

// Training-mode forward for 128 < S <= 512 with probability dropout: row statistics first (attn_stats_kernel), then one
// workgroup per (sequence, head, query block) walks the key blocks: out = sum_blocks (keep * exp2(s' - lse) / (1 - p)) V.
__global__ __launch_bounds__(512) void attn_fwd_long_dropout_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                     const float* __restrict__ v, long ld, const float* __restrict__ stats,
                                                                     float* __restrict__ out, long ldo, int S, int n_head, int head_dim,
                                                                     int head_stride, float scale, int n_blk, LimeDropout drop) {
    constexpr int NT = LB / 16, LDP = LB + 2;
    extern __shared__ float smem[];
    float* Qs = smem;
    float* Ks = Qs + LB * AB_LD;
    float* Vs = Ks + LB * AB_LD;
    float* Ps = Vs + LB * AB_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kg = lane >> 4;
    const int qb = blockIdx.x % n_blk;
    const long prob = blockIdx.x / n_blk;
    const int seq = (int)(prob / n_head), head = (int)(prob % n_head);
    const long row_base = (long)seq * S;
    const int q0 = qb * LB, q_valid = min(LB, S - q0);
    const int R0 = 16 * wave;
    stage_rows(Qs, q, ld, row_base + q0, q_valid, head * head_stride, head_dim, tid);
    __syncthreads();
    f32x4v qa[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) qa[h] = *reinterpret_cast<const f32x4v*>(&Qs[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
    float lse[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = R0 + 4 * kg + r;
        lse[r] = row < q_valid ? stats[((row_base + q0 + row) * n_head + head) * 2] : INFINITY;
    }
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
    for (int kb = 0; kb < n_blk; ++kb) {
        const int k0 = kb * LB, k_valid = min(LB, S - k0);
        __syncthreads();
        stage_rows(Ks, k, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
        stage_rows(Vs, v, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            f32x4v kf[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) kf[h] = *reinterpret_cast<const f32x4v*>(&Ks[(16 * ct + fi) * AB_LD + 8 * kg + 4 * h]);
#pragma unroll
            for (int t = 0; t < 8; ++t) a = mfma16(qa[t >> 2][t & 3], kf[t >> 2][t & 3], a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t idx = ((uint64_t)prob * S + (uint64_t)(q0 + R0 + 4 * kg + r)) * (uint64_t)S + (uint64_t)(k0 + 16 * ct + fi);
                const float pv = (16 * ct + fi < k_valid) ? __builtin_amdgcn_exp2f(a[r] * (scale * LOG2E) - lse[r]) : 0.f;
                const float f = (drop.thresh == 0 || lime_keep(drop, idx)) ? drop.scale : 0.f;
                Ps[(R0 + 4 * kg + r) * LDP + 16 * ct + fi] = pv * f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < LB / 4; ++t) {
            const int j = 4 * t + kg;
            const float a = Ps[(R0 + fi) * LDP + j];
            o0 = mfma16(a, Vs[j * AB_LD + fi], o0);
            o1 = mfma16(a, Vs[j * AB_LD + 16 + fi], o1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = R0 + 4 * kg + r;
        if (row < q_valid) {
            float* o = out + (row_base + q0 + row) * ldo + (long)head * head_dim;
            if (fi < head_dim) o[fi] = o0[r];
            if (16 + fi < head_dim) o[16 + fi] = o1[r];
        }
    }
}

__global__ __launch_bounds__(512) void attn_bwd_long_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, long ld, const float* __restrict__ dout,
                                                             long ldo, const float* __restrict__ stats, float* __restrict__ dq,
                                                             float* __restrict__ dk, float* __restrict__ dv, long ldd, int S,
                                                             int n_head, int head_dim, int head_stride, float scale, int n_blk,
                                                             LimeDropout drop, float* __restrict__ dq_slabs, long n_tok) {
    constexpr int NT = LB / 16, LDP = LB + 2;
    extern __shared__ float smem[];
    float* Qs = smem;
    float* Ks = Qs + LB * AB_LD;
    float* Vs = Ks + LB * AB_LD;
    float* Os = Vs + LB * AB_LD;
    float* Ps = Os + LB * AB_LD;
    float* Ls = Ps + LB * LDP;                  // lse of the query block
    float* Ds = Ls + LB;                        // delta of the query block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kg = lane >> 4;
    const int kb = blockIdx.x % n_blk;
    const long prob = blockIdx.x / n_blk;
    const int seq = (int)(prob / n_head), head = (int)(prob % n_head);
    const long row_base = (long)seq * S;
    const int k0 = kb * LB, k_valid = min(LB, S - k0);
    const int R0 = 16 * wave;
    stage_rows(Ks, k, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
    stage_rows(Vs, v, ld, row_base + k0, k_valid, head * head_stride, head_dim, tid);
    f32x4 av0 = {0.f, 0.f, 0.f, 0.f}, av1 = av0, ak0 = av0, ak1 = av0;
    for (int qb = 0; qb < n_blk; ++qb) {
        const int q0 = qb * LB, q_valid = min(LB, S - q0);
        __syncthreads();                        // the previous block's images are no longer read
        stage_rows(Qs, q, ld, row_base + q0, q_valid, head * head_stride, head_dim, tid);
        stage_rows(Os, dout, ldo, row_base + q0, q_valid, head * head_dim, head_dim, tid);
        if (tid < LB) {
            const bool ok = tid < q_valid;
            const float* st = stats + ((row_base + q0 + (ok ? tid : 0)) * n_head + head) * 2;
            Ls[tid] = ok ? st[0] : INFINITY;    // exp(x - inf) = 0: rows beyond S contribute nothing
            Ds[tid] = ok ? st[1] : 0.f;
        }
        __syncthreads();
        f32x4 p[NT], dp[NT];
        {
            f32x4v qa[2], oa[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                qa[h] = *reinterpret_cast<const f32x4v*>(&Qs[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
                oa[h] = *reinterpret_cast<const f32x4v*>(&Os[(R0 + fi) * AB_LD + 8 * kg + 4 * h]);
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, d0 = s0;
                f32x4v kf[2], vf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    kf[h] = *reinterpret_cast<const f32x4v*>(&Ks[(16 * ct + fi) * AB_LD + 8 * kg + 4 * h]);
                    vf[h] = *reinterpret_cast<const f32x4v*>(&Vs[(16 * ct + fi) * AB_LD + 8 * kg + 4 * h]);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    s0 = mfma16(qa[t >> 2][t & 3], kf[t >> 2][t & 3], s0);
                    d0 = mfma16(oa[t >> 2][t & 3], vf[t >> 2][t & 3], d0);
                }
                p[ct] = s0; dp[ct] = d0;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lse = Ls[R0 + 4 * kg + r], dl = Ds[R0 + 4 * kg + r];
            const uint64_t mrow = ((uint64_t)prob * S + (uint64_t)(q0 + R0 + 4 * kg + r)) * (uint64_t)S + (uint64_t)k0;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float pv = (16 * ct + fi < k_valid) ? __builtin_amdgcn_exp2f(p[ct][r] * (scale * LOG2E) - lse) : 0.f;
                // probability dropout: the forward multiplied keep / (1 - p) into P before the V product (delta = dO . O
                // already contains it)
                const float f = (drop.thresh == 0 || lime_keep(drop, mrow + (uint64_t)(16 * ct + fi))) ? drop.scale : 0.f;
                p[ct][r] = pv * f;                                   // what the dV product needs
                dp[ct][r] = scale * pv * (dp[ct][r] * f - dl);       // dS
            }
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ps[(R0 + 4 * kg + r) * LDP + 16 * ct + fi] = p[ct][r];
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < LB / 4; ++t) {      // dV[j, d] += sum_i P[i, j] dO[i, d]
            const int i = 4 * t + kg;
            const float a = Ps[i * LDP + R0 + fi];
            av0 = mfma16(a, Os[i * AB_LD + fi], av0);
            av1 = mfma16(a, Os[i * AB_LD + 16 + fi], av1);
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ps[(R0 + 4 * kg + r) * LDP + 16 * ct + fi] = dp[ct][r];
        __syncthreads();
        f32x4 aq0 = {0.f, 0.f, 0.f, 0.f}, aq1 = aq0;
#pragma unroll 8
        for (int t = 0; t < LB / 4; ++t) {
            const int j = 4 * t + kg;
            const float ds_row = Ps[(R0 + fi) * LDP + j];
            aq0 = mfma16(ds_row, Ks[j * AB_LD + fi], aq0);
            aq1 = mfma16(ds_row, Ks[j * AB_LD + 16 + fi], aq1);
            const float ds_col = Ps[j * LDP + R0 + fi];
            ak0 = mfma16(ds_col, Qs[j * AB_LD + fi], ak0);
            ak1 = mfma16(ds_col, Qs[j * AB_LD + 16 + fi], ak1);
        }
        // this key block's share of dQ: key block 0 stores straight into dq, block kb > 0 into slab kb - 1 ([tokens][n_head * 32]);
        // attn_dq_reduce_kernel adds the slabs in block order -- no atomics, the result is bitwise reproducible
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = R0 + 4 * kg + r;
            if (row < q_valid) {
                float* d = kb == 0 ? dq + (row_base + q0 + row) * ldd + (long)head * head_stride
                                   : dq_slabs + ((long)(kb - 1) * n_tok + row_base + q0 + row) * ((long)n_head * 32) + (long)head * 32;
                if (fi < head_stride) d[fi] = aq0[r];
                if (16 + fi < head_stride) d[16 + fi] = aq1[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = R0 + 4 * kg + r;
        if (row < k_valid) {
            const long o = (row_base + k0 + row) * ldd + (long)head * head_stride;
            if (fi < head_stride) { dv[o + fi] = av0[r]; dk[o + fi] = ak0[r]; }
            if (16 + fi < head_stride) { dv[o + 16 + fi] = av1[r]; dk[o + 16 + fi] = ak1[r]; }
        }
    }
}

// The blocked backward with all five products as split products on the bf16 matrix cores (lime_set_split_gemm(1), the default;
// split_mfma.h: six 16x16x32 MFMAs per 16 x 16 x 32 block, 96 matrix cycles against the 256 of eight fp32 ones).  The key block's K and
// V are staged once, the query side goes in blocks of 64 rows; K, V, Q, dO are swizzled split images that serve row reads and transposed
// reads (split_mfma.h).  Wave w owns the keys 16 w .. 16 w + 15 of the block:
//   S = Q K^T and dP = dO V^T for its 16 keys x the 64 queries: result tiles with the KEY on the lane and the queries on the registers --
//   so P~ and dS are, as they stand (split in registers), the B operands of the two products that sum over the queries:
//   dV^T[d, j] += sum_i dO^T[d, i] P~[i, j],  dK^T[d, j] += sum_i Q^T[d, i] dS[i, j]   (dO^T / Q^T: ds_read_b64_tr_b16 block reads),
//   accumulated in registers over the query blocks, no P image, no barrier between the score products and these.
//   dQ sums over the keys -- the other orientation: dS goes through an fp32 image once ([query][key], 528-byte rows); wave w = query tile
//   w & 3 x head-dim half w >> 2 reads its queries' rows back as B operands (two ds_read_b128 per step, split in registers) against K^T
//   (transposed block reads of the K image): dQ^T[d, i] = sum_j K^T[d, j] dS^T[j, i].  Slabs per key block and the reduction order over
//   key blocks as before.
// FOUR-wave workgroups, TWO per CU (60 KB of LDS each): the phases of a query block are short and separated by barriers, so one lock-step
// workgroup per CU left the matrix pipe, the VALU and the LDS each 25-40 % busy one after the other (profiles/r03_attn_bwd_sp_counters.txt);
// two independent workgroups overlap them.  Wave w owns the keys 32 w .. 32 w + 31 (two 16-key tiles; their K and V fragments stay in
// registers for the whole key block, V never goes to LDS), the query side goes in blocks of 32 rows.  dS reaches the dQ product as a
// split image [key][query] (the producer writes the three terms it has split for dK anyway, four consecutive queries = 8 bytes per
// term), read back TRANSPOSED: no second split.  Three barriers per query block, 960 matrix cycles per wave and block (dV / dK / dQ on
// the fp32 MFMA: 3072).
constexpr int LQ = 32;
__global__ __launch_bounds__(256, 2) void attn_bwd_long_sp_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, long ld, const float* __restrict__ dout,
                                                                   long ldo, const float* __restrict__ stats, float* __restrict__ dq,
                                                                   float* __restrict__ dk, float* __restrict__ dv, long ldd, int S,
                                                                   int n_head, int head_dim, int head_stride, float scale, int n_blk,
                                                                   LimeDropout drop, float* __restrict__ dq_slabs, long n_tok) {
    using namespace lime_dev;
    constexpr int KT = LB * SWZ_ROW, QT = LQ * SWZ_ROW;
    static_assert(LQ == SWZ_ROW, "the dS image has one row per key and LQ queries per row");
    extern __shared__ float smem[];
    unsigned short* const Kt = reinterpret_cast<unsigned short*>(smem);       // three swizzled bf16 images of the K block, [LB][32]
    unsigned short* const Di = Kt + 3 * KT;      // ... of dS^T: [LB keys][LQ queries]
    unsigned short* const Qi = Di + 3 * KT;      // ... of the query block's Q rows, [LQ][32]
    unsigned short* const Oi = Qi + 3 * QT;      // ... and of its dO rows
    float* const Ls = smem + 3 * KT + 3 * QT;    // (2 x 3 KT + 2 x 3 QT bf16)
    float* const Ds = Ls + LQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kg = lane >> 4;
    const int kb = blockIdx.x % n_blk;
    const long prob = blockIdx.x / n_blk;
    const int seq = (int)(prob / n_head), head = (int)(prob % n_head);
    const long row_base = (long)seq * S;
    const int k0 = kb * LB, k_valid = min(LB, S - k0);
    const int R0 = 32 * wave, qi = wave & 1, hh = wave >> 1;
    for (int e = tid; e < LB * 16; e += 256) {      // K rows -> split images (zero outside the valid rows / columns)
        const int r = e >> 4, c = (e & 15) * 2;
        const bool ok0 = r < k_valid && c < head_dim, ok1 = r < k_valid && c + 1 < head_dim;
        const long g = (row_base + k0 + (r < k_valid ? r : 0)) * ld + head * head_stride + c;
        swz_store2(Kt, KT, r, c, ok0 ? k[g] : 0.f, ok1 ? k[g + 1] : 0.f);
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[2][2] = {{z4, z4}, {z4, z4}}, ak[2][2] = {{z4, z4}, {z4, z4}};   // dV^T / dK^T of key tile u: [head dim 16 c + 4 kg + r][key R0 + 16 u + fi]
    const int n_qblk = (S + LQ - 1) / LQ;
    // the NEXT query block's Q / dO rows (two pairs each per thread) and statistics wait in registers while the current block is computed
    float pq[4], po[4], pl = INFINITY, pd = 0.f;
    auto fetch_q = [&](int qb) {
        const int q0 = qb * LQ, q_valid = min(LQ, S - q0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c = (e & 15) * 2;
            const bool ok0 = r < q_valid && c < head_dim, ok1 = r < q_valid && c + 1 < head_dim;
            const float* const qs = q + (row_base + q0 + (r < q_valid ? r : 0)) * ld + head * head_stride + c;
            const float* const os = dout + (row_base + q0 + (r < q_valid ? r : 0)) * ldo + head * head_dim + c;
            pq[2 * u] = ok0 ? qs[0] : 0.f; pq[2 * u + 1] = ok1 ? qs[1] : 0.f;
            po[2 * u] = ok0 ? os[0] : 0.f; po[2 * u + 1] = ok1 ? os[1] : 0.f;
        }
        if (tid < LQ) {
            const bool ok = tid < q_valid;
            const float* st = stats + ((row_base + q0 + (ok ? tid : 0)) * n_head + head) * 2;
            pl = ok ? st[0] : INFINITY;         // exp(x - inf) = 0: rows beyond S contribute nothing
            pd = ok ? st[1] : 0.f;
        }
    };
    fetch_q(0);
    // this wave's V rows straight from global memory into split fragments (lane: key R0 + 16 u + fi, head dims 8 kg .. 8 kg + 7)
    SplitFrag kB[2], vB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = R0 + 16 * u + fi;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * kg + e;
            x[e] = (r < k_valid && c < head_dim) ? v[(row_base + k0 + r) * ld + head * head_stride + c] : 0.f;
        }
        vB[u] = split_frag(x);
    }
    __syncthreads();                            // the K image is complete
#pragma unroll
    for (int u = 0; u < 2; ++u) kB[u] = swz_row_load(Kt, KT, R0 + 16 * u + fi, kg);
    const float c2 = scale * LOG2E;
    for (int qb = 0; qb < n_qblk; ++qb) {
        const int q0 = qb * LQ, q_valid = min(LQ, S - q0);
        __syncthreads();                        // the previous block's images are no longer read
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c = (e & 15) * 2;
            swz_store2(Qi, QT, r, c, pq[2 * u], pq[2 * u + 1]);
            swz_store2(Oi, QT, r, c, po[2 * u], po[2 * u + 1]);
        }
        if (tid < LQ) { Ls[tid] = pl; Ds[tid] = pd; }
        __syncthreads();
        if (qb + 1 < n_qblk) fetch_q(qb + 1);
        // ---- S and dP of this wave's two key tiles x the two query tiles, then P~ and dS in place -----------------------------
        f32x4 p[2][2], dp[2][2];                // [key tile u][query tile t]: [query 16 t + 4 kg + r][key R0 + 16 u + fi]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SplitFrag qA = swz_row_load(Qi, QT, 16 * t + fi, kg), oA = swz_row_load(Oi, QT, 16 * t + fi, kg);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                p[u][t] = split_mfma16(qA, kB[u], z4);
                dp[u][t] = split_mfma16(oA, vB[u], z4);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 lse4 = *reinterpret_cast<const f32x4*>(Ls + 16 * t + 4 * kg);
            const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Ds + 16 * t + 4 * kg);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = R0 + 16 * u + fi;
                const bool key_ok = col < k_valid;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * kg + r;
                    const float pv = key_ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(p[u][t][r], c2, -lse4[r])) : 0.f;
                    float f = 1.f;
                    if (drop.thresh != 0)
                        f = lime_keep(drop, ((uint64_t)prob * S + (uint64_t)(q0 + row)) * (uint64_t)S + (uint64_t)(k0 + col)) ? drop.scale : 0.f;
                    p[u][t][r] = pv * f;                                    // what the dV product needs
                    dp[u][t][r] = scale * pv * (dp[u][t][r] * f - dl4[r]);  // dS
                }
            }
        }
        // ---- dV^T += dO^T P~: the result registers of the two query tiles are the eight k values of the B operand ---------------------
        {
            const SplitFrag o0 = swz_tr_load(Oi, QT, 0, 0, fi, kg), o1 = swz_tr_load(Oi, QT, 0, 1, fi, kg);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const SplitFrag bp = split_frag(p[u][0], p[u][1]);
                av[u][0] = split_mfma16(o0, bp, av[u][0]);
                av[u][1] = split_mfma16(o1, bp, av[u][1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (one pair of transposed fragments at a time: registers)
        // ---- dK^T += Q^T dS; the same three terms of dS go to the [key][query] image the dQ product reads (8 bytes per tile and term) ----
        {
            const SplitFrag q0f = swz_tr_load(Qi, QT, 0, 0, fi, kg), q1f = swz_tr_load(Qi, QT, 0, 1, fi, kg);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const SplitFrag bd = split_frag(dp[u][0], dp[u][1]);
                const u32x4 bh = __builtin_bit_cast(u32x4, bd.h), bm = __builtin_bit_cast(u32x4, bd.m), bl = __builtin_bit_cast(u32x4, bd.l);
#pragma unroll
                for (int t = 0; t < 2; ++t) {   // elements 4 t .. 4 t + 3 of the fragment: queries 16 t + 4 kg .. + 3 of key R0 + 16 u + fi
                    unsigned short* const d = Di + swz_off(R0 + 16 * u + fi, 16 * t + 4 * kg);
                    *reinterpret_cast<u32x2*>(d) = u32x2{bh[2 * t], bh[2 * t + 1]};
                    *reinterpret_cast<u32x2*>(d + KT) = u32x2{bm[2 * t], bm[2 * t + 1]};
                    *reinterpret_cast<u32x2*>(d + 2 * KT) = u32x2{bl[2 * t], bl[2 * t + 1]};
                }
                ak[u][0] = split_mfma16(q0f, bd, ak[u][0]);
                ak[u][1] = split_mfma16(q1f, bd, ak[u][1]);
            }
        }
        __syncthreads();                        // the dS image is complete
        // ---- this key block's share of dQ^T (head dims 16 hh .. 16 hh + 15 x query tile qi) = K^T dS^T: both operands by transposed
        // block reads, k values = keys 16 t0 + 4 kg + {0..3} and 16 (t0 + 1) + 4 kg + {0..3} ------------------------------------------
        f32x4 aq = z4;
#pragma unroll
        for (int s4 = 0; s4 < LB / 32; ++s4)
            aq = split_mfma16(swz_tr_load(Kt, KT, 2 * s4, hh, fi, kg), swz_tr_load(Di, KT, 2 * s4, qi, fi, kg), aq);
        {
            const int row = 16 * qi + fi;       // aq[r] = dQ[query row][head dim 16 hh + 4 kg + r]
            if (row < q_valid) {
                float* d = kb == 0 ? dq + (row_base + q0 + row) * ldd + (long)head * head_stride
                                   : dq_slabs + ((long)(kb - 1) * n_tok + row_base + q0 + row) * ((long)n_head * 32) + (long)head * 32;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * hh + 4 * kg + r < head_stride) d[16 * hh + 4 * kg + r] = aq[r];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = R0 + 16 * u + fi;
        if (col < k_valid) {
            const long o = (row_base + k0 + col) * ldd + (long)head * head_stride;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d0 = 4 * kg + r;
                if (d0 < head_stride) { dv[o + d0] = av[u][0][r]; dk[o + d0] = ak[u][0][r]; }
                if (16 + d0 < head_stride) { dv[o + 16 + d0] = av[u][1][r]; dk[o + 16 + d0] = ak[u][1][r]; }
            }
        }
    }
}

// stats[(token, head)] = (lse from the forward, delta = dO . O): the statistics pass without its Q K^T products, for a forward that
// kept its log-sum-exp (lime_token_attention_lse_f32)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ out, long ldout, const float* __restrict__ dout, long ldo,
                                                          const float* __restrict__ lse, float* __restrict__ stats, long n_tok, int n_head,
                                                          int head_dim) {
    // one wave per token: coalesced row reads, the products parked in LDS, lane h sums head h's in column order (fixed order:
    // reproducible bits); n_head * head_dim <= 1024, n_head <= 64
    __shared__ float prod[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int width = n_head * head_dim;
    for (long row = (long)blockIdx.x * 4 + wave; row < n_tok; row += (long)gridDim.x * 4) {
        const float* po = out + row * ldout;
        const float* pd = dout + row * ldo;
        for (int c = lane; c < width; c += 64) prod[wave][c] = po[c] * pd[c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < n_head) {
            float d = 0.f;
            for (int c = 0; c < head_dim; ++c) d += prod[wave][lane * head_dim + c];
            const long e = row * n_head + lane;
            stats[e * 2] = lse[e];
            stats[e * 2 + 1] = d;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// dq[row, head * hs + j] += sum over slabs 0 .. n_slab - 1 (key blocks 1 ..) of slab[row][head * 32 + j], in slab order
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(float* __restrict__ dq, long ldd, const float* __restrict__ slabs, long n_tok,
                                                              int n_head, int hs, int n_slab) {
    const long total = n_tok * n_head * hs;
    const long wcols = (long)n_head * 32;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / (n_head * hs);
        const int c = (int)(e - row * (n_head * hs));
        const int head = c / hs, j = c - head * hs;
        float t = dq[row * ldd + c];
        for (int b = 0; b < n_slab; ++b) t += slabs[((long)b * n_tok + row) * wcols + head * 32 + j];
        dq[row * ldd + c] = t;
    }
}

// ---------------------------------------------------------------------------------------------------
// word-table gradient: dTable[ids[r], :] += dX[r, :].  A workgroup walks 512 consecutive rows, one wave per row; rows
// whose id is `hot_id` (the padding word, a large share of all tokens) are summed in registers and added once per wave.
// ---------------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ ids, const float* __restrict__ dx, long lddx,
                                                         float* __restrict__ dtable, long ldt, long rows, int dim, int hot_id,
                                                         int rows_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float hot[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) hot[j] = 0.f;
    bool any_hot = false;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    for (long r = r0 + wave; r < r1; r += 4) {
        const int id = ids[r];
        const float* p = dx + r * lddx;
        if (id == hot_id) {
            any_hot = true;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int c = lane + 64 * j;
                if (c < dim) hot[j] += p[c];
            }
        } else {
            float* t = dtable + (long)id * ldt;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int c = lane + 64 * j;
                if (c < dim) unsafeAtomicAdd(t + c, p[c]);
            }
        }
    }
    if (any_hot) {
        float* t = dtable + (long)hot_id * ldt;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            if (c < dim) unsafeAtomicAdd(t + c, hot[j]);
        }
    }
}

// Tables of at most 32 rows (the freshness / lifetime bucket embeddings, 10 rows x 500): every row receives hundreds of
// contributions, so atomics would serialise.  A workgroup owns 64 columns; its four waves walk the rows r = wave, wave + 4,
// ... and add into a private [32][64] LDS image each; the four images are summed in a fixed order.  No atomics.
__global__ __launch_bounds__(512) void embed_bwd_small_kernel(const int* __restrict__ ids, const float* __restrict__ dx, long lddx,
                                                               float* __restrict__ dtable, long ldt, long rows, int dim,
                                                               int table_rows) {
    // eight waves, each with its own [32][64] image of the table and sixteen rows in flight (one 64-column workgroup walks every
    // row: with four waves and eight rows in flight the 1760 rows of a 50-column table took 55 us of dependent round trips)
    __shared__ float acc[8][32][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    for (int t = 0; t < 32; ++t) acc[wave][t][lane] = 0.f;
    if (c < dim) {
        long r = wave;
        for (; r + 120 < rows; r += 128) {
            int id[16];
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { id[u] = ids[r + 8 * u]; v[u] = dx[(r + 8 * u) * lddx + c]; }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (id[u] >= 0 && id[u] < table_rows) acc[wave][id[u]][lane] += v[u];
        }
        for (; r < rows; r += 8) {
            const int id = ids[r];
            if (id >= 0 && id < table_rows) acc[wave][id][lane] += dx[r * lddx + c];
        }
    }
    __syncthreads();
    if (c < dim)
        for (int t = wave; t < table_rows; t += 8)
            dtable[(long)t * ldt + c] += ((acc[0][t][lane] + acc[1][t][lane]) + (acc[2][t][lane] + acc[3][t][lane])) +
                                         ((acc[4][t][lane] + acc[5][t][lane]) + (acc[6][t][lane] + acc[7][t][lane]));
}

// ---------------------------------------------------------------------------------------------------
// optimizer: sum of squares -> clip coefficient -> Adam   (trainer.py:33, 146-148)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) s += g[e] * g[e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total norm, out[1] = min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_); max_norm <= 0: 1
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += partial[e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, float bias1, float bias2_sqrt,
                                                    const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.0f;
    const float step = lr / bias1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        float gr = g[e] * gs;
        const float pe = p[e];
        if (weight_decay != 0.f) gr += weight_decay * pe;
        const float me = beta1 * m[e] + (1.0f - beta1) * gr;
        const float ve = beta2 * v[e] + (1.0f - beta2) * gr * gr;
        m[e] = me;
        v[e] = ve;
        p[e] = pe - step * (me / (sqrtf(ve) / bias2_sqrt + eps));
    }
}

// loss = mean_b (-log_softmax(logits[b])[0]);  dlogits = (softmax - onehot_0) / B      (trainer.py:71-73)
__global__ __launch_bounds__(256) void nll_softmax_kernel(const float* __restrict__ logits, long ld, int B, int K,
                                                           float* __restrict__ loss, float* __restrict__ dlogits, long ldd) {
    __shared__ float red[4];
    float part = 0.f;
    const float inv_b = 1.0f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* x = logits + (long)b * ld;
        float mx = x[0];
        for (int j = 1; j < K; ++j) mx = fmaxf(mx, x[j]);
        float s = 0.f;
        for (int j = 0; j < K; ++j) s += expf(x[j] - mx);
        const float lse = mx + logf(s);
        part += lse - x[0];
        if (dlogits)
            for (int j = 0; j < K; ++j) dlogits[(long)b * ldd + j] = (expf(x[j] - lse) - (j == 0 ? 1.0f : 0.f)) * inv_b;
    }
    const float tot = block_sum_4(part, red);
    if (threadIdx.x == 0) *loss = tot * inv_b;
}

}  // namespace

// ===================================================================================================
// C ABI
// ===================================================================================================
namespace {

struct WgradPlan { int nkt, tk, n_tiles, k_tiles, splits, rows_per_split; long np, kp; };

WgradPlan wgrad_plan(int M, int N, int K, int wg_per_cu = 1) {
    WgradPlan w;
    long best = -1;
    w.nkt = 4;
    for (int nkt = 3; nkt <= 5; ++nkt) {                          // the tile width 64 * nkt that pads K the least
        const long tk = 64L * nkt, kp = ((long)K + tk - 1) / tk * tk;
        if (best < 0 || kp < best || (kp == best && nkt == 4)) { best = kp; w.nkt = nkt; }
    }
    w.tk = 64 * w.nkt;
    w.n_tiles = (N + WG_TN - 1) / WG_TN;
    w.k_tiles = (K + w.tk - 1) / w.tk;
    w.np = (long)w.n_tiles * WG_TN;
    w.kp = (long)w.k_tiles * w.tk;
    const int ntile = w.n_tiles * w.k_tiles;
    int splits = 256 * wg_per_cu / ntile;                         // a single round of eight-wave workgroups, one per CU
    // at least 8 chunks of 64 rows per workgroup -- but the small-M problems of the layers around the encoders (B x 55 = 1760 rows: 36
    // workgroups of 9 serial chunks took 50 us) go down to 2 chunks so that a launch reaches ~150 workgroups
    const int max_splits = M >= 8192 ? (M + 511) / 512 : (M + 127) / 128;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rps = (M + splits - 1) / splits;
    rps = (rps + WG_MC - 1) / WG_MC * WG_MC;
    w.rows_per_split = rps;
    w.splits = (M + rps - 1) / rps;
    if (w.splits < 1) w.splits = 1;
    return w;
}

template <int NKT>
int launch_wgrad_dma(const WgradPlan& w, const float* dy, long ldy, const float* x, long ldx, float* ws, int M, int N, int K,
                     int ones_col, hipStream_t s) {
    constexpr int BYTES = 2 * WD_MC * (WG_TN + 64 * NKT) * 4;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)wgrad_dma_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_linear_wgrad_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        configured = true;
    }
    const int grid = w.n_tiles * w.k_tiles * w.splits;
    wgrad_dma_kernel<NKT><<<grid, WG_THREADS, BYTES, s>>>(dy, ldy, x, ldx, ws, M, N, K, w.n_tiles, w.k_tiles, w.rows_per_split, ones_col);
    return lime_check_launch("wgrad_dma_kernel");
}

template <int NKT, bool VEC>
int launch_wgrad(const WgradPlan& w, const float* dy, long ldy, const float* x, long ldx, float* ws, int M, int N, int K,
                 int ones_col, hipStream_t s) {
    constexpr int BYTES = WG_MC * ((WG_TN + 16) + (64 * NKT + 16)) * 4;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<NKT, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_linear_wgrad_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        configured = true;
    }
    const int grid = w.n_tiles * w.k_tiles * w.splits;
    wgrad_kernel<NKT, VEC><<<grid, WG_THREADS, BYTES, s>>>(dy, ldy, x, ldx, ws, M, N, K, w.n_tiles, w.k_tiles, w.rows_per_split, ones_col);
    return lime_check_launch("wgrad_kernel");
}

// extra (optional): column `cols` of the partial grid summed into extra[rows] as well (the caller guarantees the grid has that column)
int launch_reduce(const float* ws, long split_stride, int splits, long ldw, float* out, long ldo, int rows, int cols,
                  int accumulate, hipStream_t s, float* extra = nullptr) {
    const long total = (long)rows * cols;
    if (cols % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && split_stride % 4 == 0 && ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) == 0) {
        const long groups = (long)rows * (cols / 4 + (extra ? 1 : 0));
        const int grid4 = (int)((groups + 63) / 64);
        reduce_partials_vec4_kernel<<<grid4, 256, 0, s>>>(ws, split_stride, splits, ldw, out, ldo, rows, cols, accumulate, extra);
        return lime_check_launch("reduce_partials");
    }
    if (extra) {                                            // scalar layout: the extra column as a launch of its own
        const int st = launch_reduce(ws, split_stride, splits, ldw, out, ldo, rows, cols, accumulate, s);
        return st != LIME_OK ? st : launch_reduce(ws + cols, split_stride, splits, ldw, extra, 1, rows, 1, accumulate, s);
    }
    const int grid = (int)((total + 63) / 64);
    reduce_partials_kernel<<<grid, 256, 0, s>>>(ws, split_stride, splits, ldw, out, ldo, rows, cols, accumulate);
    return lime_check_launch("reduce_partials");
}

int colsum_blocks(int M) {
    int b = (M + 255) / 256;
    return b < 1 ? 1 : (b > 256 ? 256 : b);
}

int ln_blocks(int M) { const int b = (M + 15) / 16; return b > 768 ? 768 : b; }   // persistent: 3 workgroups per CU

}  // namespace

extern "C" int lime_colsum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
                               float* workspace, int64_t workspace_floats, void* stream);

// The split-product kernel (wgrad_sp_f32.hip) takes the problems that fill its 256 x 320 tiles: from 4096 rows on (below, the
// workgroups' slices are a handful of chunks), 16-byte friendly operands, at least half of the padded tile grid real.
static bool wgrad_sp_shape(int M, int N, int K) { return M >= 4096 && N % 4 == 0 && K % 4 == 0 && N >= 64 && K >= 64 && lime_wgrad_sp_plan(M, N, K).fill >= 0.5; }

extern "C" int64_t lime_linear_wgrad_workspace(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const WgradPlan w = wgrad_plan(M, N, K);
    int64_t need = (int64_t)w.splits * w.np * w.kp;
    if (wgrad_sp_shape(M, N, K)) {                                               // whichever kernel the call ends up on
        const LimeWgradSpPlan sp = lime_wgrad_sp_plan(M, N, K);
        const int64_t need_sp = (int64_t)sp.splits * sp.np * sp.kp;
        if (need_sp > need) need = need_sp;
    }
    return need + (int64_t)colsum_blocks(M) * N;                                  // + the column-sum fallback of db
}

extern "C" int lime_linear_wgrad_f32(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw, int64_t lddw,
                                     float* db, int32_t M, int32_t N, int32_t K, int32_t accumulate, float* workspace,
                                     int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(dy && x && dw && workspace, LIME_ERR_BAD_ARG, "lime_linear_wgrad_f32: null pointer");
    LIME_REQUIRE(M > 0 && N > 0 && K > 0, LIME_ERR_BAD_ARG, "lime_linear_wgrad_f32: non-positive dimension");
    LIME_REQUIRE(ldy >= N && ldx >= K && lddw >= K, LIME_ERR_BAD_ARG, "lime_linear_wgrad_f32: leading dimension smaller than the row");
    const bool vec = N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15) == 0;
    static const bool no_dma = getenv("LIME_WGRAD_NO_DMA") != nullptr;        // A/B switch for tools/, not a product option
    const WgradPlan w = wgrad_plan(M, N, K, 1);
    LIME_REQUIRE(workspace_floats >= lime_linear_wgrad_workspace(M, N, K), LIME_ERR_BAD_ARG,
                 "lime_linear_wgrad_f32: workspace holds %ld floats, lime_linear_wgrad_workspace() asks for %ld",
                 (long)workspace_floats, (long)lime_linear_wgrad_workspace(M, N, K));
    hipStream_t s = (hipStream_t)stream;
    LIME_REQUIRE(((long)w.rows_per_split + WG_MC) * (ldy > ldx ? ldy : ldx) * 4 < 0x7FFFFFF0L, LIME_ERR_UNSUPPORTED,
                 "lime_linear_wgrad_f32: a row slice spans more than 2 GB (rows %d, ld %ld)", w.rows_per_split, (long)(ldy > ldx ? ldy : ldx));
    int st;
    static const bool no_sp = getenv("LIME_WGRAD_NO_SP") != nullptr;          // A/B switch for tools/
    if (vec && !no_sp && (lime_split_mode() & 1) && wgrad_sp_shape(M, N, K)) {
        const LimeWgradSpPlan sp = lime_wgrad_sp_plan(M, N, K);
        LIME_REQUIRE(((long)sp.rows_per_split + 32) * (ldy > ldx ? ldy : ldx) * 4 < 0x7FFFFFF0L, LIME_ERR_UNSUPPORTED,
                     "lime_linear_wgrad_f32: a row slice spans more than 2 GB (rows %d, ld %ld)", sp.rows_per_split, (long)(ldy > ldx ? ldy : ldx));
        const int ones = (!sp.swap && db != nullptr && K < sp.kp) ? 1 : 0;
        st = lime_wgrad_sp_launch(sp, dy, ldy, x, ldx, workspace, M, N, K, ones, s);
        if (st != LIME_OK) return st;
        const int64_t used = (int64_t)sp.splits * sp.np * sp.kp;
        if (sp.swap) st = lime_wgrad_sp_reduce_t(sp, workspace, dw, lddw, N, K, accumulate, s);
        else st = launch_reduce(workspace, sp.np * sp.kp, sp.splits, sp.kp, dw, lddw, N, K, accumulate, s, ones ? db : nullptr);
        if (st != LIME_OK || db == nullptr || ones) return st;
        return lime_colsum_f32(dy, ldy, M, N, db, accumulate, workspace + used, workspace_floats - used, stream);
    }
    const int ones_col = (db != nullptr && K < w.kp) ? 1 : 0;          // room for a ones column in the padded tile grid
#define WGRAD(NKT) (vec ? (no_dma ? launch_wgrad<NKT, true>(w, dy, ldy, x, ldx, workspace, M, N, K, ones_col, s)              \
                                  : launch_wgrad_dma<NKT>(w, dy, ldy, x, ldx, workspace, M, N, K, ones_col, s))                \
                        : launch_wgrad<NKT, false>(w, dy, ldy, x, ldx, workspace, M, N, K, ones_col, s))
    if (w.nkt == 5) st = WGRAD(5); else if (w.nkt == 3) st = WGRAD(3); else st = WGRAD(4);
#undef WGRAD
    if (st != LIME_OK) return st;
    st = launch_reduce(workspace, w.np * w.kp, w.splits, w.kp, dw, lddw, N, K, accumulate, s, ones_col ? db : nullptr);   // db: column K of the partial tiles
    if (st != LIME_OK || db == nullptr || ones_col) return st;
    float* cws = workspace + (int64_t)w.splits * w.np * w.kp;             // K fills its tiles: a separate column-sum pass
    return lime_colsum_f32(dy, ldy, M, N, db, accumulate, cws, workspace_floats - (int64_t)w.splits * w.np * w.kp, stream);
}

extern "C" int64_t lime_colsum_workspace(int32_t M, int32_t N) {
    return M > 0 && N > 0 ? (int64_t)colsum_blocks(M) * N : 0;
}

extern "C" int lime_colsum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
                               float* workspace, int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(x && out && workspace, LIME_ERR_BAD_ARG, "lime_colsum_f32: null pointer");
    LIME_REQUIRE(M > 0 && N > 0 && ldx >= N, LIME_ERR_BAD_ARG, "lime_colsum_f32: bad dimensions");
    const int nblk = colsum_blocks(M);
    LIME_REQUIRE(workspace_floats >= (int64_t)nblk * N, LIME_ERR_BAD_ARG, "lime_colsum_f32: workspace too small (%ld < %ld)",
                 (long)workspace_floats, (long)nblk * N);
    hipStream_t s = (hipStream_t)stream;
    const int rpb = (M + nblk - 1) / nblk;
    colsum_kernel<<<dim3((N + 63) / 64, nblk), 256, 0, s>>>(x, ldx, M, N, rpb, workspace);
    const int st = lime_check_launch("colsum_kernel");
    if (st != LIME_OK) return st;
    return launch_reduce(workspace, N, nblk, N, out, N, 1, N, accumulate, s);
}

extern "C" int64_t lime_layernorm_bwd_workspace(int32_t M, int32_t E) {
    if (M <= 0 || E <= 0) return 0;
    return (int64_t)ln_blocks(M) * 3 * E;
}

static int layernorm_bwd(const float* dy, int64_t lddy, int32_t dy_div, float dy_scale, const float* y, int64_t ldy,
                         const float* gamma, const float* beta, const float* rstd, float* dz, int64_t lddz,
                         int32_t M, int32_t E, float* dgamma, float* dbeta, float* dzsum, int32_t accumulate,
                         float* workspace, int64_t workspace_floats, float* dz_drop, int64_t lddd, const LimeDropout& drop, void* stream) {
    LIME_REQUIRE(dy && y && gamma && beta && rstd && dz && workspace, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_f32: null pointer");
    LIME_REQUIRE(M > 0 && E > 0 && dy_div >= 1, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_f32: bad dimensions");
    LIME_REQUIRE(E <= 512, LIME_ERR_UNSUPPORTED, "lime_layernorm_bwd_f32: E = %d > 512", E);
    LIME_REQUIRE(lddy >= E && ldy >= E && lddz >= E, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_f32: leading dimension smaller than E");
    const int nblk = ln_blocks(M);
    LIME_REQUIRE(workspace_floats >= (int64_t)nblk * 3 * E, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_f32: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = E % 4 == 0 && lddy % 4 == 0 && ldy % 4 == 0 && lddz % 4 == 0 &&
                     ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dz) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
    LIME_REQUIRE(dz_drop == nullptr || (vec && lddd % 4 == 0 && (((uintptr_t)dz_drop) & 15) == 0 && lddd >= E), LIME_ERR_UNSUPPORTED,
                 "lime_layernorm_bwd_dropout_f32: the dropped copy needs 16-byte friendly operands (E, leading dimensions multiples of 4)");
    if (vec) {
        const int v4 = (E + 63) / 64;
#define LN_BWD_V(C) layernorm_bwd_vec_kernel<C><<<nblk, 256, 0, s>>>(dy, lddy, dy_div, dy_scale, y, ldy, gamma, beta, rstd, dz, lddz, M, E, workspace, dz_drop, lddd, drop)
        if (v4 <= 2) LN_BWD_V(2); else if (v4 <= 5) LN_BWD_V(5); else LN_BWD_V(8);
#undef LN_BWD_V
    } else {
        const int cpl = (E + 63) / 64;
#define LN_BWD(C) layernorm_bwd_kernel<C><<<nblk, 256, 0, s>>>(dy, lddy, dy_div, dy_scale, y, ldy, gamma, beta, rstd, dz, lddz, M, E, workspace)
        if (cpl <= 2) LN_BWD(2); else if (cpl <= 5) LN_BWD(5); else LN_BWD(8);
#undef LN_BWD
    }
    int st = lime_check_launch("layernorm_bwd_kernel");
    if (st != LIME_OK) return st;
    if (E % 4 == 0 && ((((uintptr_t)dgamma) | ((uintptr_t)dbeta) | ((uintptr_t)dzsum) | ((uintptr_t)workspace)) & 15) == 0) {
        reduce_ln3_kernel<<<(3 * E / 4 + 15) / 16, 256, 0, s>>>(workspace, nblk, E, dgamma, dbeta, dzsum, accumulate);
        return lime_check_launch("reduce_ln3_kernel");
    }
    float* outs[3] = {dgamma, dbeta, dzsum};
    for (int k = 0; k < 3; ++k) {
        if (!outs[k]) continue;
        st = launch_reduce(workspace + (long)k * E, 3L * E, nblk, E, outs[k], E, 1, E, accumulate, s);
        if (st != LIME_OK) return st;
    }
    return LIME_OK;
}

extern "C" int lime_layernorm_bwd_f32(const float* dy, int64_t lddy, int32_t dy_div, float dy_scale, const float* y, int64_t ldy,
                                      const float* gamma, const float* beta, const float* rstd, float* dz, int64_t lddz,
                                      int32_t M, int32_t E, float* dgamma, float* dbeta, float* dzsum, int32_t accumulate,
                                      float* workspace, int64_t workspace_floats, void* stream) {
    return layernorm_bwd(dy, lddy, dy_div, dy_scale, y, ldy, gamma, beta, rstd, dz, lddz, M, E, dgamma, dbeta, dzsum, accumulate, workspace,
                         workspace_floats, nullptr, 0, lime_make_dropout(0.f, 0, 0), stream);
}

extern "C" int lime_layernorm_bwd_dropout_f32(const float* dy, int64_t lddy, int32_t dy_div, float dy_scale, const float* y, int64_t ldy,
                                              const float* gamma, const float* beta, const float* rstd, float* dz, int64_t lddz,
                                              int32_t M, int32_t E, float* dgamma, float* dbeta, float* dzsum, int32_t accumulate,
                                              float* workspace, int64_t workspace_floats, float* dz_drop, int64_t lddd, float dropout_p,
                                              uint64_t seed, uint32_t site, void* stream) {
    LIME_REQUIRE(dz_drop != nullptr, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_dropout_f32: dz_drop is NULL");
    LIME_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, LIME_ERR_BAD_ARG, "lime_layernorm_bwd_dropout_f32: dropout_p outside [0, 1)");
    return layernorm_bwd(dy, lddy, dy_div, dy_scale, y, ldy, gamma, beta, rstd, dz, lddz, M, E, dgamma, dbeta, dzsum, accumulate, workspace,
                         workspace_floats, dz_drop, lddd, lime_make_dropout(dropout_p, seed, site), stream);
}

extern "C" int lime_relu_bwd_f32(float* dh, int64_t lddh, const float* h, int64_t ldh, int64_t rows, int32_t cols, float scale,
                                 void* stream) {
    LIME_REQUIRE(dh && h, LIME_ERR_BAD_ARG, "lime_relu_bwd_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && cols > 0 && lddh >= cols && ldh >= cols, LIME_ERR_BAD_ARG, "lime_relu_bwd_f32: bad dimensions");
    if (rows == 0) return LIME_OK;
    const long total = rows * cols;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    relu_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dh, lddh, h, ldh, rows, cols, scale);
    return lime_check_launch("relu_bwd_kernel");
}

namespace {
template <int SP>
int launch_attn_bwd(const float* q, const float* k, const float* v, long ld, const float* dout, long ldo, float* dq, float* dk,
                    float* dv, long ldd, int n_seq, int S, int n_head, int head_dim, int head_stride, float scale,
                    const LimeDropout& drop, const unsigned char* key_mask, hipStream_t s) {
    constexpr int PPW = 8 / (SP / 16);
    constexpr int BYTES = PPW * (4 * SP * AB_LD + SP * (SP + 2)) * 4;
    static bool configured = false;
    static int n_cu = 256;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)token_attn_bwd_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_bwd_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n_cu = cus;
        configured = true;
    }
    const long n_group = ((long)n_seq * n_head + PPW - 1) / PPW;
    const int grid = (int)(n_group < n_cu ? n_group : n_cu);             // persistent: one workgroup per CU
    // vector staging: 32-float head rows on 16-byte boundaries with zero padding columns, dO pairs on 8-byte boundaries
    const bool vec = head_stride == 32 && ld % 4 == 0 && ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0 &&
                     head_dim % 2 == 0 && ldo % 2 == 0 && (((uintptr_t)dout) & 7) == 0;
    token_attn_bwd_kernel<SP><<<grid, 512, BYTES, s>>>(q, k, v, ld, dout, ldo, dq, dk, dv, ldd, n_seq, S, n_head, head_dim,
                                                      head_stride, scale, vec ? 1 : 0, drop, key_mask);
    return lime_check_launch("token_attn_bwd_kernel");
}

template <int SP>
int launch_attn_fwd_dropout(const float* q, const float* k, const float* v, long ld, float* out, long ldo, int n_seq, int S,
                            int n_head, int head_dim, int head_stride, float scale, const LimeDropout& drop, hipStream_t s) {
    constexpr int PPW = 8 / (SP / 16);
    constexpr int BYTES = PPW * (2 * SP * AB_LD) * 4;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)token_attn_fwd_dropout_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_dropout_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        configured = true;
    }
    const long n_group = ((long)n_seq * n_head + PPW - 1) / PPW;
    const bool vec = head_stride == 32 && ld % 4 == 0 && ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0;
    token_attn_fwd_dropout_kernel<SP><<<(unsigned)n_group, 512, BYTES, s>>>(q, k, v, ld, out, ldo, n_seq, S, n_head, head_dim, head_stride,
                                                                           scale, drop, vec ? 1 : 0);
    return lime_check_launch("token_attn_fwd_dropout_kernel");
}
}  // namespace

extern "C" int64_t lime_token_attention_stats_workspace(int32_t n_seq, int32_t S, int32_t n_head) {
    return S > 128 ? (int64_t)n_seq * S * n_head * 2 : 0;            // lse and delta per (token, head) for the blocked paths
}

extern "C" int64_t lime_token_attention_bwd_workspace(int32_t n_seq, int32_t S, int32_t n_head) {
    if (S <= 128) return 0;
    const int64_t n_blk = (S + LB - 1) / LB;
    // the row statistics + one [tokens][n_head * 32] slab per key block behind the first (their shares of dq, summed in block order)
    return lime_token_attention_stats_workspace(n_seq, S, n_head) + (n_blk - 1) * (int64_t)n_seq * S * n_head * 32;
}

static int attention_bwd(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out,
                         int64_t ld_out, const float* dout, int64_t ldo, float* dq, float* dk, float* dv,
                         int64_t ld_dqkv, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim,
                         int32_t head_stride, float scale, float* workspace, int64_t workspace_floats,
                         float dropout_p, uint64_t seed, uint32_t site, const uint8_t* key_mask, const float* lse, void* stream) {
    LIME_REQUIRE(q && k && v && dout && dq && dk && dv, LIME_ERR_BAD_ARG, "lime_token_attention_bwd_f32: null pointer");
    LIME_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, LIME_ERR_BAD_ARG, "lime_token_attention_bwd_f32: dropout_p outside [0, 1)");
    const LimeDropout drop = lime_make_dropout(dropout_p, seed, site);
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0, LIME_ERR_BAD_ARG, "lime_token_attention_bwd_f32: bad dimensions");
    LIME_REQUIRE(S <= 512 && head_dim <= 32 && head_stride >= head_dim && head_stride <= 32, LIME_ERR_UNSUPPORTED,
                 "lime_token_attention_bwd_f32: needs S <= 512 and head_dim <= head_stride <= 32 (S=%d head_dim=%d head_stride=%d)",
                 S, head_dim, head_stride);
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * head_stride && ld_dqkv >= (int64_t)n_head * head_stride && ldo >= (int64_t)n_head * head_dim,
                 LIME_ERR_BAD_ARG, "lime_token_attention_bwd_f32: leading dimension smaller than the row");
    if (n_seq == 0) return LIME_OK;
    hipStream_t s = (hipStream_t)stream;
    if (S <= 32) return launch_attn_bwd<32>(q, k, v, ld_qkv, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride, scale, drop, key_mask, s);
    if (S <= 64) return launch_attn_bwd<64>(q, k, v, ld_qkv, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride, scale, drop, key_mask, s);
    if (S <= 128) {
        if (key_mask == nullptr && (lime_split_mode() & 1)) {      // every product on the bf16 matrix cores (token_attn_bwd_sp_f32.hip)
            const int st = lime_token_attention_bwd_sp(q, k, v, ld_qkv, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride,
                                                       scale, drop, s);
            if (st != LIME_PP_NOT_APPLICABLE) return st;
        }
        return launch_attn_bwd<128>(q, k, v, ld_qkv, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride, scale, drop, key_mask, s);
    }
    // blocked path
    LIME_REQUIRE(key_mask == nullptr, LIME_ERR_UNSUPPORTED, "lime_token_attention_bwd_f32: a key mask needs S <= 128");
    LIME_REQUIRE(out && ld_out >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG,
                 "lime_token_attention_bwd_f32: S > 128 needs the forward output `out` (delta = dO . O)");
    LIME_REQUIRE(workspace && workspace_floats >= lime_token_attention_bwd_workspace(n_seq, S, n_head), LIME_ERR_BAD_ARG,
                 "lime_token_attention_bwd_f32: S > 128 needs lime_token_attention_bwd_workspace() floats of workspace");
    const int n_blk = (S + LB - 1) / LB;
    const long n_prob = (long)n_seq * n_head;
    LIME_REQUIRE(n_prob * n_blk < 0x7FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_token_attention_bwd_f32: too many blocks");
    hipError_t e = hipSuccess;
    float* const dq_slabs = workspace + lime_token_attention_stats_workspace(n_seq, S, n_head);
    const long n_tok = (long)n_seq * S;
    const bool spx = (lime_split_mode() & 1) != 0;             // Q K^T / dO V^T as split products on the bf16 matrix cores
    if (lse) {                                                 // the forward kept its log-sum-exp: only delta = dO . O is left
        LIME_REQUIRE(n_head <= 64 && n_head * head_dim <= 1024, LIME_ERR_UNSUPPORTED, "lime_token_attention_bwd_lse_f32: n_head > 64 or n_head * head_dim > 1024");
        attn_delta_kernel<<<(unsigned)((n_tok + 3) / 4 > 8192 ? 8192 : (n_tok + 3) / 4), 256, 0, s>>>(out, ld_out, dout, ldo, lse, workspace, n_tok,
                                                                                                    n_head, head_dim);
    } else if (spx) attn_stats_kernel<true><<<(unsigned)(n_prob * n_blk), 512, 0, s>>>(q, k, ld_qkv, out, ld_out, dout, ldo, workspace, S, n_head,
                                                                             head_dim, head_stride, scale, n_blk);
    else attn_stats_kernel<false><<<(unsigned)(n_prob * n_blk), 512, 0, s>>>(q, k, ld_qkv, out, ld_out, dout, ldo, workspace, S, n_head,
                                                                           head_dim, head_stride, scale, n_blk);
    int st = lime_check_launch("attn_stats_kernel");
    if (st != LIME_OK) return st;
    if (spx) {
        constexpr int BYTES_SP = (3 * LB * lime_dev::SWZ_ROW + 3 * LQ * lime_dev::SWZ_ROW + 2 * LQ) * 4;
        static_assert(2 * BYTES_SP <= 163840, "LDS budget: two workgroups per CU");
        static bool configured_sp = false;
        if (!configured_sp) {
            e = hipFuncSetAttribute((const void*)attn_bwd_long_sp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES_SP);
            LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_bwd_f32: cannot reserve %d bytes of LDS: %s", BYTES_SP,
                         hipGetErrorString(e));
            configured_sp = true;
        }
        attn_bwd_long_sp_kernel<<<(unsigned)(n_prob * n_blk), 256, BYTES_SP, s>>>(q, k, v, ld_qkv, dout, ldo, workspace, dq, dk, dv,
                                                                                 ld_dqkv, S, n_head, head_dim, head_stride, scale,
                                                                                 n_blk, drop, dq_slabs, n_tok);
        st = lime_check_launch("attn_bwd_long_sp_kernel");
        if (st != LIME_OK || n_blk == 1) return st;
        const long total_sp = n_tok * n_head * head_stride;
        attn_dq_reduce_kernel<<<(unsigned)((total_sp + 255) / 256 > 8192 ? 8192 : (total_sp + 255) / 256), 256, 0, s>>>(dq, ld_dqkv, dq_slabs, n_tok,
                                                                                                                  n_head, head_stride, n_blk - 1);
        return lime_check_launch("attn_dq_reduce_kernel");
    }
    constexpr int BYTES = (4 * LB * AB_LD + LB * (LB + 2) + 2 * LB) * 4;
    static bool configured = false;
    if (!configured) {
        e = hipFuncSetAttribute((const void*)attn_bwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_bwd_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        configured = true;
    }
    attn_bwd_long_kernel<<<(unsigned)(n_prob * n_blk), 512, BYTES, s>>>(q, k, v, ld_qkv, dout, ldo, workspace, dq, dk, dv, ld_dqkv, S,
                                                                       n_head, head_dim, head_stride, scale, n_blk, drop, dq_slabs, n_tok);
    st = lime_check_launch("attn_bwd_long_kernel");
    if (st != LIME_OK || n_blk == 1) return st;
    const long total = n_tok * n_head * head_stride;
    attn_dq_reduce_kernel<<<(unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256), 256, 0, s>>>(dq, ld_dqkv, dq_slabs, n_tok, n_head,
                                                                                                          head_stride, n_blk - 1);
    return lime_check_launch("attn_dq_reduce_kernel");
}

extern "C" int lime_token_attention_bwd_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out,
                                            int64_t ld_out, const float* dout, int64_t ldo, float* dq, float* dk, float* dv,
                                            int64_t ld_dqkv, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim,
                                            int32_t head_stride, float scale, float* workspace, int64_t workspace_floats,
                                            float dropout_p, uint64_t seed, uint32_t site, const uint8_t* key_mask, void* stream) {
    return attention_bwd(q, k, v, ld_qkv, out, ld_out, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride, scale,
                         workspace, workspace_floats, dropout_p, seed, site, key_mask, nullptr, stream);
}

extern "C" int lime_token_attention_bwd_lse_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out,
                                                int64_t ld_out, const float* lse, const float* dout, int64_t ldo, float* dq, float* dk,
                                                float* dv, int64_t ld_dqkv, int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim,
                                                int32_t head_stride, float scale, float* workspace, int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(lse != nullptr, LIME_ERR_BAD_ARG, "lime_token_attention_bwd_lse_f32: lse is NULL");
    return attention_bwd(q, k, v, ld_qkv, out, ld_out, dout, ldo, dq, dk, dv, ld_dqkv, n_seq, S, n_head, head_dim, head_stride, scale,
                         workspace, workspace_floats, 0.f, 0, 0, nullptr, lse, stream);
}

extern "C" int lime_token_attention_dropout_f32(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, int64_t ldo,
                                                int32_t n_seq, int32_t S, int32_t n_head, int32_t head_dim, int32_t head_stride,
                                                float scale, float dropout_p, uint64_t seed, uint32_t site, float* workspace,
                                                int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(q && k && v && out, LIME_ERR_BAD_ARG, "lime_token_attention_dropout_f32: null pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && n_head > 0 && head_dim > 0, LIME_ERR_BAD_ARG, "lime_token_attention_dropout_f32: bad dimensions");
    LIME_REQUIRE(S <= 512 && head_dim <= 32 && head_stride >= head_dim && head_stride <= 32, LIME_ERR_UNSUPPORTED,
                 "lime_token_attention_dropout_f32: needs S <= 512 and head_dim <= head_stride <= 32");
    LIME_REQUIRE(ld_qkv >= (int64_t)n_head * head_stride && ldo >= (int64_t)n_head * head_dim, LIME_ERR_BAD_ARG,
                 "lime_token_attention_dropout_f32: leading dimension smaller than the row");
    LIME_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, LIME_ERR_BAD_ARG, "lime_token_attention_dropout_f32: dropout_p outside [0, 1)");
    if (n_seq == 0) return LIME_OK;
    const LimeDropout drop = lime_make_dropout(dropout_p, seed, site);
    hipStream_t s = (hipStream_t)stream;
    if (head_stride == 32 && (S == 32 || S == 64 || S == 128)) {       // the split-product forward with the mask on its probability registers
        const int st = lime_token_attention_sp(q, k, v, (long)ld_qkv, nullptr, nullptr, out, (long)ldo, n_seq, S, n_head, head_dim, scale, nullptr, s, &drop);
        if (st != LIME_PP_NOT_APPLICABLE) return st;
    }
    if (S <= 32) return launch_attn_fwd_dropout<32>(q, k, v, ld_qkv, out, ldo, n_seq, S, n_head, head_dim, head_stride, scale, drop, s);
    if (S <= 64) return launch_attn_fwd_dropout<64>(q, k, v, ld_qkv, out, ldo, n_seq, S, n_head, head_dim, head_stride, scale, drop, s);
    if (S <= 128) return launch_attn_fwd_dropout<128>(q, k, v, ld_qkv, out, ldo, n_seq, S, n_head, head_dim, head_stride, scale, drop, s);
    LIME_REQUIRE(workspace && workspace_floats >= lime_token_attention_stats_workspace(n_seq, S, n_head), LIME_ERR_BAD_ARG,
                 "lime_token_attention_dropout_f32: S > 128 needs lime_token_attention_bwd_workspace() floats of workspace");
    const int n_blk = (S + LB - 1) / LB;
    const long n_prob = (long)n_seq * n_head;
    LIME_REQUIRE(n_prob * n_blk < 0x7FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_token_attention_dropout_f32: too many blocks");
    if (lime_split_mode() & 1)
        attn_stats_kernel<true><<<(unsigned)(n_prob * n_blk), 512, 0, s>>>(q, k, ld_qkv, nullptr, 0, nullptr, 0, workspace, S, n_head, head_dim,
                                                                        head_stride, scale, n_blk);
    else
        attn_stats_kernel<false><<<(unsigned)(n_prob * n_blk), 512, 0, s>>>(q, k, ld_qkv, nullptr, 0, nullptr, 0, workspace, S, n_head, head_dim,
                                                                         head_stride, scale, n_blk);
    int st = lime_check_launch("attn_stats_kernel");
    if (st != LIME_OK) return st;
    constexpr int BYTES = (3 * LB * AB_LD + LB * (LB + 2)) * 4;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_long_dropout_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "lime_token_attention_dropout_f32: cannot reserve %d bytes of LDS: %s", BYTES,
                     hipGetErrorString(e));
        configured = true;
    }
    attn_fwd_long_dropout_kernel<<<(unsigned)(n_prob * n_blk), 512, BYTES, s>>>(q, k, v, ld_qkv, workspace, out, ldo, S, n_head, head_dim,
                                                                               head_stride, scale, n_blk, drop);
    return lime_check_launch("attn_fwd_long_dropout_kernel");
}

extern "C" int lime_embed_bwd_f32(const int32_t* ids, const float* dx, int64_t lddx, float* dtable, int64_t ld_table, int64_t rows,
                                  int32_t dim, int32_t hot_id, void* stream) {
    LIME_REQUIRE(ids && dx && dtable, LIME_ERR_BAD_ARG, "lime_embed_bwd_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && lddx >= dim && ld_table >= dim, LIME_ERR_BAD_ARG, "lime_embed_bwd_f32: bad dimensions");
    LIME_REQUIRE(dim <= 512, LIME_ERR_UNSUPPORTED, "lime_embed_bwd_f32: dim = %d > 512", dim);
    if (rows == 0) return LIME_OK;
    const int rpb = 512;
    const int grid = (int)((rows + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    if (dim <= 320) embed_bwd_kernel<5><<<grid, 256, 0, s>>>(ids, dx, lddx, dtable, ld_table, rows, dim, hot_id, rpb);
    else embed_bwd_kernel<8><<<grid, 256, 0, s>>>(ids, dx, lddx, dtable, ld_table, rows, dim, hot_id, rpb);
    return lime_check_launch("embed_bwd_kernel");
}

extern "C" int lime_embed_bwd_small_f32(const int32_t* ids, const float* dx, int64_t lddx, float* dtable, int64_t ld_table,
                                        int64_t rows, int32_t dim, int32_t table_rows, void* stream) {
    LIME_REQUIRE(ids && dx && dtable, LIME_ERR_BAD_ARG, "lime_embed_bwd_small_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && lddx >= dim && ld_table >= dim, LIME_ERR_BAD_ARG, "lime_embed_bwd_small_f32: bad dimensions");
    LIME_REQUIRE(table_rows >= 1 && table_rows <= 32, LIME_ERR_UNSUPPORTED, "lime_embed_bwd_small_f32: table_rows = %d outside [1, 32]", table_rows);
    if (rows == 0) return LIME_OK;
    embed_bwd_small_kernel<<<(dim + 63) / 64, 512, 0, (hipStream_t)stream>>>(ids, dx, lddx, dtable, ld_table, rows, dim, table_rows);
    return lime_check_launch("embed_bwd_small_kernel");
}

extern "C" int lime_grad_clip_coef_f32(const float* g, int64_t n, float max_norm, float* out2, float* workspace,
                                       int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(g && out2 && workspace, LIME_ERR_BAD_ARG, "lime_grad_clip_coef_f32: null pointer");
    LIME_REQUIRE(n > 0 && workspace_floats >= 1024, LIME_ERR_BAD_ARG, "lime_grad_clip_coef_f32: n <= 0 or workspace < 1024 floats");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    sumsq_kernel<<<grid, 256, 0, s>>>(g, n, workspace);
    int st = lime_check_launch("sumsq_kernel");
    if (st != LIME_OK) return st;
    clip_coef_kernel<<<1, 256, 0, s>>>(workspace, grid, max_norm, out2);
    return lime_check_launch("clip_coef_kernel");
}

extern "C" int lime_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, const float* grad_scale, void* stream) {
    LIME_REQUIRE(p && g && m && v, LIME_ERR_BAD_ARG, "lime_adam_f32: null pointer");
    LIME_REQUIRE(n >= 0 && step >= 1, LIME_ERR_BAD_ARG, "lime_adam_f32: n < 0 or step < 1");
    if (n == 0) return LIME_OK;
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = sqrt(1.0 - pow((double)beta2, (double)step));
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    adam_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)b1, (float)b2, grad_scale);
    return lime_check_launch("adam_kernel");
}

extern "C" int lime_nll_softmax_f32(const float* logits, int64_t ld, int32_t B, int32_t K, float* loss, float* dlogits, int64_t ldd,
                                    void* stream) {
    LIME_REQUIRE(logits && loss, LIME_ERR_BAD_ARG, "lime_nll_softmax_f32: null pointer");
    LIME_REQUIRE(B > 0 && K > 0 && ld >= K && (!dlogits || ldd >= K), LIME_ERR_BAD_ARG, "lime_nll_softmax_f32: bad dimensions");
    nll_softmax_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, ld, B, K, loss, dlogits, ldd);
    return lime_check_launch("nll_softmax_kernel");
}
