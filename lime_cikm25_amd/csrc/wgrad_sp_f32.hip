// lime_linear_wgrad_f32, big-M problems on the bf16 matrix cores with fp32-level arithmetic: dW[n, k] = sum_m dY[m, n] X[m, k]
// with the split product of gemm_sp_f32.hip (each fp32 operand = three bf16 terms, six v_mfma_f32_16x16x32_bf16 per block, fp32
// accumulation: the error of one fp32 rounding per product, as the fp32 MFMA commits).
//
// The reduction index of a weight gradient is the TOKEN index m, and both operands are stored token-major ([M, N] and [M, K]): a
// lane's MFMA fragment -- eight consecutive m of one column -- is a strided read.  The 32-token chunks go global -> LDS by LDS-DMA
// exactly as they lie in memory ([m][column], fp32), two stages; a fragment is eight ds_read_b32 down a column (hipcc pairs them
// into ds_read2st64_b32) and is split into its bf16 terms in registers.  Groups of eight chunk rows (one lane group kg each) are
// 64 bytes apart modulo the bank width, so the four lane groups of a fragment read hit disjoint banks.
//
// One eight-wave workgroup per CU: tile 256 (n) x 320 (k) of dW over one slice of the M rows (the slices' partial tiles are summed
// in a fixed order by reduce_partials_kernel: backward_f32.hip); wave w owns rows 64 (w & 3) of n and the k half w >> 2
// (4 x 10 accumulator tiles, 160 registers), as in gemm_sp_f32.hip.  A ones column appended to X (column K of the tile, patched
// into the LDS image) makes column K of dW the bias gradient.  Shapes whose K pads badly (K = 512 against N = 300) are run
// transposed by the dispatcher (the roles of dY and X swapped, the reduction writes dW^T back transposed).
#include "common.h"
#include "gemm_pp.h"
#include "lds_dma.h"
#include "split_mfma.h"

using namespace lime_dev;

namespace {

constexpr int TN = 256, TK = 320, MC = 32;
constexpr int A_GRP = 8 * TN + 16;                 // floats per group of eight chunk rows (+ 64 bytes: the bank offset of the next group)
constexpr int B_GRP = 8 * TK + 16;
constexpr int A_FLOATS = 4 * A_GRP, B_FLOATS = 4 * B_GRP, STAGE = A_FLOATS + B_FLOATS;
constexpr int CT = TK / 32;                        // 16-column accumulator tiles per wave along k: 10
static_assert(2 * STAGE * 4 <= 163840, "LDS budget");

using Split = SplitFrag;                            // split_mfma.h
__device__ __forceinline__ Split split8(const float (&x)[8]) { return split_frag(x); }
__device__ __forceinline__ f32x4 mfma6(const Split& w, const Split& a, f32x4 c) { return split_mfma16(w, a, c); }
// eight floats down a column of the [m][COLS] chunk image: rows 8 kg .. 8 kg + 7 of the lane's group
template <int COLS>
__device__ __forceinline__ void column8(const float* p, float (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[j * COLS];
}

__global__ __launch_bounds__(512, 2) void wgrad_sp_kernel(const float* __restrict__ dy, long ldy, const float* __restrict__ x,
                                                           long ldx, float* __restrict__ ws, int M, int N, int K, int n_tiles,
                                                           int k_tiles, int rows_per_split, int ones_col) {
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];      // ONE __shared__ object (see gemm_sp_f32.hip)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;
    const int fi = lane & 15, kg = lane >> 4;
    const int ntile = n_tiles * k_tiles;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile = logical - split * ntile;
    const int n0 = (tile / k_tiles) * TN, k0 = (tile % k_tiles) * TK;
    const long m_begin = (long)split * rows_per_split;
    const long m_end_l = m_begin + rows_per_split;
    const int rows_here = (int)((m_end_l < (long)M ? m_end_l : (long)M) - m_begin);
    const int ldy4 = (int)ldy * 4, ldx4 = (int)ldx * 4;

    f32x4 acc[4][CT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's share of the tile: row tiles / column tiles that hold real rows of dW (the ones column counts as a column)
    const int n_left = N - n0 - 64 * wr, k_left = K + (ones_col ? 1 : 0) - k0 - 16 * CT * wc;
    const bool active = n_left > 0 && k_left > 0;
    const int nct = __builtin_amdgcn_readfirstlane(k_left <= 0 ? 0 : (k_left >= 16 * CT ? CT : (k_left + 15) >> 4));

    if (rows_here > 0) {
        // Descriptors end with the last row of the slice: rows beyond it are out of range and land as zeros.  Everything that varies
        // goes into the VECTOR offset (the range check's operand); columns beyond N / K carry OOB from the start.
        const int a_cols = (N - n0 < TN ? N - n0 : TN), b_cols = (K - k0 < TK ? K - k0 : TK);
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(dy + m_begin * ldy + n0), 0, (rows_here - 1) * ldy4 + a_cols * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + m_begin * ldx + k0), 0, (rows_here - 1) * ldx4 + b_cols * 4, 0x00020000);
        // A pieces: chunk row q = wave + 8 j (j < 4), lane l = columns 4 l .. 4 l + 3.  B pieces: qb = wave + 8 j (j < 5) of the 40:
        // group g = qb / 10, floats [256 (qb % 10) + 4 l, + 4) of the group's packed [8][320] image.
        // (the per-lane offsets are recomputed per chunk -- ~40 VALU instructions against ~800 of splitting: five more live registers
        // put six in scratch)
        auto issue = [&](int stage, int m0) {
            float* const sb = lds + stage * STAGE;
            int ln = lane;
            asm volatile("" : "+v"(ln));               // (opaque: hipcc otherwise hoists the offsets out of the chunk loop and spills them)
            const unsigned a_base = (4 * ln < a_cols) ? (unsigned)ln * 16u : OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wave + 8 * j;
                dma16(rs_a, reinterpret_cast<unsigned char*>(sb + j * A_GRP + wave * TN), a_base + (unsigned)((m0 + r) * ldy4), 0);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int qb = wave + 8 * j, g = qb / 10, f = 256 * (qb - 10 * g) + 4 * ln, rr = f / TK, c = f - rr * TK;
                const unsigned off = (c < b_cols) ? (unsigned)(m0 + 8 * g + rr) * (unsigned)ldx4 + (unsigned)c * 4u : OOB;
                dma16(rs_b, reinterpret_cast<unsigned char*>(sb + A_FLOATS + g * B_GRP + 256 * (qb - 10 * g)), off, 0);
            }
        };
        // the ones column of X: element K - k0 of every chunk row, written by the lane whose DMA piece covers it (after it landed)
        const int c1 = K - k0;
        auto patch_ones = [&](int stage, int m0) {
            float* const sb = lds + stage * STAGE + A_FLOATS;
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int qb = wave + 8 * j, g = qb / 10, f = 256 * (qb - 10 * g) + 4 * ln, rr = f / TK, c = f - rr * TK;
                if (c == c1) sb[g * B_GRP + f] = (m0 + 8 * g + rr < rows_here) ? 1.0f : 0.f;
            }
        };

        const float* const a_rd = lds + kg * A_GRP + 64 * wr + fi;
        const float* const b_rd = lds + A_FLOATS + kg * B_GRP + 16 * CT * wc + fi;
        auto compute = [&](int stage) {
            const float* const ap = a_rd + stage * STAGE;
            const float* const bp = b_rd + stage * STAGE;
            float r[8], xa[2][8];                      // activation fragments two row tiles ahead (all four at once: 11 registers in scratch)
            column8<TK>(bp, r);
            column8<TN>(ap, xa[0]);
            column8<TN>(ap + 16, xa[1]);
            Split a[4];
            Split w = split8(r);
            if (CT > 1) column8<TK>(bp + 16, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // the first column tile's MFMAs go out behind each activation split
                a[i] = split8(xa[i & 1]);
                if (i + 2 < 4) column8<TN>(ap + 16 * (i + 2), xa[i & 1]);
                if (0 < nct) acc[i][0] = mfma6(w, a[i], acc[i][0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (CT > 1) w = split8(r);
#pragma unroll
            for (int t = 1; t < CT; ++t) {
                Split wn = w;
                if (t < nct) {                         // one basic block: the next fragment's reads, this tile's MFMAs, the next split
                    if (t + 1 < CT) column8<TK>(bp + 16 * (t + 1), r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][t] = mfma6(w, a[i], acc[i][t]);
                    if (t + 1 < CT) wn = split8(r);
                }
                __builtin_amdgcn_sched_barrier(0);
                w = wn;
            }
        };

        issue(0, 0);
        int stage = 0;
        for (int m0 = 0; m0 < rows_here; m0 += MC) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of chunk m0 have landed
            if (ones_col && c1 >= 0 && c1 < TK) patch_ones(stage, m0);
            lds_barrier();                                             // everybody's have; the other stage is no longer read
            if (m0 + MC < rows_here) issue(stage ^ 1, m0 + MC);
            if (active) compute(stage);
            __builtin_amdgcn_sched_barrier(0);
            stage ^= 1;
        }
    }
    // partial tile -> ws[split][n][k] over the padded [n_tiles * 256, k_tiles * 320] grid: lane (fi, kg) holds row 16 i + fi of the
    // wave's 64 rows of n, columns 16 t + 4 kg + r of its k half
    const long ldw = (long)k_tiles * TK;
    float* const o = ws + (long)split * ((long)n_tiles * TN) * ldw + (long)(n0 + 64 * wr + fi) * ldw + k0 + 16 * CT * wc + 4 * kg;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) *reinterpret_cast<f32x4*>(o + (long)16 * i * ldw + 16 * t) = acc[i][t];
}

// dW^T partials -> dW: out[n * ldo + k] (+)= sum_s ws[s][k * ldw + n]  (n < rows, k < cols; n is the fast index of ws)
__global__ __launch_bounds__(256) void reduce_partials_t_kernel(const float* __restrict__ ws, long split_stride, int splits, long ldw,
                                                                 float* __restrict__ out, long ldo, int rows, int cols, int accumulate) {
    const long total = (long)rows * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e / rows), n = (int)(e - (long)k * rows);
        const float* p = ws + (long)k * ldw + n;
        float t = 0.f;
        int s = 0;
        for (; s + 8 <= splits; s += 8) {                  // eight independent loads in flight, summed in split order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(s + u) * split_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; s < splits; ++s) t += p[(long)s * split_stride];
        float* const q = out + (long)n * ldo + k;
        *q = accumulate ? *q + t : t;
    }
}

inline long pad_to(long v, long t) { return (v + t - 1) / t * t; }

}  // namespace

LimeWgradSpPlan lime_wgrad_sp_plan(int M, int N, int K) {
    LimeWgradSpPlan w{};
    // dW or dW^T: whichever pads the 256 x 320 tile grid less
    const long direct = pad_to(N, TN) * pad_to(K, TK), swapped = pad_to(K, TN) * pad_to(N, TK);
    w.swap = swapped < direct;
    const int n = w.swap ? K : N, k = w.swap ? N : K;
    w.n_tiles = (n + TN - 1) / TN;
    w.k_tiles = (k + TK - 1) / TK;
    w.np = (long)w.n_tiles * TN;
    w.kp = (long)w.k_tiles * TK;
    w.fill = (double)N * K / (double)(w.np * w.kp);
    const int ntile = w.n_tiles * w.k_tiles;
    int splits = 256 / ntile;                                      // one round of eight-wave workgroups, one per CU
    const int max_splits = (M + 255) / 256;                        // at least 8 chunks per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rps = (M + splits - 1) / splits;
    rps = (rps + MC - 1) / MC * MC;
    w.rows_per_split = rps;
    w.splits = (M + rps - 1) / rps;
    if (w.splits < 1) w.splits = 1;
    return w;
}

// Launches the split-product weight gradient into the workspace; the caller sums the partial tiles (direct layout:
// reduce_partials_kernel over [np, kp]; swapped: lime_wgrad_sp_reduce_t).  `ones_col` only in the direct layout.
int lime_wgrad_sp_launch(const LimeWgradSpPlan& w, const float* dy, long ldy, const float* x, long ldx, float* ws, int M, int N, int K,
                         int ones_col, hipStream_t s) {
    const int grid = w.n_tiles * w.k_tiles * w.splits;
    if (w.swap) wgrad_sp_kernel<<<grid, 512, 0, s>>>(x, ldx, dy, ldy, ws, M, K, N, w.n_tiles, w.k_tiles, w.rows_per_split, 0);
    else wgrad_sp_kernel<<<grid, 512, 0, s>>>(dy, ldy, x, ldx, ws, M, N, K, w.n_tiles, w.k_tiles, w.rows_per_split, ones_col);
    return lime_check_launch("wgrad_sp_kernel");
}

int lime_wgrad_sp_reduce_t(const LimeWgradSpPlan& w, const float* ws, float* dw, long lddw, int N, int K, int accumulate, hipStream_t s) {
    const long total = (long)N * K;
    const int grid = (int)((total + 255) / 256);
    reduce_partials_t_kernel<<<grid, 256, 0, s>>>(ws, w.np * w.kp, w.splits, w.kp, dw, lddw, N, K, accumulate);
    return lime_check_launch("reduce_partials_t_kernel");
}
