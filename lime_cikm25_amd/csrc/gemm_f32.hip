// lime_linear_f32: C = epilogue(A . W^T + bias) on the exact-fp32 matrix cores of gfx950.
//
// Every nn.Linear of the scoring path goes through this kernel (see include/lime_hip.h for the list
// of reference call sites).  Design, MI355X-first:
//   * v_mfma_f32_32x32x2_f32 (64 cycles / SIMD, bit-exact fp32 fma chain) -- the path's tolerance is
//     1e-3 against an fp32 CPU forward, and gfx950 has no xf32/TF32 mode, so the fp32 roof is the
//     157 TFLOP/s matrix rate, 16x below bf16.  At that rate one ds_read_b128 feeds four MFMAs, so the
//     kernel is paced by MFMA issue as long as operands arrive: LDS double buffer, global->register
//     prefetch of chunk t+1 issued before the MFMAs of chunk t, one barrier per 32-deep K chunk.
//   * both operands are K-contiguous row-major ([rows][K]: activations and nn.Linear weights as
//     stored), staged as [rows][36] fp32 so that a wave's ds_read_b128 (16-lane groups, 64 banks) is
//     conflict free; lane (i, h) fetches k = 8s+4h .. 8s+4h+3 and the four MFMAs of step s pair the
//     same k on both operands (the MFMA's k index is only a summation label).
//   * the word-embedding gather + positional table is fused into the A fetch (and into the residual
//     of the out_proj epilogue), bias / ReLU / tanh / residual / gated residual / LayerNorm into the
//     epilogue, so the encoder layer's activations cross HBM once per GEMM, not once per op.
//   * 1-D grid, XCD-aware bijective remap: an XCD walks whole row panels, so the N-tiles that re-read
//     one A panel hit the same 4 MiB L2.
#include "common.h"

namespace {

constexpr int BK = 32;    // K depth of one staged chunk
constexpr int LDK = 36;   // LDS row pitch in floats: 144 B, conflict-free for ds_read_b128, 16-B aligned

struct GemmP {
    const float* a; long lda; const int* a_ids; const float* a_pe; long lda_pe; int a_period;
    const float* w; long ldw; const float* bias;
    const float* res; long ldr; int res_div; const int* res_ids; const float* res_pe; long ldr_pe; int res_period;
    const float* gate_scale; int gate;
    const float* ln_g; const float* ln_b; float ln_eps;
    float* c; long ldc; int M, N, K; int act;
    int n_row_blocks, n_col_blocks;
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

// TM x TN MFMA tiles (32x32) per wave, WM x WN waves per workgroup (WM * WN == 4).
// ROWFULL: the workgroup spans every output column (WN == 1, n_col_blocks == 1) -> LayerNorm in the epilogue.
template <int TM, int TN, int WM, int WN, int VEC, bool ROWFULL>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int TPR = BK / VEC;          // threads per staged row
    constexpr int RPP = 256 / TPR;         // rows per staging pass
    constexpr int APASS = BM / RPP, WPASS = BN / RPP;
    static_assert(WM * WN == 4, "four waves per workgroup");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "staging passes must tile the block");
    typedef typename VecT<VEC>::T vec_t;

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, p.n_row_blocks * p.n_col_blocks);
    const int row_blk = logical / p.n_col_blocks, col_blk = logical - row_blk * p.n_col_blocks;
    const long row0 = (long)row_blk * BM;
    const int col0 = col_blk * BN;

    // ---- staging set-up: this thread's rows and k offset inside a chunk ---------------------------
    const int srow = tid / TPR, sk = (tid % TPR) * VEC;
    const float* arow[APASS];
    const float* aperow[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const long r = row0 + srow + i * RPP;
        arow[i] = nullptr;
        aperow[i] = nullptr;
        if (r < p.M) {
            if (p.a_ids) {
                arow[i] = p.a + (long)p.a_ids[r] * p.lda;
                if (p.a_pe) aperow[i] = p.a_pe + (long)(r % p.a_period) * p.lda_pe;
            } else {
                arow[i] = p.a + r * p.lda;
            }
        }
    }
    const float* wrow[WPASS];
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
        const int n = col0 + srow + i * RPP;
        wrow[i] = (n < p.N) ? p.w + (long)n * p.ldw : nullptr;
    }

    vec_t areg[APASS], wreg[WPASS];
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
        const bool kin = k < p.K;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            vec_t v = vzero<VEC>();
            if (kin && arow[i]) {
                v = *reinterpret_cast<const vec_t*>(arow[i] + k);
                if (aperow[i]) v += *reinterpret_cast<const vec_t*>(aperow[i] + k);
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WPASS; ++i) {
            vec_t v = vzero<VEC>();
            if (kin && wrow[i]) v = *reinterpret_cast<const vec_t*>(wrow[i] + k);
            wreg[i] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            *reinterpret_cast<vec_t*>(&As[buf][(srow + i * RPP) * LDK + sk]) = areg[i];
#pragma unroll
        for (int i = 0; i < WPASS; ++i)
            *reinterpret_cast<vec_t*>(&Ws[buf][(srow + i * RPP) * LDK + sk]) = wreg[i];
    };

    // ---- main loop ---------------------------------------------------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wrow0 = (wave / WN) * TM * 32, wcol0 = (wave % WN) * TN * 32;
    const int fi = lane & 31, fh = lane >> 5;
    const int a_off = (wrow0 + fi) * LDK + fh * 4;
    const int w_off = (wcol0 + fi) * LDK + fh * 4;

    const int nchunk = (p.K + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < nchunk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nchunk) fetch((t + 1) * BK);               // in flight under the MFMAs below
        const int kleft = p.K - t * BK;
        const int nstep = kleft >= BK ? BK / 8 : (kleft + 7) / 8;   // zero-filled tail beyond K
        const float* Ab = &As[buf][a_off];
        const float* Wb = &Ws[buf][w_off];
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            if (s < nstep) {
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
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], wf[j][u], acc[i][j], 0, 0, 0);
            }
        }
        if (t + 1 < nchunk) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            const bool rin = row < p.M;
            const float* rrow = nullptr;
            const float* rperow = nullptr;
            float gs = 1.f;
            if (rin && has_res) {
                if (p.res_ids) {
                    rrow = p.res + (long)p.res_ids[row] * p.ldr;
                    if (p.res_pe) rperow = p.res_pe + (long)(row % p.res_period) * p.ldr_pe;
                } else {
                    rrow = p.res + (row / p.res_div) * p.ldr;
                }
            }
            if (rin && p.gate) gs = p.gate_scale[row];
            float vals[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wcol0 + j * 32 + fi;
                const bool cin = col < p.N;
                float v = 0.f;
                if (rin && cin) {
                    float x = 0.f;
                    if (rrow) {
                        x = rrow[col];
                        if (rperow) x += rperow[col];
                    }
                    const float b = p.bias ? p.bias[col] : 0.f;
                    if (p.gate) {
                        const float g = lime_sigmoid(gs * acc[i][j][r] + b);
                        const float wc = gs * x;
                        v = g * wc + (1.f - g) * x;
                    } else {
                        v = apply_act(acc[i][j][r] + b, p.act) + x;
                    }
                }
                vals[j] = v;
            }
            if constexpr (ROWFULL) {
                if (p.ln_g) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) s += vals[j];
                    const float mean = wave_half_sum(s) / (float)p.N;
                    float q = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = col0 + wcol0 + j * 32 + fi;
                        const float d = (col < p.N) ? vals[j] - mean : 0.f;
                        vals[j] = d;
                        q += d * d;
                    }
                    const float rstd = 1.0f / sqrtf(wave_half_sum(q) / (float)p.N + p.ln_eps);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = col0 + wcol0 + j * 32 + fi;
                        if (col < p.N) vals[j] = vals[j] * rstd * p.ln_g[col] + p.ln_b[col];
                    }
                }
            }
            if (rin) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wcol0 + j * 32 + fi;
                    if (col < p.N) p.c[row * p.ldc + col] = vals[j];
                }
            }
        }
    }
}

template <int TM, int TN, int WM, int WN, bool ROWFULL>
int launch_cfg(const GemmP& p0, int vec, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    GemmP p = p0;
    p.n_row_blocks = (p.M + BM - 1) / BM;
    p.n_col_blocks = (p.N + BN - 1) / BN;
    const dim3 grid((unsigned)(p.n_row_blocks * p.n_col_blocks)), block(256);
    if (vec == 4) hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WM, WN, 4, ROWFULL>), grid, block, 0, stream, p);
    else if (vec == 2) hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WM, WN, 2, ROWFULL>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WM, WN, 1, ROWFULL>), grid, block, 0, stream, p);
    return lime_check_launch("lime_linear_f32");
}

inline bool aligned(const void* ptr, long ld, int vec) {
    return ptr == nullptr || (((uintptr_t)ptr % (vec * sizeof(float))) == 0 && (ld % vec) == 0);
}

}  // namespace

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
    LIME_REQUIRE(!a->gate || (a->gate_scale && a->res && !a->res_ids && a->res_div == 1), LIME_ERR_BAD_ARG,
                 "lime_linear_f32: gate needs gate_scale and a dense residual with res_div == 1");
    LIME_REQUIRE(!a->ln_gamma || a->ln_beta, LIME_ERR_BAD_ARG, "lime_linear_f32: ln_gamma without ln_beta");
    LIME_REQUIRE(a->act >= LIME_ACT_NONE && a->act <= LIME_ACT_SIGMOID, LIME_ERR_BAD_ARG, "lime_linear_f32: bad act %d", a->act);
    if (a->M == 0) return LIME_OK;

    GemmP p;
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids; p.a_pe = a->a_pe; p.lda_pe = a->lda_pe; p.a_period = a->a_period;
    p.w = a->w; p.ldw = a->ldw; p.bias = a->bias;
    p.res = a->res; p.ldr = a->ldr; p.res_div = a->res_div > 0 ? a->res_div : 1; p.res_ids = a->res_ids;
    p.res_pe = a->res_pe; p.ldr_pe = a->ldr_pe; p.res_period = a->res_period;
    p.gate_scale = a->gate_scale; p.gate = a->gate;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps;
    p.c = a->c; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act;
    p.n_row_blocks = p.n_col_blocks = 0;

    int vec = 4;
    while (vec > 1 && !((a->K % vec) == 0 && aligned(a->a, a->lda, vec) && aligned(a->w, a->ldw, vec) &&
                        aligned(a->a_pe, a->lda_pe, vec)))
        vec >>= 1;
    hipStream_t s = (hipStream_t)stream;
    if (a->ln_gamma) {
        LIME_REQUIRE(a->N <= 416, LIME_ERR_UNSUPPORTED, "lime_linear_f32: LayerNorm epilogue needs N <= 416 (N=%d)", a->N);
        if (a->N <= 320) return launch_cfg<1, 10, 4, 1, true>(p, vec, s);
        return launch_cfg<1, 13, 4, 1, true>(p, vec, s);
    }
    if (a->M >= 4096) return launch_cfg<2, 2, 2, 2, false>(p, vec, s);
    return launch_cfg<1, 1, 2, 2, false>(p, vec, s);
}
