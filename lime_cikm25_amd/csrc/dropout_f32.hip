// Dropout inside the token encoders in training mode (reference: inplace dropout on the word embeddings
// newsEncoders.py:311-312, PositionalEncoding.dropout :827, and the three dropouts of nn.TransformerEncoderLayer built at
// :244-247 -- dropout1 after out_proj, dropout after the ReLU, dropout2 after linear2; the attention-probability dropout
// lives in the attention kernels).  Masks come from dropout.h: nothing is stored, the backward regenerates them.
// torch's own Philox stream cannot be reproduced bit for bit (it differs between its CPU and GPU generators too): parity is
// checked against a torch statement of the same layer fed with THESE masks (tests/test_dropout_gpu.py).
#include "common.h"
#include "dropout.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd,
                                                       long rows, int cols, LimeDropout d) {
    const long total = rows * cols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / cols;
        const int c = (int)(e - r * cols);
        dst[r * ldd + c] = lime_keep(d, (uint64_t)e) ? src[r * lds + c] * d.scale : 0.f;
    }
}

// cols % 4 == 0 and 16-byte aligned rows: four elements per thread, one hash
__global__ __launch_bounds__(256) void dropout_vec4_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd,
                                                            long rows, int cols, LimeDropout d) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int c4n = cols >> 2;
    const long total = rows * c4n;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / c4n;
        const int c = (int)(q - r * c4n) * 4;
        const unsigned m = lime_keep4(d, (uint64_t)q);             // elements 4 q .. 4 q + 3 = (r, c .. c + 3)
        v4 v = *reinterpret_cast<const v4*>(src + r * lds + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (m >> e) & 1u ? v[e] * d.scale : 0.f;
        *reinterpret_cast<v4*>(dst + r * ldd + c) = v;
    }
}

// two dropout sites over the same tensor in one pass (the backward through the positional and the embedding dropout of the layer input)
__global__ __launch_bounds__(256) void dropout2_vec4_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd,
                                                             long rows, int cols, LimeDropout d1, LimeDropout d2) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int c4n = cols >> 2;
    const long total = rows * c4n;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / c4n;
        const int c = (int)(q - r * c4n) * 4;
        const unsigned m = lime_keep4(d1, (uint64_t)q) & lime_keep4(d2, (uint64_t)q);
        v4 v = *reinterpret_cast<const v4*>(src + r * lds + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (m >> e) & 1u ? (v[e] * d1.scale) * d2.scale : 0.f;     // the same two roundings as two passes
        *reinterpret_cast<v4*>(dst + r * ldd + c) = v;
    }
}

// out[r, c] = drop_pe(drop_emb(table[ids[r], c]) + pe[r % period, c])
__global__ __launch_bounds__(256) void embed_pe_dropout_kernel(const int* __restrict__ ids, const float* __restrict__ table,
                                                                long ld_table, const float* __restrict__ pe, long ld_pe, int period,
                                                                float* __restrict__ out, long ldo, long rows, int dim,
                                                                LimeDropout d_emb, LimeDropout d_pe) {
    const long total = rows * dim;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / dim;
        const int c = (int)(e - r * dim);
        float v = table[(long)ids[r] * ld_table + c];
        v = lime_keep(d_emb, (uint64_t)e) ? v * d_emb.scale : 0.f;
        if (pe) v += pe[(r % period) * ld_pe + c];
        out[r * ldo + c] = lime_keep(d_pe, (uint64_t)e) ? v * d_pe.scale : 0.f;
    }
}

// y = LayerNorm(res + drop(t)); one wave per row, CPL columns per lane; rstd kept for the backward
template <int CPL>
__global__ __launch_bounds__(256) void dropout_add_ln_kernel(const float* __restrict__ t, long ldt, const float* __restrict__ res,
                                                              long ldr, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float* __restrict__ y, long ldy, float* __restrict__ rstd,
                                                              long M, int E, LimeDropout d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_e = 1.0f / (float)E;
    for (long r = (long)blockIdx.x * 4 + wave; r < M; r += (long)gridDim.x * 4) {
        float v[CPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            float x = 0.f;
            if (c < E) {
                const float tv = t[r * ldt + c];
                x = res[r * ldr + c] + (lime_keep(d, (uint64_t)(r * E + c)) ? tv * d.scale : 0.f);
            }
            v[j] = x;
            s += x;
        }
        const float mean = wave_sum(s) * inv_e;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            const float dlt = c < E ? v[j] - mean : 0.f;
            v[j] = dlt;
            q += dlt * dlt;
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * inv_e + eps);
        if (lane == 0 && rstd) rstd[r] = rs;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + 64 * j;
            if (c < E) y[r * ldy + c] = v[j] * rs * gamma[c] + beta[c];
        }
    }
}

// The same with 16 lanes per row (four rows per wave at a time), 16-byte accesses and one hash per four elements:
// E % 4 == 0, 16-byte aligned rows.
template <int V4>        // float4 per lane: ceil(E / 64)
__global__ __launch_bounds__(256) void dropout_add_ln_vec_kernel(const float* __restrict__ t, long ldt, const float* __restrict__ res,
                                                                  long ldr, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, float* __restrict__ y, long ldy, float* __restrict__ rstd,
                                                                  long M, int E, LimeDropout d) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const v4 zero = {0.f, 0.f, 0.f, 0.f};
    const float inv_e = 1.0f / (float)E;
    const int e4 = E >> 2;
    for (long r4 = ((long)blockIdx.x * 4 + wave) * 4; r4 < M; r4 += (long)gridDim.x * 16) {
        const long r = r4 + rg;
        const bool rin = r < M;
        v4 v[V4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c4 = sub + 16 * j;
            v4 x = zero;
            if (rin && c4 < e4) {
                const v4 tv = *reinterpret_cast<const v4*>(t + r * ldt + 4 * c4);
                x = *reinterpret_cast<const v4*>(res + r * ldr + 4 * c4);
                const unsigned m = lime_keep4(d, (uint64_t)(r * e4 + c4));
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] += (m >> e) & 1u ? tv[e] * d.scale : 0.f;
            }
            v[j] = x;
            s += (x[0] + x[1]) + (x[2] + x[3]);
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s * inv_e;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c4 = sub + 16 * j;
            if (c4 < e4) {
                v[j] = v[j] - mean;
                q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
            }
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
        const float rs = 1.0f / sqrtf(q * inv_e + eps);
        if (sub == 0 && rin && rstd) rstd[r] = rs;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c4 = sub + 16 * j;
            if (rin && c4 < e4) {
                const v4 ga = *reinterpret_cast<const v4*>(gamma + 4 * c4), be = *reinterpret_cast<const v4*>(beta + 4 * c4);
                *reinterpret_cast<v4*>(y + r * ldy + 4 * c4) = v[j] * rs * ga + be;
            }
        }
    }
}

__global__ __launch_bounds__(256) void embed_pe_dropout_vec4_kernel(const int* __restrict__ ids, const float* __restrict__ table,
                                                                     long ld_table, const float* __restrict__ pe, long ld_pe, int period,
                                                                     float* __restrict__ out, long ldo, long rows, int dim,
                                                                     LimeDropout d_emb, LimeDropout d_pe) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int c4n = dim >> 2;
    const long total = rows * c4n;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / c4n;
        const int c = (int)(q - r * c4n) * 4;
        v4 v = *reinterpret_cast<const v4*>(table + (long)ids[r] * ld_table + c);
        const unsigned m1 = lime_keep4(d_emb, (uint64_t)q), m2 = lime_keep4(d_pe, (uint64_t)q);
        v4 p = {0.f, 0.f, 0.f, 0.f};
        if (pe) p = *reinterpret_cast<const v4*>(pe + (r % period) * ld_pe + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = (m1 >> e) & 1u ? v[e] * d_emb.scale : 0.f;
            x += p[e];
            v[e] = (m2 >> e) & 1u ? x * d_pe.scale : 0.f;
        }
        *reinterpret_cast<v4*>(out + r * ldo + c) = v;
    }
}

int grid_for(long total) {
    const long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int lime_dropout_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, float p,
                                uint64_t seed, uint32_t site, void* stream) {
    LIME_REQUIRE(src && dst, LIME_ERR_BAD_ARG, "lime_dropout_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && cols > 0 && lds >= cols && ldd >= cols, LIME_ERR_BAD_ARG, "lime_dropout_f32: bad dimensions");
    LIME_REQUIRE(p >= 0.f && p < 1.f, LIME_ERR_BAD_ARG, "lime_dropout_f32: p = %g outside [0, 1)", (double)p);
    if (rows == 0) return LIME_OK;
    const LimeDropout d = lime_make_dropout(p, seed, site);
    if (cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0)
        dropout_vec4_kernel<<<grid_for(rows * (cols / 4)), 256, 0, (hipStream_t)stream>>>(src, lds, dst, ldd, rows, cols, d);
    else
        dropout_kernel<<<grid_for(rows * cols), 256, 0, (hipStream_t)stream>>>(src, lds, dst, ldd, rows, cols, d);
    return lime_check_launch("dropout_kernel");
}

extern "C" int lime_dropout2_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, float p, uint64_t seed,
                                 uint32_t site1, uint32_t site2, void* stream) {
    LIME_REQUIRE(src && dst, LIME_ERR_BAD_ARG, "lime_dropout2_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && cols > 0 && lds >= cols && ldd >= cols, LIME_ERR_BAD_ARG, "lime_dropout2_f32: bad dimensions");
    LIME_REQUIRE(p >= 0.f && p < 1.f, LIME_ERR_BAD_ARG, "lime_dropout2_f32: p = %g outside [0, 1)", (double)p);
    if (rows == 0) return LIME_OK;
    if (cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        dropout2_vec4_kernel<<<grid_for(rows * (cols / 4)), 256, 0, (hipStream_t)stream>>>(src, lds, dst, ldd, rows, cols,
                                                                                          lime_make_dropout(p, seed, site1),
                                                                                          lime_make_dropout(p, seed, site2));
        return lime_check_launch("dropout2_vec4_kernel");
    }
    const int st = lime_dropout_f32(src, lds, dst, ldd, rows, cols, p, seed, site1, stream);     // other layouts: two passes
    return st != LIME_OK ? st : lime_dropout_f32(dst, ldd, dst, ldd, rows, cols, p, seed, site2, stream);
}

extern "C" int lime_embed_pe_dropout_f32(const int32_t* ids, const float* table, int64_t ld_table, const float* pe, int64_t ld_pe,
                                         int32_t period, float* out, int64_t ldo, int64_t rows, int32_t dim, float p, uint64_t seed,
                                         uint32_t site_emb, uint32_t site_pe, void* stream) {
    LIME_REQUIRE(ids && table && out, LIME_ERR_BAD_ARG, "lime_embed_pe_dropout_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && ld_table >= dim && ldo >= dim && (!pe || (period > 0 && ld_pe >= dim)), LIME_ERR_BAD_ARG,
                 "lime_embed_pe_dropout_f32: bad dimensions");
    LIME_REQUIRE(p >= 0.f && p < 1.f, LIME_ERR_BAD_ARG, "lime_embed_pe_dropout_f32: p = %g outside [0, 1)", (double)p);
    if (rows == 0) return LIME_OK;
    const LimeDropout d1 = lime_make_dropout(p, seed, site_emb), d2 = lime_make_dropout(p, seed, site_pe);
    const bool vec = dim % 4 == 0 && ld_table % 4 == 0 && ldo % 4 == 0 && (!pe || ld_pe % 4 == 0) &&
                     ((((uintptr_t)table) | ((uintptr_t)out) | ((uintptr_t)pe)) & 15) == 0;
    if (vec)
        embed_pe_dropout_vec4_kernel<<<grid_for(rows * (dim / 4)), 256, 0, (hipStream_t)stream>>>(ids, table, ld_table, pe, ld_pe,
                                                                                                period > 0 ? period : 1, out, ldo, rows, dim, d1, d2);
    else
        embed_pe_dropout_kernel<<<grid_for(rows * dim), 256, 0, (hipStream_t)stream>>>(ids, table, ld_table, pe, ld_pe, period > 0 ? period : 1,
                                                                                     out, ldo, rows, dim, d1, d2);
    return lime_check_launch("embed_pe_dropout_kernel");
}

extern "C" int lime_dropout_add_layernorm_f32(const float* t, int64_t ldt, const float* res, int64_t ldr, const float* gamma,
                                              const float* beta, float eps, float* y, int64_t ldy, float* rstd, int64_t M, int32_t E,
                                              float p, uint64_t seed, uint32_t site, void* stream) {
    LIME_REQUIRE(t && res && gamma && beta && y, LIME_ERR_BAD_ARG, "lime_dropout_add_layernorm_f32: null pointer");
    LIME_REQUIRE(M >= 0 && E > 0 && ldt >= E && ldr >= E && ldy >= E, LIME_ERR_BAD_ARG, "lime_dropout_add_layernorm_f32: bad dimensions");
    LIME_REQUIRE(E <= 512, LIME_ERR_UNSUPPORTED, "lime_dropout_add_layernorm_f32: E = %d > 512", E);
    LIME_REQUIRE(p >= 0.f && p < 1.f, LIME_ERR_BAD_ARG, "lime_dropout_add_layernorm_f32: p = %g outside [0, 1)", (double)p);
    if (M == 0) return LIME_OK;
    const long g = (M + 3) / 4;
    const int grid = (int)(g > 8192 ? 8192 : g);
    const LimeDropout d = lime_make_dropout(p, seed, site);
    hipStream_t s = (hipStream_t)stream;
    const int cpl = (E + 63) / 64;
    const bool vec = E % 4 == 0 && ldt % 4 == 0 && ldr % 4 == 0 && ldy % 4 == 0 &&
                     ((((uintptr_t)t) | ((uintptr_t)res) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
    if (vec) {
        const long g16 = (M + 15) / 16;
        const int vgrid = (int)(g16 > 2048 ? 2048 : g16);
#define DALV(C) dropout_add_ln_vec_kernel<C><<<vgrid, 256, 0, s>>>(t, ldt, res, ldr, gamma, beta, eps, y, ldy, rstd, M, E, d)
        if (cpl <= 2) DALV(2); else if (cpl <= 5) DALV(5); else DALV(8);
#undef DALV
        return lime_check_launch("dropout_add_ln_vec_kernel");
    }
#define DAL(C) dropout_add_ln_kernel<C><<<grid, 256, 0, s>>>(t, ldt, res, ldr, gamma, beta, eps, y, ldy, rstd, M, E, d)
    if (cpl <= 2) DAL(2); else if (cpl <= 5) DAL(5); else DAL(8);
#undef DAL
    return lime_check_launch("dropout_add_ln_kernel");
}
