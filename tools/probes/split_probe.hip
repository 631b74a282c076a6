// Probe (diagnostic only): exact-fp32-level GEMM inner loop on the bf16 matrix cores by operand splitting.
//   x = hi + mid + lo (three bf16 terms, round-to-nearest each), product = hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi
//   (six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block; the dropped terms are <= 2^-24 of |a||w|).
// Measures (1) the arithmetic against an fp64 host product, beside the plain fp32 MFMA's error, and (2) the rate of the inner
// loop -- fragment reads from a static fp32 LDS image, the split in registers, the MFMAs -- for two wave tiles, no DMA, no barrier.
//   hipcc -O3 --offload-arch=gfx950 -o split_probe split_probe.hip && ./split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float lo16(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi16(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

struct Split { bf16x8 h, m, l; };
// 8 floats (two b128) -> three bf16x8: 11 VALU per pair of elements
__device__ __forceinline__ Split split8(f32x4 x0, f32x4 x1) {
    u32x4 h, m, l;
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned ph = pk(a, b);
        const float ra = a - lo16(ph), rb = b - hi16(ph);
        const unsigned pm = pk(ra, rb);
        const float sa = ra - lo16(pm), sb = rb - hi16(pm);
        h[p] = ph; m[p] = pm; l[p] = pk(sa, sb);
    }
    return Split{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}
__device__ __forceinline__ f32x4 mfma6(const Split& w, const Split& a, f32x4 c) {
    // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l, a.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, a.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.m, a.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.m, a.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, a.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, a.h, c, 0, 0, 0);
    return c;
}

// ---- (1) arithmetic: C[16 tokens][16 cols] = A[16][K] W[16][K]^T, one wave, split path and fp32-MFMA path
__global__ void k_check(const float* A, const float* W, int K, float* c_split, float* c_f32) {
    const int lane = threadIdx.x, fi = lane & 15, kg = lane >> 4;
    f32x4 cs = {0, 0, 0, 0}, cf = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        const f32x4 a0 = *(const f32x4*)(A + fi * K + k0 + 8 * kg), a1 = *(const f32x4*)(A + fi * K + k0 + 8 * kg + 4);
        const f32x4 w0 = *(const f32x4*)(W + fi * K + k0 + 8 * kg), w1 = *(const f32x4*)(W + fi * K + k0 + 8 * kg + 4);
        cs = mfma6(split8(w0, w1), split8(a0, a1), cs);
#pragma unroll
        for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q], a0[q], cf, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[q], a1[q], cf, 0, 0, 0);
    }
    // lane (fi, kg) holds token fi, columns 4 kg + r
    for (int r = 0; r < 4; ++r) { c_split[fi * 16 + 4 * kg + r] = cs[r]; c_f32[fi * 16 + 4 * kg + r] = cf[r]; }
}

// ---- (2) rate: NW waves per workgroup (one workgroup per CU), wave tile = 16 RT rows x 16 CT columns; the fp32 LDS image is
// [row][32 floats] (static contents); an iteration = one 32-deep chunk.  PRE: the W operand is read pre-split (three bf16 images).
template <int RT, int CT, int NW, bool PRE>
__global__ __launch_bounds__(NW * 64) void k_rate(float* out, int iters) {
    constexpr int AROWS = 256, WROWS = 320, PITCH = 36;          // 36-float rows: conflict-free b128 reads (the real image swizzles instead)
    __shared__ __attribute__((aligned(16))) float As[AROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float Ws[PRE ? WROWS * 16 * 3 : WROWS * PITCH];      // PRE: 3 bf16 images [row][32 bf16]
    for (int e = threadIdx.x; e < AROWS * PITCH; e += NW * 64) As[e] = (float)((e * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    for (int e = threadIdx.x; e < (int)(sizeof(Ws) / 4); e += NW * 64) Ws[e] = (float)((e * 40503u + 17) & 0xFFFF) / 65536.0f - 0.5f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, kg = lane >> 4;
    const int arow0 = (wave * 16 * RT) % AROWS, wrow0 = ((wave * 16 * CT) / AROWS * 16 * CT) % WROWS;
    f32x4 acc[RT][CT];
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        Split a[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float* p = As + (arow0 + 16 * i + fi) * PITCH + ((2 * kg + 2 * (it & 1)) & 7) * 4;      // varies per iteration: no hoisting
            a[i] = split8(*(const f32x4*)p, *(const f32x4*)(p + 4));
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            Split w;
            if constexpr (PRE) {
                const bf16x8* p = (const bf16x8*)Ws + ((wrow0 + 16 * j) % WROWS + fi) * 4 + kg;
                w.h = p[0]; w.m = p[WROWS * 4]; w.l = p[2 * WROWS * 4];
            } else {
                const float* p = Ws + ((wrow0 + 16 * j) % WROWS + fi) * PITCH + ((2 * kg + 2 * (it & 1)) & 7) * 4;
                w = split8(*(const f32x4*)p, *(const f32x4*)(p + 4));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i][j] = mfma6(w, a[i], acc[i][j]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

// the fp32 MFMA loop of the same shape (the baseline the split has to beat): 16 k per iteration x 2
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) void k_rate_f32(float* out, int iters) {
    constexpr int AROWS = 256, WROWS = 320, PITCH = 36;
    __shared__ __attribute__((aligned(16))) float As[AROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float Ws[WROWS * PITCH];
    for (int e = threadIdx.x; e < AROWS * PITCH; e += NW * 64) As[e] = (float)((e * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    for (int e = threadIdx.x; e < WROWS * PITCH; e += NW * 64) Ws[e] = (float)((e * 40503u + 17) & 0xFFFF) / 65536.0f - 0.5f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, kg = lane >> 4;
    const int arow0 = (wave * 16 * RT) % AROWS, wrow0 = ((wave * 16 * CT) / AROWS * 16 * CT) % WROWS;
    f32x4 acc[RT][CT];
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 a[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) a[i] = *(const f32x4*)(As + (arow0 + 16 * i + fi) * PITCH + ((kg + 4 * half + (it & 1)) & 7) * 4);
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const f32x4 w = *(const f32x4*)(Ws + ((wrow0 + 16 * j) % WROWS + fi) * PITCH + ((kg + 4 * half + (it & 1)) & 7) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], a[i][q], acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

// 32x32x16 form: wave tile 32 RT rows x 32 CT columns; per 32-deep chunk two k steps; lane (row = l & 31, khalf = l >> 5) holds
// k = 16 ks + 8 khalf .. + 7 of its row.  Same fragment count / split work per chunk as the 16x16x32 form, half the MFMA instructions.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma6_32(const Split& w, const Split& a, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, a.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, a.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, a.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.h, c, 0, 0, 0);
    return c;
}
template <int RT, int CT, int NW>
__global__ __launch_bounds__(NW * 64) void k_rate32(float* out, int iters) {
    constexpr int AROWS = 256, WROWS = 320, PITCH = 36;
    __shared__ __attribute__((aligned(16))) float As[AROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float Ws[WROWS * PITCH];
    for (int e = threadIdx.x; e < AROWS * PITCH; e += NW * 64) As[e] = (float)((e * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    for (int e = threadIdx.x; e < WROWS * PITCH; e += NW * 64) Ws[e] = (float)((e * 40503u + 17) & 0xFFFF) / 65536.0f - 0.5f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 31, kh = lane >> 5;
    const int arow0 = (wave * 32 * RT) % AROWS, wrow0 = ((wave * 32 * CT) / AROWS * 32 * CT) % WROWS;
    f32x16 acc[RT][CT];
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Split a[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float* p = As + (arow0 + 32 * i + fi) * PITCH + ((2 * (2 * ks + kh) + 2 * (it & 1)) & 7) * 4;
                a[i] = split8(*(const f32x4*)p, *(const f32x4*)(p + 4));
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const float* p = Ws + ((wrow0 + 32 * j) % WROWS + fi) * PITCH + ((2 * (2 * ks + kh) + 2 * (it & 1)) & 7) * 4;
                const Split w = split8(*(const f32x4*)p, *(const f32x4*)(p + 4));
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][j] = mfma6_32(w, a[i], acc[i][j]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < RT; ++i) for (int j = 0; j < CT; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <typename F>
double time_it(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / reps;
}

int main() {
    // (1) arithmetic
    for (int K : {32, 320, 512, 4096}) {
        std::vector<float> A(16 * K), W(16 * K);
        unsigned x = 777 + K;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return (float)(x >> 8) / 8388608.0f - 1.0f; };
        for (auto& v : A) v = rnd() * (1.0f + 3.0f * (rnd() > 0.9f));
        for (auto& v : W) v = rnd() * 0.1f;
        float *dA, *dW, *dcs, *dcf;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dcs, 1024); hipMalloc(&dcf, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dW, K, dcs, dcf);
        float cs[256], cf[256];
        hipMemcpy(cs, dcs, 1024, hipMemcpyDeviceToHost); hipMemcpy(cf, dcf, 1024, hipMemcpyDeviceToHost);
        double es = 0, ef = 0, scale = 0;
        for (int t = 0; t < 16; ++t)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[t * K + k] * W[n * K + k]; mag += fabs((double)A[t * K + k] * W[n * K + k]); }
                es = fmax(es, fabs(cs[t * 16 + n] - ref) / mag); ef = fmax(ef, fabs(cf[t * 16 + n] - ref) / mag); scale = fmax(scale, mag);
            }
        printf("K=%4d  max |c - fp64| / sum|a w|:  split-bf16 x6 %.3e   fp32 MFMA %.3e\n", K, es, ef);
    }
    // (2) rate
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 400, blocks = 256;
    auto report = [&](const char* name, double t, int rt, int ct, int nw) {
        const double fl = (double)blocks * nw * iters * rt * ct * 16.0 * 16.0 * 32.0 * 2.0;
        printf("%-46s %8.1f us  %7.1f TF fp32-equivalent (%.2f x the 157.3 TF fp32-MFMA peak)\n", name, t * 1e6, fl / t / 1e12, fl / t / 157.3e12);
    };
    for (int rep = 0; rep < 2; ++rep) {
        report("split, wave 64x160, 8 waves (2 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<4, 10, 8, false>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 4, 10, 8);
        report("split, wave 32x320, 8 waves (2 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<2, 20, 8, false>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 2, 20, 8);
        report("split, wave 32x160, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate<2, 10, 8, false>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 2, 10, 8);
        report("split, wave 64x160, 4 waves (1 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<4, 10, 4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 4, 10, 4);
        report("split, wave 128x160, 4 waves (1 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<8, 10, 4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 8, 10, 4);
        report("split, wave 64x320, 4 waves (1 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<4, 20, 4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 4, 20, 4);
        report("split, wave 96x160, 4 waves (1 / SIMD)", time_it([&] { hipLaunchKernelGGL((k_rate<6, 10, 4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 6, 10, 4);
        report("split 32x32x16, wave 64x160, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate32<2, 5, 8>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 4, 10, 8);
        report("split 32x32x16, wave 64x160, 4 waves", time_it([&] { hipLaunchKernelGGL((k_rate32<2, 5, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 4, 10, 4);
        report("split 32x32x16, wave 128x96, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate32<4, 3, 8>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 8, 6, 8);
        if (0) report("W pre-split, wave 64x160, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate<4, 10, 8, true>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 4, 10, 8);
        if (0) report("W pre-split, wave 32x320, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate<2, 20, 8, true>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 2, 20, 8);
        report("fp32 MFMA 16x16x4, wave 32x320, 8 waves", time_it([&] { hipLaunchKernelGGL((k_rate_f32<2, 20, 8>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 20), 2, 20, 8);
        report("fp32 MFMA 16x16x4, wave 32x320, 4 waves", time_it([&] { hipLaunchKernelGGL((k_rate_f32<2, 20, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 20), 2, 20, 4);
    }
    return 0;
}
