// Micro-probes of the fp32 MFMA issue rate on gfx950 (diagnostic only, not part of the product).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V0: pure 32x32x2 MFMA, NACC independent accumulators, operands in registers
template <int NACC>
__global__ __launch_bounds__(256) void k_pure32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V1: pure 16x16x4 MFMA
template <int NACC>
__global__ __launch_bounds__(256) void k_pure16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V2: 2x2 tiles of 32x32 per wave fed by ds_read_b128 from a static LDS image (no staging, no barrier)
template <bool BARRIER>
__global__ __launch_bounds__(256) void k_lds32(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128 * 36], Ws[128 * 36];
    for (int e = threadIdx.x; e < 128 * 36; e += 256) { As[e] = e * 1e-4f; Ws[e] = e * 2e-4f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const float* Ab = &As[((wave >> 1) * 64 + fi) * 36 + fh * 4];
    const float* Wb = &Ws[((wave & 1) * 64 + fi) * 36 + fh * 4];
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 af[2], wf[2];
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * 36 + s * 8);
            for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const f32x4*>(Wb + j * 32 * 36 + s * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], wf[j][u], acc[i][j], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V3: sustained-clock probe.  Operands are 8 random values per lane in [-1, 1) (data-dependent switching power), four
// independent accumulators, s_memtime around the loop: cycles / wall time = the shader clock the chip actually holds under
// fp32 MFMA load, which is what the 157.3 TFLOP/s figure (2.4 GHz) has to be scaled by.
__global__ __launch_bounds__(256) void k_rand32(float* out, unsigned long long* cyc, const float* rnd, int iters, int zero) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        a[u] = zero ? 0.f : rnd[(threadIdx.x * 16 + u) & 4095];
        b[u] = zero ? 0.f : rnd[(threadIdx.x * 16 + 8 + u) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// V4: the same sustained probe on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: 8 bf16 per lane and operand)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_rand_bf16(float* out, const float* rnd, int iters, int zero) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
    for (int u = 0; u < 4; ++u)
        for (int e = 0; e < 8; ++e) {
            a[u][e] = (__bf16)(zero ? 0.f : rnd[(threadIdx.x * 64 + u * 8 + e) & 4095]);
            b[u][e] = (__bf16)(zero ? 0.f : rnd[(threadIdx.x * 64 + 32 + u * 8 + e) & 4095]);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// V4b: the two 16x16 bf16 shapes: v_mfma_f32_16x16x16_bf16 (4 bf16 per lane and operand: what ONE b128 fragment of a 16-deep fp32
// chunk splits into) and v_mfma_f32_16x16x32_bf16 (8 per lane) -- is the k = 16 form issued at the k = 32 form's FLOP rate?
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int K32>
__global__ __launch_bounds__(256) void k_rand_bf16_16(float* out, const float* rnd, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[4];
    for (int u = 0; u < 4; ++u)
        for (int e = 0; e < 8; ++e) {
            a[u][e] = (__bf16)rnd[(threadIdx.x * 64 + u * 8 + e) & 4095];
            b[u][e] = (__bf16)rnd[(threadIdx.x * 64 + 32 + u * 8 + e) & 4095];
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (K32) {
                    acc[4 * u + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[i], acc[4 * u + i], 0, 0, 0);
                } else {
                    const bf16x4 a4 = {a[u][0], a[u][1], a[u][2], a[u][3]}, b4 = {b[i][0], b[i][1], b[i][2], b[i][3]};
                    acc[4 * u + i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a4), __builtin_bit_cast(s16x4, b4), acc[4 * u + i], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// V5: 16x16x4 fp32 with random operands (is the small tile cheaper or dearer in power?)
__global__ __launch_bounds__(256) void k_rand16(float* out, const float* rnd, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) { a[u] = rnd[(threadIdx.x * 16 + u) & 4095]; b[u] = rnd[(threadIdx.x * 16 + 8 + u) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V6: k_rand16 again, but with the register budget of two waves per SIMD: hipcc then selects the VGPR form of the MFMA
// (accumulators in arch VGPRs instead of AccVGPRs).  Same instruction stream otherwise.
__global__ __launch_bounds__(256, 2) void k_rand16_vgpr(float* out, const float* rnd, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) { a[u] = rnd[(threadIdx.x * 16 + u) & 4095]; b[u] = rnd[(threadIdx.x * 16 + 8 + u) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
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
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    {
        float h[4096];
        unsigned x = 12345;
        for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 8388608.0f - 1.0f; }
        float* rnd; hipMalloc(&rnd, sizeof(h)); hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
        unsigned long long* cyc; hipMalloc(&cyc, 1024 * 8);
        for (int zero = 1; zero >= 0; --zero) {
            for (int reps : {5, 400}) {                 // a burst, then ~0.5 s of back-to-back launches
                const int blocks = 1024, it2 = 1000;
                double t = time_it([&] { hipLaunchKernelGGL(k_rand32, dim3(blocks), dim3(256), 0, 0, out, cyc, rnd, it2, zero); }, reps);
                unsigned long long hc[1024]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
                double mean = 0; for (int i = 0; i < blocks; ++i) mean += hc[i]; mean /= blocks;
                // per wave: it2 * 32 MFMAs of 64 cycles when the pipe is never idle; 4 waves per SIMD take turns
                printf("rand32 %s reps=%3d: %.2f TF   %.0f s_memtime cycles per wave loop, %.1f us per launch\n", zero ? "zeros " : "random", reps,
                       (double)blocks * 4 * it2 * 32 * 4096 / t / 1e12, mean, t * 1e6);
            }
        }
        {
            const int blocks = 1024, it2 = 4000;
            double t = time_it([&] { hipLaunchKernelGGL(k_rand_bf16_16<0>, dim3(blocks), dim3(256), 0, 0, out, rnd, it2); }, 100);
            printf("bf16 16x16x16 random sustained: %.1f TF  (%.1f cycles per MFMA at 2.4 GHz)\n", (double)blocks * 4 * it2 * 16 * 8192 / t / 1e12,
                   t * 2.4e9 / ((double)blocks / 256 * it2 * 16));
            t = time_it([&] { hipLaunchKernelGGL(k_rand_bf16_16<1>, dim3(blocks), dim3(256), 0, 0, out, rnd, it2); }, 100);
            printf("bf16 16x16x32 random sustained: %.1f TF  (%.1f cycles per MFMA at 2.4 GHz)\n", (double)blocks * 4 * it2 * 16 * 16384 / t / 1e12,
                   t * 2.4e9 / ((double)blocks / 256 * it2 * 16));
        }
        for (int zero = 1; zero >= 0; --zero) {
            const int blocks = 1024, it2 = 4000;
            double t = time_it([&] { hipLaunchKernelGGL(k_rand_bf16, dim3(blocks), dim3(256), 0, 0, out, rnd, it2, zero); }, 200);
            printf("bf16 32x32x16 %s sustained: %.1f TF  (%.1f us per launch)\n", zero ? "zeros " : "random",
                   (double)blocks * 4 * it2 * 16 * 32768 / t / 1e12, t * 1e6);
        }
        {
            const int blocks = 1024, it2 = 1000;
            double t = time_it([&] { hipLaunchKernelGGL(k_rand16, dim3(blocks), dim3(256), 0, 0, out, rnd, it2); }, 200);
            printf("f32 16x16x4 random sustained: %.2f TF\n", (double)blocks * 4 * it2 * 64 * 2048 / t / 1e12);
            for (int blk : {256, 512}) {
                t = time_it([&] { hipLaunchKernelGGL(k_rand16_vgpr, dim3(blk), dim3(256), 0, 0, out, rnd, it2); }, 100);
                printf("f32 16x16x4 random, VGPR-form accumulators, %d waves/SIMD: %.2f TF\n", blk / 256, (double)blk * 4 * it2 * 64 * 2048 / t / 1e12);
                t = time_it([&] { hipLaunchKernelGGL(k_rand16, dim3(blk), dim3(256), 0, 0, out, rnd, it2); }, 100);
                printf("f32 16x16x4 random, AGPR accumulators,      %d waves/SIMD: %.2f TF\n", blk / 256, (double)blk * 4 * it2 * 64 * 2048 / t / 1e12);
            }
        }
    }
    for (int blocks : {256}) {
        double t;
        t = time_it([&] { hipLaunchKernelGGL((k_pure32<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        printf("pure32 nacc=1 blocks=%4d: %.2f TF\n", blocks, (double)blocks * 4 * iters * 4 * 1 * 4096 / t / 1e12);
        t = time_it([&] { hipLaunchKernelGGL((k_pure32<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        printf("pure32 nacc=4 blocks=%4d: %.2f TF\n", blocks, (double)blocks * 4 * iters * 4 * 4 * 4096 / t / 1e12);
        t = time_it([&] { hipLaunchKernelGGL((k_pure16<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        printf("pure16 nacc=8 blocks=%4d: %.2f TF\n", blocks, (double)blocks * 4 * iters * 4 * 8 * 2048 / t / 1e12);
        t = time_it([&] { hipLaunchKernelGGL((k_lds32<false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
        printf("lds32 nobar   blocks=%4d: %.2f TF\n", blocks, (double)blocks * 4 * iters * 64 * 4096 / t / 1e12);
        t = time_it([&] { hipLaunchKernelGGL((k_lds32<true>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
        printf("lds32 barrier blocks=%4d: %.2f TF\n", blocks, (double)blocks * 4 * iters * 64 * 4096 / t / 1e12);
    }
    return 0;
}
