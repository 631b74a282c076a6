// Does `buffer_load_dwordx4 ... lds` hold up `s_waitcnt lgkmcnt(0)` (and with it every `s_waitcnt lgkmcnt + s_barrier`)?
// Diagnostic only.   hipcc -O3 --offload-arch=gfx950 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// PIECE: 0 = a DMA instruction covers 16 rows x 64 B (the GEMM's staging pattern), 1 = 8 rows x 128 B (whole cache lines)
template <int MODE, int PIECE = 0>   // 0: DMA only; 1: + s_waitcnt lgkmcnt(0) after each batch; 2: + s_waitcnt vmcnt(0) after each batch
__global__ __launch_bounds__(256, 2) void k(const float* src, float* out, int iters, long rows) {
    __shared__ __attribute__((aligned(16))) float lds[14336];            // 56 KB: two workgroups per CU
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7FFFFFF0, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned row = (blockIdx.x * 977u + wave * 131u + (PIECE ? (lane >> 3) : (lane >> 2))) % (unsigned)rows;
    unsigned voff = row * 1200u + (PIECE ? (lane & 7) : (lane & 3)) * 16u;
    const unsigned wrap = (unsigned)(rows * 1200 - 7 * 19200 - 20000);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + (wave * 7 + j) * 256), 16, voff + j * 19200u, (it % 18) * 64, 0, 0);
#endif
        }
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        voff += 7 * 19200u;
        if (voff > wrap) voff -= wrap;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
template <int MODE, int PIECE = 0>
float run(const float* src, float* out, long rows) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PIECE>), dim3(512), dim3(256), 0, 0, src, out, 2000, rows);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PIECE>), dim3(512), dim3(256), 0, 0, src, out, 2000, rows);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const long rows = 225280;
    float *src, *out; (void)hipMalloc(&src, rows * 1200 + 4096); (void)hipMalloc(&out, 512 * 256 * 4);
    (void)hipMemset(src, 0, rows * 1200 + 4096);
    const double bytes = 512.0 * 4 * 2000 * 7 * 1024;
    for (long r : {rows, 4096L, 640L}) {          // 270 MB (HBM), 4.9 MB (L2 / MALL), 0.77 MB (L2)
        float t0 = run<0>(src, out, r), t1 = run<1>(src, out, r), t2 = run<2>(src, out, r);
        printf("source %6ld rows of 1200 B\n", r);
        printf("  DMA only            : %.3f ms  %.2f TB/s  (%.0f cycles per 7-DMA batch at 2.4 GHz)\n", t0, bytes / t0 / 1e9, t0 * 1e-3 * 2.4e9 / 2000);
        printf("  + lgkmcnt(0) / batch: %.3f ms  %.2f TB/s  (%.0f cycles)\n", t1, bytes / t1 / 1e9, t1 * 1e-3 * 2.4e9 / 2000);
        printf("  + vmcnt(0) / batch  : %.3f ms  %.2f TB/s  (%.0f cycles)\n", t2, bytes / t2 / 1e9, t2 * 1e-3 * 2.4e9 / 2000);
        float t3 = run<0, 1>(src, out, r);
        printf("  DMA only, 8 rows x 128 B per instruction: %.3f ms  %.2f TB/s  (%.0f cycles)\n", t3, bytes / t3 / 1e9, t3 * 1e-3 * 2.4e9 / 2000);
    }
    return 0;
}
