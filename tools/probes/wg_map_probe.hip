// Which workgroups of a 512-WG launch (2 resident per CU by LDS) share a CU?  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 -o wg_map_probe wg_map_probe.hip && ./wg_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    __shared__ float big[15000];                       // 60 KB: two workgroups per CU
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw = __builtin_amdgcn_s_getreg(0xF804);   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(0xF814);  // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < spin; ++i) s += big[(threadIdx.x + i) % 15000];
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t0;
        out[blockIdx.x * 4 + 3] = (unsigned)(t1 - t0) + (s == 12345.f);
    }
}
int main() {
    const int n = 512;
    unsigned* d; (void)hipMalloc(&d, n * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 20000);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned> h(n * 4);
    (void)hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    unsigned tmin = ~0u;
    for (int b = 0; b < n; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    for (int b = 0; b < n; ++b) {
        unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xF;
        unsigned cu_id = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
        if (b < 40) printf("wg %3d: xcc %u se %u sh %u cu %2u  start +%u ticks  dur %u\n", b, xcc, se, sh, cu_id, h[b * 4 + 2] - tmin, h[b * 4 + 3]);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", cu.size());
    int shown = 0;
    std::map<int, int> delta;
    for (auto& kv : cu) {
        if (shown++ < 12) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
    }
    for (auto& kv : delta) printf("pair distance %d: %d CUs\n", kv.first, kv.second);
    return 0;
}
