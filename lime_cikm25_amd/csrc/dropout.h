// Counter-based dropout masks for the training step: the decision for element `idx` of dropout site `site` under `seed`
// is a pure function of the three, so the backward kernels regenerate the forward's mask instead of storing it.  One
// splitmix64 value serves the four elements idx & ~3 .. idx | 3 (16 bits each): keep <=> bits >= p * 2^16; kept elements are
// scaled by 1 / (1 - p) (nn.Dropout's convention).  Kernels that walk four consecutive elements hash once (lime_keep4).
#pragma once
#include <stdint.h>

struct LimeDropout {
    uint64_t key;          // seed and site mixed on the host side of the launch
    uint32_t thresh;       // p * 2^16 (0: keep everything)
    float scale;           // 1 / (1 - p)
};

static inline LimeDropout lime_make_dropout(float p, uint64_t seed, uint32_t site) {
    LimeDropout d;
    d.key = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(site + 1) * 0xD1B54A32D192ED03ull;
    double t = (double)p * 65536.0 + 0.5;
    d.thresh = p <= 0.f ? 0u : (t >= 65535.0 ? 0xFFFFu : (uint32_t)t);
    d.scale = p <= 0.f ? 1.0f : 1.0f / (1.0f - p);
    return d;
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t lime_hash4(const LimeDropout& d, uint64_t idx4) {      // idx4 = element index >> 2
    uint64_t z = idx4 + d.key;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ bool lime_keep(const LimeDropout& d, uint64_t idx) {
    return (uint32_t)((lime_hash4(d, idx >> 2) >> (16 * (unsigned)(idx & 3))) & 0xFFFFu) >= d.thresh;
}
// the four elements 4 idx4 .. 4 idx4 + 3 at once: bit e of the result = keep element e
__device__ __forceinline__ unsigned lime_keep4(const LimeDropout& d, uint64_t idx4) {
    const uint64_t z = lime_hash4(d, idx4);
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= ((uint32_t)((z >> (16 * e)) & 0xFFFFu) >= d.thresh ? 1u : 0u) << e;
    return m;
}
#endif
