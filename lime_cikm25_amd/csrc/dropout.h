// Counter-based dropout masks for the training step: the decision for element `idx` of dropout site `site` under `seed`
// is a pure function of the three (a splitmix64 finaliser), so the backward kernels regenerate the forward's mask instead
// of storing it.  keep <=> hash >= p * 2^32; kept elements are scaled by 1 / (1 - p) (nn.Dropout's convention).
#pragma once
#include <stdint.h>

struct LimeDropout {
    uint64_t key;          // seed and site mixed on the host side of the launch
    uint32_t thresh;       // p * 2^32 (0: keep everything)
    float scale;           // 1 / (1 - p)
};

static inline LimeDropout lime_make_dropout(float p, uint64_t seed, uint32_t site) {
    LimeDropout d;
    d.key = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(site + 1) * 0xD1B54A32D192ED03ull;
    double t = (double)p * 4294967296.0;
    d.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t);
    d.scale = p <= 0.f ? 1.0f : 1.0f / (1.0f - p);
    return d;
}

#if defined(__HIPCC__)
__device__ __forceinline__ bool lime_keep(const LimeDropout& d, uint64_t idx) {
    uint64_t z = idx + d.key;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32) >= d.thresh;
}
#endif
