// The "split product" on the bf16 matrix cores (see gemm_sp_f32.hip): an fp32 value is three bf16 terms x = hi + mid + lo with exact
// residuals, a product block is six MFMAs (hh, hm, mh, mm, hl, lh; small terms first, fp32 accumulation) -- the error of one fp32
// rounding per product.  Helpers shared by the attention kernels that take their Q K^T / dO V^T products there.  gfx950 only.
#pragma once
#include "lds_dma.h"

namespace lime_dev {

struct SplitFrag { bf16x8 h, m, l; };                 // one MFMA operand fragment (8 k-values per lane) in its three terms
struct SplitPair { unsigned h, m, l; };               // two values, packed bf16 pairs

__device__ __forceinline__ SplitPair split_pair(float a, float b) {
    SplitPair r;
    r.h = pack_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, r.h << 16), rb = b - __builtin_bit_cast(float, r.h & 0xFFFF0000u);
    r.m = pack_bf16(ra, rb);
    r.l = pack_bf16(ra - __builtin_bit_cast(float, r.m << 16), rb - __builtin_bit_cast(float, r.m & 0xFFFF0000u));
    return r;
}
__device__ __forceinline__ SplitFrag split_frag(const float (&x)[8]) {
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const SplitPair t = split_pair(x[2 * q], x[2 * q + 1]);
        h[q] = t.h; m[q] = t.m; l[q] = t.l;
    }
    return SplitFrag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}
__device__ __forceinline__ SplitFrag split_frag(const f32x4 x0, const f32x4 x1) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    return split_frag(x);
}
// four values -> two packed pairs per term (8 bytes each)
__device__ __forceinline__ void split_quad(float x0, float x1, float x2, float x3, u32x2& h, u32x2& m, u32x2& l) {
    const SplitPair a = split_pair(x0, x1), b = split_pair(x2, x3);
    h = u32x2{a.h, b.h};
    m = u32x2{a.m, b.m};
    l = u32x2{a.l, b.l};
}
// C (16 x 16) += A (16 x 32) B (32 x 16), both operands in three terms
__device__ __forceinline__ f32x4 split_mfma16(const SplitFrag& a, const SplitFrag& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// C (32 x 32) += A (32 x 16) B (16 x 32)
__device__ __forceinline__ f32x16 split_mfma32(const SplitFrag& a, const SplitFrag& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// A [rows][32] operand block as three bf16 images in LDS: image t of row r starts at img + t * term_stride + r * SPLIT_PITCH (bf16
// units); 80-byte rows keep the fragment reads (ds_read_b128, 16 lanes = 16 rows) conflict free.
constexpr int SPLIT_PITCH = 40;
__device__ __forceinline__ void split_store2(unsigned short* img, int term_stride, int r, int c, float a, float b) {     // c even
    const SplitPair t = split_pair(a, b);
    unsigned short* const d = img + r * SPLIT_PITCH + c;
    *reinterpret_cast<unsigned*>(d) = t.h;
    *reinterpret_cast<unsigned*>(d + term_stride) = t.m;
    *reinterpret_cast<unsigned*>(d + 2 * term_stride) = t.l;
}
// the fragment of row `r`, k-values 8 kg .. 8 kg + 7
__device__ __forceinline__ SplitFrag split_load(const unsigned short* img, int term_stride, int r, int kg) {
    const unsigned short* const p = img + r * SPLIT_PITCH + 8 * kg;
    SplitFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p);
    f.m = *reinterpret_cast<const bf16x8*>(p + term_stride);
    f.l = *reinterpret_cast<const bf16x8*>(p + 2 * term_stride);
    return f;
}

}  // namespace lime_dev
