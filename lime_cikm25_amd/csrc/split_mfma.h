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

// ---------------------------------------------------------------------------------------------------------------------------------------
// A [rows][32] operand block as three bf16 images that serve BOTH operand kinds (token_attn_bwd_sp_f32.hip, the blocked attention
// backward): rows for a product that sums over the 32 columns (ds_read_b128: eight consecutive columns of one row), and TRANSPOSED for
// a product that sums over the rows, through gfx950's ds_read_b64_tr_b16 -- per 16-lane group a block of four rows x 16 columns arrives
// column-major: lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 of the block and receives column
// (lane & 15) of its four rows.  Row r = 64 bytes, its 16-byte chunk ch at position ch ^ swz_chunk(r), swz_chunk = 2 * bit 2 of r + bit 3
// of r: rows 4 apart share their banks (64-byte rows, 64 banks) -- bit 2 moves them to the other half of the bank group, bit 3 to the
// other chunk of the half -- so the row reads (16 lanes = 16 rows, one chunk), the transposed block reads (a 32-lane half = 8 consecutive
// rows x 32 bytes) and 8-byte staging writes are conflict free.  Image t of the block starts at img + t * term (bf16 units).
constexpr int SWZ_ROW = 32;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;
__device__ __forceinline__ int swz_chunk(int r) { return ((r >> 1) & 2) | ((r >> 3) & 1); }
__device__ __forceinline__ int swz_off(int r, int c) { return r * SWZ_ROW + (((c >> 3) ^ swz_chunk(r)) << 3) + (c & 7); }
__device__ __forceinline__ void swz_store2(unsigned short* img, int term, int r, int c, float a, float b) {       // c even
    const SplitPair t = split_pair(a, b);
    unsigned short* const d = img + swz_off(r, c);
    *reinterpret_cast<unsigned*>(d) = t.h;
    *reinterpret_cast<unsigned*>(d + term) = t.m;
    *reinterpret_cast<unsigned*>(d + 2 * term) = t.l;
}
__device__ __forceinline__ void swz_store4(unsigned short* img, int term, int r, int c, float x0, float x1, float x2, float x3) {   // c % 4 == 0
    u32x2 h, m, l;
    split_quad(x0, x1, x2, x3, h, m, l);
    unsigned short* const d = img + swz_off(r, c);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + term) = m;
    *reinterpret_cast<u32x2*>(d + 2 * term) = l;
}
// the fragment of row r, k values = columns 8 kg .. 8 kg + 7 (one chunk)
__device__ __forceinline__ SplitFrag swz_row_load(const unsigned short* img, int term, int r, int kg) {
    const unsigned short* const p = img + r * SWZ_ROW + ((kg ^ swz_chunk(r)) << 3);
    SplitFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p);
    f.m = *reinterpret_cast<const bf16x8*>(p + term);
    f.l = *reinterpret_cast<const bf16x8*>(p + 2 * term);
    return f;
}
// the A operand (rows = columns 16 c + fi of the image) of a product that sums over image rows: k values = rows 16 t0 + 4 kg + {0..3}
// and 16 (t0 + 1) + 4 kg + {0..3} -- the order in which two 16 x 16 MFMA result tiles hold their rows in a lane's registers.  Two
// transposed block reads per term; the swizzle is the same for the rows of a block and for rows 16 apart.  EXEC must be all ones.
__device__ __forceinline__ bf16x8 swz_tr_pair(const unsigned short* p0) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(p0));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(p0 + 16 * SWZ_ROW));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ SplitFrag swz_tr_load(const unsigned short* img, int term, int t0, int c, int fi, int kg) {
    const unsigned short* const p0 = img + swz_off(16 * t0 + 4 * kg + (fi >> 2), 16 * c + 4 * (fi & 3));
    SplitFrag f;
    f.h = swz_tr_pair(p0);
    f.m = swz_tr_pair(p0 + term);
    f.l = swz_tr_pair(p0 + 2 * term);
    return f;
}

}  // namespace lime_dev
