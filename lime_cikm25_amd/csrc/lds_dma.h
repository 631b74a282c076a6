// Device helpers shared by the bf16 encoder-block kernels (ffn_bf16.hip): buffer resources, LDS-DMA, the swizzled
// [row][64-byte] LDS image of gemm_pp_f32.hip, bf16 packing, counted vmcnt waits.  gfx950 only.
#pragma once
#include "common.h"

namespace lime_dev {

constexpr unsigned OOB = 0x80000000u;              // a buffer offset beyond num_records: loads / DMAs return zeros, stores are dropped
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
// 16 bytes per lane global -> LDS (lane l lands at lds_base + 16 l); out-of-range offsets write zeros.  Counts in vmcnt.
// (The builtin only exists in the device pass; inside a kernel TEMPLATE it makes the host pass drop the launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_base, 16, voff, soff, 0, 0);
#endif
}
// "all but the N youngest vector-memory operations of this wave are done" (loads, stores and LDS-DMA count together, in
// issue order: MI355X_MICROARCH.md)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Workgroup barrier with LDS-DMA left in flight across it (never __syncthreads(): its fence drains vmcnt)
__device__ __forceinline__ void ring_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One accumulator register -> a VGPR, HERE.  Left to itself hipcc copies every accumulator of a kernel out of the AccVGPRs in front
// of the first VALU use (160 copies ahead of a store epilogue: the VGPR file overflows and loop-carried values go to scratch);
// the "a" constraint keeps the value in its AccVGPR until this instruction.  The caller puts mfma_settle() between the last MFMA
// and the first of these: inline asm is outside the compiler's MFMA hazard bookkeeping.
__device__ __forceinline__ float acc_read(float a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
__device__ __forceinline__ f32x4 acc_read4(const f32x4& a) { return f32x4{acc_read(a[0]), acc_read(a[1]), acc_read(a[2]), acc_read(a[3])}; }
__device__ __forceinline__ void mfma_settle() {        // > the 16 cycles of a v_mfma_f32_16x16x32_bf16 plus its write-back
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Swizzle of the [row][four 16-byte segments] image: physical segment = logical ^ swz4((row >> 2) & 3) -- conflict free for
// the ds_read_b128 fragment reads of the 16x16 MFMA lane layout (gemm_pp_f32.hip has the derivation).
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

// two floats -> two bf16 in one register, round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer form -- add 0x7FFF +
// lsb, shift, merge -- is nine VALU instructions per pair: 4.6k of a wave's 5.8k VALU instructions per tile in the feed-forward kernel)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 v) {
    f32x4 r;
    r[0] = __builtin_bit_cast(float, v[0] << 16);
    r[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    r[2] = __builtin_bit_cast(float, v[1] << 16);
    r[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return r;
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4_bf16(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    u32x2 o;
    o[0] = pack_bf16(v[0], v[1]);
    o[1] = pack_bf16(v[2], v[3]);
    __builtin_amdgcn_raw_buffer_store_b64(o, r, voff, soff, 0);
}
// Sum over the 16 lanes of a DPP row (the 16 tokens of an MFMA tile); every lane ends up with the total.
__device__ __forceinline__ float row16_sum(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, true));       // row_ror:4
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true));       // row_ror:8
    return v;
}

}  // namespace lime_dev
