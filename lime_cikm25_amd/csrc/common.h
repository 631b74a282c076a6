// Shared host/device helpers for liblime_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/lime_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void lime_set_error(const char* fmt, ...);
void lime_set_last_linear_kernel(const char* fmt, ...);      // lime_last_linear_kernel(): which instantiation ran

#define LIME_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            lime_set_error(__VA_ARGS__);         \
            return (code);                       \
        }                                        \
    } while (0)

// called right after a kernel launch; never synchronises
static inline int lime_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lime_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LIME_ERR_LAUNCH;
    }
    return LIME_OK;
}

__device__ __forceinline__ float wave_half_sum(float v) {
    // sum over the 32 lanes of this lane's half-wave (xor offsets < 32 never cross the halves)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float wave_half_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_half_sum(v);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = wave_half_max(v);
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence
// also waits vmcnt(0): every global load still in flight (a prefetch) and every store (an epilogue) would be drained
// at each barrier.  The kernels here exchange data between waves through LDS only.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float lime_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// XCD-aware bijective remap of a 1-D grid: the 8 XCDs take workgroups round-robin by blockIdx, so
// logical ids are handed out such that each XCD walks one contiguous range (tiles that share
// operand panels then share an L2).  Placement only changes speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
