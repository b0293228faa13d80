// segmi — MI355X (gfx950 / CDNA4) kernels for the segmentation training hot path.
// Internal helpers shared by every .hip translation unit.  Nothing here is part of the C ABI
// (see include/segmi.h for the exported surface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/segmi.h"

#define SEGMI_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Every exported entry point returns a segmi_status; launches are asynchronous on the caller's
// stream, so only launch-configuration errors can be reported here.
static inline int segmi_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

static inline int segmi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// MI355X: 256 CUs in 8 XCDs.  Memory-bound grid-stride kernels cap their grid here.
#define SEGMI_NUM_CU 256
#define SEGMI_MAX_GRID (SEGMI_NUM_CU * 8)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: the dispatcher places block b on XCD b%8,
// so give each XCD a contiguous range of tile ids (neighbouring tiles share operand panels in that
// XCD's private L2).  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    unsigned q = nwg / nx, r = nwg % nx;
    unsigned xcd = bid % nx, idx = bid / nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
