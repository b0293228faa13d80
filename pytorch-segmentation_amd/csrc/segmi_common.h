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

// A grid-stride loop over rows r = (n * A + a) * B + b needs (n, a, b) in every trip: two 64-bit divisions per trip cost more
// instructions than the 4 loads + 1 store of a float4 elementwise kernel (bilinear_fwd at 150 classes: 42 trips per thread, measured
// ALU-bound at 2.8 TB/s).  RowWalk3 divides ONCE (the start row and the stride) and then carries.
struct RowWalk3 {
    int b, a, n;          // current coordinates
    int db, da, dn;       // the stride, decomposed the same way
    int A, B;
    __device__ __forceinline__ void init(long r, long stride, int A_, int B_) {
        A = A_; B = B_;
        long t = r / B;
        b = (int)(r - t * B);
        long nn = t / A;
        a = (int)(t - nn * A); n = (int)nn;
        t = stride / B;
        db = (int)(stride - t * B);
        nn = t / A;
        da = (int)(t - nn * A); dn = (int)nn;
    }
    __device__ __forceinline__ void step() {
        b += db;
        const int c = b >= B ? 1 : 0;
        b -= c ? B : 0;
        a += da + c;
        const int c2 = a >= A ? 1 : 0;
        a -= c2 ? A : 0;
        n += dn + c2;
    }
};
