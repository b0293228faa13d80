// Two-kernel reproducer for the round-2 observation (profiles/r02_two_process_repeatability.txt): packed-fp32 VALU results
// (the HIGH half of v_pk_mul_f32 / v_pk_fma_f32 with op_sel) of one process's HBM-streaming kernel differed while ANOTHER process ran
// v_mfma_f32_32x32x16_bf16 kernels on the same GPU.  This program isolates the two ingredients:
//
//   pk_mfma_repro victim  <seconds> [pk|asm|asm2|scalar]   streams y = a * s + b over 64 MB with (pk) compiler-formed packed math, (asm) a
//                                                     hand-written v_pk_fma_f32 ... op_sel_hi:[1,0,1], or (scalar) plain v_fma_f32, and
//                                                     checks every launch's output against a scalar recomputation on the device
//   pk_mfma_repro aggressor <seconds> [bf16|f32]      keeps every CU busy with bf16 (or fp32) MFMA loops
//
// tools/gpu_round.sh `pkrepro` runs victim x {no aggressor, fp32 aggressor, bf16 aggressor} as separate PROCESSES sharing the GPU
// and prints the mismatch counts.   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_mfma_repro tools/probes/pk_mfma_repro.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

// ---- victim kernels: one float4 per thread and trip, like the bilinear / BN-apply streams of the product
__global__ __launch_bounds__(256) void victim_pk(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, float s, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i];
        // two float2 FMAs with a broadcast scalar: the SLP vectoriser forms v_pk_fma_f32 / v_pk_mul_f32 with op_sel here
        f32x2 lo = {u.x, u.y}, hi = {u.z, u.w};
        const f32x2 ss = {s, s};
        lo = lo * ss + (f32x2){v.x, v.y};
        hi = hi * ss + (f32x2){v.z, v.w};
        y[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
}
__global__ __launch_bounds__(256) void victim_asm(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, float s, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i];
        f32x2 lo = {u.x, u.y}, hi = {u.z, u.w}, blo = {v.x, v.y}, bhi = {v.z, v.w}, ss = {s, 0.f}, dlo, dhi;
        // high lane takes src1 from its LOW half (broadcast of s): the op_sel form the round-2 disassembly showed
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(dlo) : "v"(lo), "v"(ss), "v"(blo));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(dhi) : "v"(hi), "v"(ss), "v"(bhi));
        y[i] = make_float4(dlo.x, dlo.y, dhi.x, dhi.y);
    }
}
// the exact instruction forms of the SLP-built bilinear_fwd_kernel (profiles/r03_pk_two_process.txt): half-swapping v_pk_mov_b32,
// v_pk_mul_f32 with op_sel:[1,0] op_sel_hi:[0,1] (src0 halves crossed), plain v_pk_fma_f32.  y.xy = (a.x, a.y) * (s, t) crossed:
//   m = { a.y' ... } — the arithmetic below is checked against its own scalar restatement in check_asm2
__global__ __launch_bounds__(256) void victim_asm2(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, float s, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i];
        f32x2 w = {s, 1.0f - s}, p0 = {u.x, u.y}, p1 = {u.z, u.w}, q0 = {v.x, v.y}, q1 = {v.z, v.w}, t0, t1, m0, m1, d0, d1;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t0) : "v"(p0), "v"(q0));      // t0 = {p0.y, q0.x}
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t1) : "v"(p1), "v"(q1));      // t1 = {p1.y, q1.x}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(m0) : "v"(w), "v"(t0));   // m0 = {w.y*t0.x, w.x*t0.y}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(m1) : "v"(w), "v"(t1));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(w), "v"(p0), "v"(m0));       // d0 = w * p0 + m0
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(w), "v"(p1), "v"(m1));
        y[i] = make_float4(d0.x, d0.y, d1.x, d1.y);
    }
}
__device__ __forceinline__ float mul_scalar(float a, float b) {
    float d;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float fma_scalar(float a, float s, float b);
__global__ __launch_bounds__(256) void check_asm2(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ y, float s, long n,
                                                  unsigned long long* __restrict__ bad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i], g = y[i];
        const float wx = s, wy = 1.0f - s;
        const float e[4] = {fma_scalar(wx, u.x, mul_scalar(wy, u.y)), fma_scalar(wy, u.y, mul_scalar(wx, v.x)),
                            fma_scalar(wx, u.z, mul_scalar(wy, u.w)), fma_scalar(wy, u.w, mul_scalar(wx, v.z))};
        const float q[4] = {g.x, g.y, g.z, g.w};
        for (int c = 0; c < 4; ++c)
            if (__float_as_uint(e[c]) != __float_as_uint(q[c])) {
                atomicAdd(&bad[0], 1ull);
                atomicAdd(&bad[1 + c], 1ull);
                atomicCAS(&bad[5], 0ull, (unsigned long long)i + 1ull);
            }
    }
}
__device__ __forceinline__ float fma_scalar(float a, float s, float b) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(s), "v"(b));
    return d;
}
__global__ __launch_bounds__(256) void victim_scalar(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, float s, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i];
        y[i] = make_float4(fma_scalar(u.x, s, v.x), fma_scalar(u.y, s, v.y), fma_scalar(u.z, s, v.z), fma_scalar(u.w, s, v.w));
    }
}
// bit-exact check against scalar FMAs; records the number of wrong components, which component, and one example
__global__ __launch_bounds__(256) void check(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ y, float s, long n,
                                             unsigned long long* __restrict__ bad /* [0] total, [1..4] per component, [5] first index + 1 */) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 u = a[i], v = b[i], g = y[i];
        const float e[4] = {fma_scalar(u.x, s, v.x), fma_scalar(u.y, s, v.y), fma_scalar(u.z, s, v.z), fma_scalar(u.w, s, v.w)};
        const float q[4] = {g.x, g.y, g.z, g.w};
        for (int c = 0; c < 4; ++c)
            if (__float_as_uint(e[c]) != __float_as_uint(q[c])) {
                atomicAdd(&bad[0], 1ull);
                atomicAdd(&bad[1 + c], 1ull);
                atomicCAS(&bad[5], 0ull, (unsigned long long)i + 1ull);
            }
    }
}

// ---- aggressors: every wave issues MFMAs back to back (register operands only) for `iters` rounds
__global__ __launch_bounds__(256) void aggressor_bf16(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void aggressor_f32(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: %s victim|aggressor <seconds> [variant]\n", argv[0]); return 1; }
    const double secs = atof(argv[2]);
    const char* var = argc > 3 ? argv[3] : "";
    if (!strcmp(argv[1], "aggressor")) {
        float* out; CK(hipMalloc(&out, 4));
        const bool f32 = !strcmp(var, "f32");
        long launches = 0;
        const double t0 = now();
        while (now() - t0 < secs) {
            for (int k = 0; k < 8; ++k) {
                if (f32) hipLaunchKernelGGL(aggressor_f32, dim3(2048), dim3(256), 0, 0, out, 2000);
                else     hipLaunchKernelGGL(aggressor_bf16, dim3(2048), dim3(256), 0, 0, out, 4000);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
        printf("aggressor %s: %ld launches in %.1f s\n", f32 ? "f32" : "bf16", launches, now() - t0);
        return 0;
    }
    const long n = 4L << 20;                                   // 4 Mi float4 = 64 MB per array
    std::vector<float> ha(4 * n), hb(4 * n);
    unsigned x = 12345u;
    for (long i = 0; i < 4 * n; ++i) { x = x * 1664525u + 1013904223u; ha[i] = (float)((int)(x >> 8) - (1 << 23)) * 1e-6f; x = x * 1664525u + 1013904223u; hb[i] = (float)((int)(x >> 8) - (1 << 23)) * 3e-7f; }
    float4 *a, *b, *y; unsigned long long* bad;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMalloc(&bad, 48));
    CK(hipMemcpy(a, ha.data(), n * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 48));
    const int mode = !strcmp(var, "asm") ? 1 : (!strcmp(var, "scalar") ? 2 : (!strcmp(var, "asm2") ? 3 : 0));
    long launches = 0, bad_launches = 0;
    unsigned long long prev = 0, h[6];
    const double t0 = now();
    while (now() - t0 < secs) {
        const float s = 1.0f + 1e-3f * (float)(launches % 977);
        if (mode == 0) hipLaunchKernelGGL(victim_pk, dim3(4096), dim3(256), 0, 0, a, b, y, s, n);
        else if (mode == 1) hipLaunchKernelGGL(victim_asm, dim3(4096), dim3(256), 0, 0, a, b, y, s, n);
        else if (mode == 3) hipLaunchKernelGGL(victim_asm2, dim3(4096), dim3(256), 0, 0, a, b, y, s, n);
        else hipLaunchKernelGGL(victim_scalar, dim3(4096), dim3(256), 0, 0, a, b, y, s, n);
        if (mode == 3) hipLaunchKernelGGL(check_asm2, dim3(4096), dim3(256), 0, 0, a, b, y, s, n, bad);
        else hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, a, b, y, s, n, bad);
        ++launches;
        if (launches % 16 == 0) {
            CK(hipMemcpy(h, bad, 48, hipMemcpyDeviceToHost));
            if (h[0] != prev) { ++bad_launches; prev = h[0]; }
        }
    }
    CK(hipMemcpy(h, bad, 48, hipMemcpyDeviceToHost));
    printf("victim %-6s: %ld launches of 16 Mi results in %.1f s; wrong components %llu (x %llu, y %llu, z %llu, w %llu), first wrong float4 index %lld, "
           "16-launch windows with new errors %ld\n", mode == 0 ? "pk" : (mode == 1 ? "asm" : (mode == 3 ? "asm2" : "scalar")), launches, now() - t0, h[0], h[1], h[2], h[3], h[4],
           (long long)h[5] - 1, bad_launches);
    return h[0] ? 3 : 0;
}
