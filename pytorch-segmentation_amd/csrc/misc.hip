// Layout transforms at the model boundary, strided row copies (concat / filter padding),
// dropout, and the status helpers of the C ABI.
#include "rowgeom.h"

namespace {

// [N][C][HW] -> [N][HW][ld]: 32x32 LDS transpose tiles; channels C..ld-1 zero-filled
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? src[((long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        if (p < HW && c < ld) dst[((long)n * HW + p) * ld + c] = tile[tx][j];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        tile[j][tx] = (p < HW && c < C) ? src[((long)n * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        if (c < C && p < HW) dst[((long)n * C + c) * HW + p] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void copy_rows_v4_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                           long rows, int c4copy, int c4fill) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4fill) return;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y)
        st4(dst + r * ldd + c4 * 4, c4 < c4copy ? ld4(src + r * lds + c4 * 4) : zero4());
}
__global__ __launch_bounds__(256) void copy_rows_scalar_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                               long rows, int C, int Cfill) {
    const long total = rows * Cfill;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / Cfill;
        const int c = (int)(i - r * Cfill);
        dst[r * ldd + c] = c < C ? src[r * lds + c] : 0.f;
    }
}

// counter-based mask: splitmix64 of (seed, index) -> uniform in [0,1)
__device__ __forceinline__ float u01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
template <bool CHANNELWISE>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long HW,
                                                      long rows, int C, float p, float scale, uint64_t seed0,
                                                      const uint64_t* __restrict__ epoch) {
    // epoch (optional, device memory): a per-step counter folded into the seed ON THE DEVICE, so that a step captured once
    // into a hipGraph draws fresh masks on every replay (a by-value seed would be frozen into the graph)
    const uint64_t seed = epoch ? seed0 + *epoch * 0xD1B54A32D192ED03ull : seed0;
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        const uint64_t base = CHANNELWISE ? (uint64_t)(r / HW) * C : (uint64_t)r * C;
        const int c = c4 * 4;
        float4 v = ld4(x + r * ldx + c);
        v.x = u01(seed, base + c) >= p ? v.x * scale : 0.f;
        v.y = u01(seed, base + c + 1) >= p ? v.y * scale : 0.f;
        v.z = u01(seed, base + c + 2) >= p ? v.z * scale : 0.f;
        v.w = u01(seed, base + c + 3) >= p ? v.w * scale : 0.f;
        st4(y + r * ldy + c, v);
    }
}

}  // namespace

extern "C" {

const char* segmi_strerror(int status) {
    switch (status) {
        case SEGMI_OK: return "ok";
        case SEGMI_ERR_BADARG: return "bad argument (null pointer, non-positive size or unsupported combination)";
        case SEGMI_ERR_ALIGN: return "alignment: ld / channel count / pointer must be a multiple of 4 floats (16 B)";
        case SEGMI_ERR_WORKSPACE: return "workspace missing or smaller than segmi_*_workspace() reports";
        case SEGMI_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        default: return "unknown segmi status";
    }
}
int segmi_abi_version(void) { return 10; }

int segmi_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int ld, segmi_stream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C || N > 65535) return SEGMI_ERR_BADARG;
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((ld + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, ld);
    return segmi_launch_status();
}
int segmi_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int ld, segmi_stream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C || N > 65535) return SEGMI_ERR_BADARG;
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, ld);
    return segmi_launch_status();
}

int segmi_copy_rows(const float* src, int ld_src, float* dst, int ld_dst, long rows, int C, int Cfill, segmi_stream_t stream) {
    if (!src || !dst || rows <= 0 || C <= 0 || Cfill < C || ld_src < C || ld_dst < Cfill) return SEGMI_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = !(C & 3) && !(Cfill & 3) && !(ld_src & 3) && !(ld_dst & 3) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15);
    if (v4) {
        RowGeom g = row_geom(rows, Cfill, 4, SEGMI_MAX_GRID);
        hipLaunchKernelGGL(copy_rows_v4_kernel, g.grid, g.block, 0, st, src, ld_src, dst, ld_dst, rows, C / 4, Cfill / 4);
    } else {
        long blocks = (rows * Cfill + 255) / 256;
        if (blocks > SEGMI_MAX_GRID) blocks = SEGMI_MAX_GRID;
        hipLaunchKernelGGL(copy_rows_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, ld_src, dst, ld_dst, rows, C, Cfill);
    }
    return segmi_launch_status();
}

int segmi_dropout(const float* x, int ldx, float* y, int ldy, int N, long HW, int C, float p, int channelwise,
                  uint64_t seed, const uint64_t* seed_epoch_dev, segmi_stream_t stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || !(p >= 0.f && p < 1.f)) return SEGMI_ERR_BADARG;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C) return SEGMI_ERR_ALIGN;
    const long rows = (long)N * HW;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
    const float scale = 1.f / (1.f - p);
    if (channelwise) hipLaunchKernelGGL((dropout_kernel<true>), g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, HW, rows, C, p, scale, seed, seed_epoch_dev);
    else             hipLaunchKernelGGL((dropout_kernel<false>), g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, HW, rows, C, p, scale, seed, seed_epoch_dev);
    return segmi_launch_status();
}

}  // extern "C"
