// Per-pixel losses, forward and backward, over NHWC logits [rows, C] (row stride ld) and int64
// targets [rows].  Wave-cooperative: LPP lanes share one pixel (each lane owns float4 channel
// groups g, g+LPP, ...), reductions over channels are 8-lane xor-shuffles, so a wave streams
// 8 pixels x up to 128 B per instruction, fully coalesced.
//
// CrossEntropyLoss2d  utils/losses.py:24-31  (aten::log_softmax + aten::nll_loss2d, ignore_index,
//                      reduction='mean' over non-ignored pixels; all-ignored -> NaN like torch)
#include "segmi_common.h"

namespace {

constexpr int LPP = 8;  // lanes per pixel

__device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

// per-pixel max and log-sum-exp over C channels; returns lse, and the target logit in xt
__device__ __forceinline__ float pixel_lse(const float* row, int C, int g, long t, float& xt) {
    const int c4n = (C + 3) >> 2;
    float m = -INFINITY;
    for (int q = g; q < c4n; q += LPP) {
        const float4 v = ld4(row + q * 4);
        const int c = q * 4;
        m = fmaxf(m, v.x);
        if (c + 1 < C) m = fmaxf(m, v.y);
        if (c + 2 < C) m = fmaxf(m, v.z);
        if (c + 3 < C) m = fmaxf(m, v.w);
    }
    m = grp_max(m);
    float s = 0.f, x_t = 0.f;
    for (int q = g; q < c4n; q += LPP) {
        const float4 v = ld4(row + q * 4);
        const int c = q * 4;
        s += expf(v.x - m);
        if (c + 1 < C) s += expf(v.y - m);
        if (c + 2 < C) s += expf(v.z - m);
        if (c + 3 < C) s += expf(v.w - m);
        if (t >= c && t < c + 4) { const int o = (int)(t - c); x_t = o == 0 ? v.x : o == 1 ? v.y : o == 2 ? v.z : v.w; }
    }
    s = grp_sum(s);
    xt = grp_sum(x_t);
    return m + logf(s);
}

// part[block] = {sum of -log p_t over valid pixels, number of valid pixels}
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                     long rows, int C, long ignore, float* __restrict__ lse_out,
                                                     double* __restrict__ part) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    float lsum = 0.f, lcnt = 0.f;
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        float xt;
        const float lse = pixel_lse(logits + r * ld, C, g, valid ? t : -1, xt);
        if (g == 0) {
            lse_out[r] = lse;
            if (valid) { lsum += lse - xt; lcnt += 1.f; }
        }
    }
    lsum = wave_sum(lsum); lcnt = wave_sum(lcnt);
    __shared__ float sm[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave] = lsum; sm[4 + wave] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (double)sm[0] + sm[1] + sm[2] + sm[3];
        part[2 * blockIdx.x + 1] = (double)sm[4] + sm[5] + sm[6] + sm[7];
    }
}

__global__ void ce_finalize_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out) {
    // single wave; nparts <= SEGMI_MAX_GRID
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) { s += part[2 * i]; c += part[2 * i + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if (threadIdx.x == 0) { out[0] = (float)(s / c); out[1] = (float)c; }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, long rows, int C, long ignore,
                                                     const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                     float* __restrict__ dl, int lddl) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    const float gs = grad_out[0] / loss_out[1];
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        const float l = lse[r];
        const float* row = logits + r * ld;
        for (int q = g; q < c4n; q += LPP) {
            float4 d = zero4();
            if (valid) {
                const float4 v = ld4(row + q * 4);
                const int c = q * 4;
                d.x = (expf(v.x - l) - (t == c ? 1.f : 0.f)) * gs;
                d.y = c + 1 < C ? (expf(v.y - l) - (t == c + 1 ? 1.f : 0.f)) * gs : 0.f;
                d.z = c + 2 < C ? (expf(v.z - l) - (t == c + 2 ? 1.f : 0.f)) * gs : 0.f;
                d.w = c + 3 < C ? (expf(v.w - l) - (t == c + 3 ? 1.f : 0.f)) * gs : 0.f;
            }
            st4(dl + r * lddl + q * 4, d);
        }
    }
}

int ce_blocks(long rows) {
    long b = (rows + 31) / 32;
    if (b < 1) b = 1;
    if (b > SEGMI_MAX_GRID) b = SEGMI_MAX_GRID;
    return (int)b;
}

}  // namespace

extern "C" {

size_t segmi_ce_workspace(long rows) { return (size_t)ce_blocks(rows) * 2 * sizeof(double); }

int segmi_ce_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float* lse,
                 float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_ce_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, ld, target, rows, C, ignore_index, lse, (double*)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, loss_out);
    return segmi_launch_status();
}

int segmi_ce_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C,
                 long ignore_index, const float* loss_out, const float* grad_out, float* dlogits, int lddl,
                 segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(rows)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, rows, C,
                       ignore_index, loss_out, grad_out, dlogits, lddl);
    return segmi_launch_status();
}

}  // extern "C"
