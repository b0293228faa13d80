// Depthwise 3x3 convolution (Xception SeparableConv2d) and the 2x2 pixel shuffles that turn
// nn.ConvTranspose2d(kernel=2, stride=2) into a 1x1 convolution on the MFMA path.
//
//   depthwise : models/deeplabv3_plus.py:80 (`groups=in_channels`), reached from Block :89-132 and the exit-flow
//               separable convs :160-165 — 63 launches per forward for Xception, 0.75 % of the MACs: pure HBM
//               streaming (one read of x per output, 9 taps from L1/L2), so it is NOT reshaped into a GEMM.
//   shuffles  : models/unet.py:37 `nn.ConvTranspose2d(in, in//2, kernel_size=2, stride=2)`:
//               y[n, 2h+r, 2w+s, k] = b[k] + sum_c x[n,h,w,c] * W[c,k,r,s]  ==  a 1x1 conv with 4K outputs
//               (column k*4 + r*2 + s) followed by depth_to_space; its backward starts with space_to_depth.
//
// All kernels: NHWC fp32, a thread owns one float4 channel group and walks pixels (rowgeom.h).
#include "rowgeom.h"
#include "welford.h"
#include <cstdlib>

namespace {

struct DwGeom {
    int N, H, W, C, P, Q, R, S, stride, pad, dil;
};

// BN-statistics epilogue of the forward kernels (STATS): every thread keeps a Welford run over the outputs it writes, the block's
// thread rows merge through LDS (Chan, fixed tree) and the block writes ONE {count, mean, M2} partial per channel —
// stats[blockIdx.y][3][C], the layout of the dense convolution's epilogue (conv_igemm.hip) — so the BatchNorm behind a depthwise
// layer (SeparableConv2d: depthwise -> BN -> pointwise, models/deeplabv3_plus.py:76-86) never reads y for its statistics.
__device__ __forceinline__ void dw_stats_block(const Wf4& wf, bool cok, int c4, int C, float* __restrict__ stats) {
    __shared__ Wf4 sm[256];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    sm[t] = wf;
    __syncthreads();
    for (int s = blockDim.y >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.y < s) {
            Wf4 a = sm[t];
            wf_merge(a, sm[t + s * blockDim.x]);
            sm[t] = a;
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        const Wf4 a = sm[t];
        float* o = stats + (long)blockIdx.y * 3 * C + c4 * 4;
        st4(o, make_float4(a.n, a.n, a.n, a.n));
        st4(o + C, a.mean);
        st4(o + 2 * C, a.m2);
    }
}

// y[n,p,q,c] = sum_{r,s} x[n, p*stride - pad + r*dil, q*stride - pad + s*dil, c] * w[r,s,c]
template <int MAXT, bool STATS>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         float* __restrict__ y, int ldy, DwGeom g, float* __restrict__ stats) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 * 4 < g.C;
    if (!STATS && !cok) return;
    Wf4 wf;
    wf_init(wf);
    if (cok) {
    const int T = g.R * g.S;
    float4 wt[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) wt[t] = t < T ? ld4(w + (long)t * g.C + c4 * 4) : zero4();
    const long rows = (long)g.N * g.P * g.Q;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int q = (int)(row % g.Q);
        const long t1 = row / g.Q;
        const int pp = (int)(t1 % g.P), n = (int)(t1 / g.P);
        float4 acc = zero4();
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < T) {
                const int r = t / g.S, s = t - r * g.S;
                const int h = pp * g.stride - g.pad + r * g.dil, ww = q * g.stride - g.pad + s * g.dil;
                if ((unsigned)h < (unsigned)g.H && (unsigned)ww < (unsigned)g.W) {
                    const float4 v = ld4(x + ((long)(n * g.H + h) * g.W + ww) * ldx + c4 * 4);
                    acc.x = fmaf(v.x, wt[t].x, acc.x); acc.y = fmaf(v.y, wt[t].y, acc.y);
                    acc.z = fmaf(v.z, wt[t].z, acc.z); acc.w = fmaf(v.w, wt[t].w, acc.w);
                }
            }
        }
        st4(y + row * ldy + c4 * 4, acc);
        if (STATS) wf_push(wf, acc);
    }
    }
    if (STATS) dw_stats_block(wf, cok, c4, g.C, stats);
}

// dx[n,h,w,c] = sum_{r,s} dy[n, (h + pad - r*dil)/stride, (w + pad - s*dil)/stride, c] * w[r,s,c]   (where divisible)
template <int MAXT>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                           float* __restrict__ dx, int lddx, DwGeom g) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= g.C) return;
    const int T = g.R * g.S;
    float4 wt[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) wt[t] = t < T ? ld4(w + (long)t * g.C + c4 * 4) : zero4();
    const long rows = (long)g.N * g.H * g.W;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int ww = (int)(row % g.W);
        const long t1 = row / g.W;
        const int h = (int)(t1 % g.H), n = (int)(t1 / g.H);
        float4 acc = zero4();
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < T) {
                const int r = t / g.S, s = t - r * g.S;
                const int th = h + g.pad - r * g.dil, tw = ww + g.pad - s * g.dil;
                if (th >= 0 && tw >= 0 && th % g.stride == 0 && tw % g.stride == 0) {
                    const int pp = th / g.stride, q = tw / g.stride;
                    if (pp < g.P && q < g.Q) {
                        const float4 v = ld4(dy + ((long)(n * g.P + pp) * g.Q + q) * lddy + c4 * 4);
                        acc.x = fmaf(v.x, wt[t].x, acc.x); acc.y = fmaf(v.y, wt[t].y, acc.y);
                        acc.z = fmaf(v.z, wt[t].z, acc.z); acc.w = fmaf(v.w, wt[t].w, acc.w);
                    }
                }
            }
        }
        st4(dx + row * lddx + c4 * 4, acc);
    }
}

// part[blockIdx.y][t][C] = sum over this block's pixel range of dy[pix,c] * x[tap t of pix, c]
template <int MAXT>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                           float* __restrict__ part, DwGeom g) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 * 4 < g.C;
    const int T = g.R * g.S;
    float4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = zero4();
    const long rows = (long)g.N * g.P * g.Q;
    if (cok)
        for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
            const int q = (int)(row % g.Q);
            const long t1 = row / g.Q;
            const int pp = (int)(t1 % g.P), n = (int)(t1 / g.P);
            const float4 gy = ld4(dy + row * lddy + c4 * 4);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                if (t < T) {
                    const int r = t / g.S, s = t - r * g.S;
                    const int h = pp * g.stride - g.pad + r * g.dil, ww = q * g.stride - g.pad + s * g.dil;
                    if ((unsigned)h < (unsigned)g.H && (unsigned)ww < (unsigned)g.W) {
                        const float4 v = ld4(x + ((long)(n * g.H + h) * g.W + ww) * ldx + c4 * 4);
                        acc[t].x = fmaf(v.x, gy.x, acc[t].x); acc[t].y = fmaf(v.y, gy.y, acc[t].y);
                        acc[t].z = fmaf(v.z, gy.z, acc[t].z); acc[t].w = fmaf(v.w, gy.w, acc[t].w);
                    }
                }
            }
        }
    __shared__ float4 sm[256];
    const int tix = threadIdx.y * blockDim.x + threadIdx.x;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < T) {   // T is uniform: the barriers below are reached by every thread
            sm[tix] = acc[t];
            __syncthreads();
            for (int s = blockDim.y >> 1; s > 0; s >>= 1) {
                if ((int)threadIdx.y < s) {
                    float4 a = sm[tix], b = sm[tix + s * blockDim.x];
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                    sm[tix] = a;
                }
                __syncthreads();
            }
            if (threadIdx.y == 0 && cok) st4(part + ((long)blockIdx.y * T + t) * g.C + c4 * 4, sm[tix]);
            __syncthreads();
        }
    }
}


// ---- 3x3, stride 1, pad == dil == D ("same" depthwise convolutions: all of Xception's except the four strided ones): a thread
// computes a strip of 4 consecutive outputs of one image row, so the 3 x (4 + 2D) input taps it loads serve 4 outputs —
// 4.5 float4 loads per output for D = 1 instead of 9 (the one-output-per-thread kernels above are load-issue bound: 28 us per
// 728-channel 32x32 layer whose tensors stream in 10 us).
// FLIP: the filter is read rotated by 180 degrees — the data gradient of the same convolution (dx = dy (*) rot180(w)).
// PRE (round 6): the input is the PRE-NORMALISATION tensor z of the BatchNorm(+ReLU) in front of this layer; every tap is
// max(fmaf(z, scale[c], shift[c]), 0) evaluated on the loaded value — bn_apply_kernel's own expression, so the result equals
// bn_apply followed by the plain kernel bit for bit — and the normalised tensor between a pointwise convolution's BatchNorm and the
// next depthwise layer (Xception: models/deeplabv3_plus.py:99-119) is never written or read.  Taps outside the image are zeros of
// the NORMALISED tensor (zero padding applies after the BatchNorm), not transforms of a zero.
__device__ __forceinline__ float4 dw_pre(float4 v, const float4& sc, const float4& sh, bool relu) {
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}
template <int D, bool FLIP, bool STATS, bool PRE, bool HOIST>
__global__ __launch_bounds__(256) void dw3x3_strip_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, int ldy, int N, int H, int W, int C,
                                                          float* __restrict__ stats, const float* __restrict__ pre_scale,
                                                          const float* __restrict__ pre_shift, int pre_relu) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 * 4 < C;
    if (!STATS && !cok) return;
    Wf4 wf;
    wf_init(wf);
    if (cok) {
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = ld4(w + (long)(FLIP ? 8 - t : t) * C + c4 * 4);
    const float4 psc = PRE ? ld4(pre_scale + c4 * 4) : zero4(), psh = PRE ? ld4(pre_shift + c4 * 4) : zero4();
    const int QS = (W + 3) >> 2;
    const long strips = (long)N * H * QS;
    for (long sidx = (long)blockIdx.y * blockDim.y + threadIdx.y; sidx < strips; sidx += (long)gridDim.y * blockDim.y) {
        const int qs = (int)(sidx % QS);
        const long t1 = sidx / QS;
        const int p = (int)(t1 % H), n = (int)(t1 / H);
        const int q0 = qs * 4;
        float4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
        if (PRE || HOIST) {
            // ALL 3 x (4 + 2D) taps of the strip are requested before the first one is used (loads at clamped, always valid addresses;
            // the out-of-image ones are zeroed afterwards).  Left to itself the compiler emits load-row / wait / 48 FMAs three times
            // per strip — three HBM round trips in series for 4 outputs (rounds 4-5: 17-18 us for a 10 us transfer on the
            // 728-channel 32x32 layers) — and with the BatchNorm transform on the loaded value it waited tap by tap (first PRE build:
            // 22.8 us, 200 against 125 us at 256x256).
            float4 v[3][4 + 2 * D];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int h = p + (r - 1) * D;
                const int hc = (unsigned)h < (unsigned)H ? h : p;
                const float* rowp = x + ((long)(n * H + hc) * W) * ldx + c4 * 4;
#pragma unroll
                for (int j = 0; j < 4 + 2 * D; ++j) {
                    const int ww = q0 - D + j;
                    v[r][j] = ld4(rowp + (long)((unsigned)ww < (unsigned)W ? ww : q0) * ldx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool hok = (unsigned)(p + (r - 1) * D) < (unsigned)H;
#pragma unroll
                for (int j = 0; j < 4 + 2 * D; ++j) {
                    const int ww = q0 - D + j;
                    v[r][j] = (hok && (unsigned)ww < (unsigned)W) ? (PRE ? dw_pre(v[r][j], psc, psh, pre_relu != 0) : v[r][j]) : zero4();
                }
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const float4 f = wt[r * 3 + s2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 u = v[r][j + s2 * D];
                        acc[j].x = fmaf(u.x, f.x, acc[j].x); acc[j].y = fmaf(u.y, f.y, acc[j].y);
                        acc[j].z = fmaf(u.z, f.z, acc[j].z); acc[j].w = fmaf(u.w, f.w, acc[j].w);
                    }
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = p + (r - 1) * D;
            if ((unsigned)h >= (unsigned)H) continue;
            const float* rowp = x + ((long)(n * H + h) * W) * ldx + c4 * 4;
            float4 v[4 + 2 * D];
#pragma unroll
            for (int j = 0; j < 4 + 2 * D; ++j) {
                const int ww = q0 - D + j;
                v[j] = (unsigned)ww < (unsigned)W ? ld4(rowp + (long)ww * ldx) : zero4();
            }
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const float4 f = wt[r * 3 + s2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 u = v[j + s2 * D];
                    acc[j].x = fmaf(u.x, f.x, acc[j].x); acc[j].y = fmaf(u.y, f.y, acc[j].y);
                    acc[j].z = fmaf(u.z, f.z, acc[j].z); acc[j].w = fmaf(u.w, f.w, acc[j].w);
                }
            }
        }
        }
        float* o = y + ((long)(n * H + p) * W + q0) * ldy + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q0 + j < W) {
                st4(o + (long)j * ldy, acc[j]);
                if (STATS) wf_push(wf, acc[j]);
            }
    }
    }
    if (STATS) dw_stats_block(wf, cok, c4, C, stats);
}

// filter gradient of the same convolutions, same strips: part[blockIdx.y][t][C] = sum over this block's strips of dy (x) x-taps
template <int D, bool PRE, bool HOIST>
__global__ __launch_bounds__(256) void dw3x3_strip_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                                float* __restrict__ part, int N, int H, int W, int C,
                                                                const float* __restrict__ pre_scale, const float* __restrict__ pre_shift,
                                                                int pre_relu) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 * 4 < C;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = zero4();
    const float4 psc = (PRE && cok) ? ld4(pre_scale + c4 * 4) : zero4(), psh = (PRE && cok) ? ld4(pre_shift + c4 * 4) : zero4();
    const int QS = (W + 3) >> 2;
    const long strips = (long)N * H * QS;
    if (cok)
        for (long sidx = (long)blockIdx.y * blockDim.y + threadIdx.y; sidx < strips; sidx += (long)gridDim.y * blockDim.y) {
            const int qs = (int)(sidx % QS);
            const long t1 = sidx / QS;
            const int p = (int)(t1 % H), n = (int)(t1 / H);
            const int q0 = qs * 4;
            float4 gy[4];
            const float* gp = dy + ((long)(n * H + p) * W + q0) * lddy + c4 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) gy[j] = q0 + j < W ? ld4(gp + (long)j * lddy) : zero4();
            if (PRE || HOIST) {
                // all taps requested before the first use (see dw3x3_strip_kernel)
                float4 v[3][4 + 2 * D];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int h = p + (r - 1) * D;
                    const int hc = (unsigned)h < (unsigned)H ? h : p;
                    const float* rowp = x + ((long)(n * H + hc) * W) * ldx + c4 * 4;
#pragma unroll
                    for (int j = 0; j < 4 + 2 * D; ++j) {
                        const int ww = q0 - D + j;
                        v[r][j] = ld4(rowp + (long)((unsigned)ww < (unsigned)W ? ww : q0) * ldx);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const bool hok = (unsigned)(p + (r - 1) * D) < (unsigned)H;
#pragma unroll
                    for (int j = 0; j < 4 + 2 * D; ++j) {
                        const int ww = q0 - D + j;
                        v[r][j] = (hok && (unsigned)ww < (unsigned)W) ? (PRE ? dw_pre(v[r][j], psc, psh, pre_relu != 0) : v[r][j]) : zero4();
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 u = v[r][j + s2 * D];
                            float4& a = acc[r * 3 + s2];
                            a.x = fmaf(u.x, gy[j].x, a.x); a.y = fmaf(u.y, gy[j].y, a.y);
                            a.z = fmaf(u.z, gy[j].z, a.z); a.w = fmaf(u.w, gy[j].w, a.w);
                        }
                }
            } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int h = p + (r - 1) * D;
                if ((unsigned)h >= (unsigned)H) continue;
                const float* rowp = x + ((long)(n * H + h) * W) * ldx + c4 * 4;
                float4 v[4 + 2 * D];
#pragma unroll
                for (int j = 0; j < 4 + 2 * D; ++j) {
                    const int ww = q0 - D + j;
                    v[j] = (unsigned)ww < (unsigned)W ? ld4(rowp + (long)ww * ldx) : zero4();
                }
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 u = v[j + s2 * D];
                        float4& a = acc[r * 3 + s2];
                        a.x = fmaf(u.x, gy[j].x, a.x); a.y = fmaf(u.y, gy[j].y, a.y);
                        a.z = fmaf(u.z, gy[j].z, a.z); a.w = fmaf(u.w, gy[j].w, a.w);
                    }
            }
            }
        }
    // block reduction over the thread rows: all nine taps go to LDS at once (36 KB), ONE barrier, then thread (tx, ty) adds the rows
    // of taps ty, ty + ry, ... in a fixed order (round 4: a barrier tree per tap, 36 barriers per block for 2 strips of work per thread)
    __shared__ float4 sm[9][256];
    const int tix = threadIdx.y * blockDim.x + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 9; ++t) sm[t][tix] = acc[t];
    __syncthreads();
    if (cok)
        for (int t = threadIdx.y; t < 9; t += blockDim.y) {
            float4 a = sm[t][threadIdx.x];
            for (int r = 1; r < (int)blockDim.y; ++r) {
                const float4 b = sm[t][r * blockDim.x + threadIdx.x];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            st4(part + ((long)blockIdx.y * 9 + t) * C + c4 * 4, a);
        }
}

// out[i] = sum_p part[p][i]; block = (32 elements, 8 part lanes)
__global__ __launch_bounds__(256) void dw_sum_parts_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 32 + threadIdx.x;
    float a = 0.f;
    if (i < n) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four independent loads per step (a serial chain over up to 1024
        int p = threadIdx.y;                              // partials cost 14 us per depthwise filter gradient)
        for (; p + 24 < nparts; p += 32) {
            a0 += part[(long)p * n + i]; a1 += part[(long)(p + 8) * n + i];
            a2 += part[(long)(p + 16) * n + i]; a3 += part[(long)(p + 24) * n + i];
        }
        for (; p < nparts; p += 8) a0 += part[(long)p * n + i];
        a = (a0 + a1) + (a2 + a3);
    }
    __shared__ float sm[8][33];
    sm[threadIdx.y][threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.y == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][threadIdx.x];
        out[i] = t;
    }
}

// src[pix, k*4 + rs] -> dst[(2h + r, 2w + s), k] (+ bias[k]);  a thread moves a 4x4 (channel x position) block
__global__ __launch_bounds__(256) void depth_to_space2_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                              const float* __restrict__ bias, int N, int H, int W, int K) {
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;   // channels k4*4 .. +3
    if (k4 * 4 >= K) return;
    const float4 bv = bias ? ld4(bias + k4 * 4) : zero4();
    const long rows = (long)N * H * W;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int w = (int)(row % W);
        const long t1 = row / W;
        const int h = (int)(t1 % H), n = (int)(t1 / H);
        const float* sp = src + row * lds + k4 * 16;
        const float4 c0 = ld4(sp), c1 = ld4(sp + 4), c2 = ld4(sp + 8), c3 = ld4(sp + 12);   // channel k+j: (rs = 0..3)
        float* o = dst + ((long)(n * 2 * H + 2 * h) * (2 * W) + 2 * w) * ldd + k4 * 4;
        const long down = (long)2 * W * ldd;
        st4(o, make_float4(c0.x + bv.x, c1.x + bv.y, c2.x + bv.z, c3.x + bv.w));                 // r=0, s=0
        st4(o + ldd, make_float4(c0.y + bv.x, c1.y + bv.y, c2.y + bv.z, c3.y + bv.w));           // r=0, s=1
        st4(o + down, make_float4(c0.z + bv.x, c1.z + bv.y, c2.z + bv.z, c3.z + bv.w));          // r=1, s=0
        st4(o + down + ldd, make_float4(c0.w + bv.x, c1.w + bv.y, c2.w + bv.z, c3.w + bv.w));    // r=1, s=1
    }
}
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                              int N, int H, int W, int K) {
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 * 4 >= K) return;
    const long rows = (long)N * H * W;   // H, W = the LOW-resolution size; src is [N, 2H, 2W, K]
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int w = (int)(row % W);
        const long t1 = row / W;
        const int h = (int)(t1 % H), n = (int)(t1 / H);
        const float* i0 = src + ((long)(n * 2 * H + 2 * h) * (2 * W) + 2 * w) * lds + k4 * 4;
        const long down = (long)2 * W * lds;
        const float4 p00 = ld4(i0), p01 = ld4(i0 + lds), p10 = ld4(i0 + down), p11 = ld4(i0 + down + lds);
        float* o = dst + row * ldd + k4 * 16;
        st4(o, make_float4(p00.x, p01.x, p10.x, p11.x));
        st4(o + 4, make_float4(p00.y, p01.y, p10.y, p11.y));
        st4(o + 8, make_float4(p00.z, p01.z, p10.z, p11.z));
        st4(o + 12, make_float4(p00.w, p01.w, p10.w, p11.w));
    }
}

bool dw_ok(const segmi_conv_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->R <= 0 || d->S <= 0) return false;
    if (d->K != d->C || d->R * d->S > 9 || d->stride <= 0 || d->dil <= 0 || d->pad < 0) return false;
    if (d->P != (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1) return false;
    if (d->Q != (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1) return false;
    return d->P > 0 && d->Q > 0;
}
DwGeom dw_geom(const segmi_conv_desc* d) {
    DwGeom g;
    g.N = d->N; g.H = d->H; g.W = d->W; g.C = d->C; g.P = d->P; g.Q = d->Q; g.R = d->R; g.S = d->S;
    g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
    return g;
}
// number of pixel-range partials of the depthwise wgrad: enough blocks to fill the chip (>= ~4 per CU together with the
// channel tiles) while every block still reduces >= 32 pixels per row-lane
int dw_parts(long rows, int C) {
    RowGeom g = row_geom(rows, C, 1, 1);
    long want = (4L * SEGMI_NUM_CU + g.grid.x - 1) / g.grid.x;
    long cap = rows / ((long)g.ry * 8);
    long p = want < cap ? want : cap;
    if (p < 1) p = 1;
    if (p > 1024) p = 1024;
    return (int)p;
}
// the strip kernels serve 3x3, stride 1, pad == dil in {1, 2} (0 otherwise); SEGMI_DW_STRIP=0 keeps the one-output-per-thread kernels
int g_dw_strip = -1;
int dw_strip(const segmi_conv_desc* d) {
    if (g_dw_strip < 0) {
        const char* e = getenv("SEGMI_DW_STRIP");
        g_dw_strip = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!g_dw_strip || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != d->dil || d->dil > 2) return 0;
    return d->dil;
}

}  // namespace

extern "C" {

// strips per thread of the strip kernels (tuning hook SEGMI_DW_SPT; default 1 = one strip per thread, the whole grid resident at once)
static int dw_spt() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SEGMI_DW_SPT"); v = (e && atoi(e) > 0) ? atoi(e) : 1; }
    return v;
}
// SEGMI_DW_HOIST=1: the plain strip kernels with all taps requested up front too (A/B hook).  Default off: measured 154.5 against
// 156.5 img/s at cfg5 in alternating runs (profiles/r06_dw_hoist_ab.txt) — without the BatchNorm transform on the loaded value the
// compiler's row-by-row schedule already overlaps the rows, and the 18 live float4 taps cost occupancy.  The PRE kernels always hoist.
static bool dw_hoist() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SEGMI_DW_HOIST"); v = (e && atoi(e) == 1) ? 1 : 0; }
    return v == 1;
}
static RowGeom dw_fwd_geom(const segmi_conv_desc* d, int strip) {
    if (strip) return row_geom((long)d->N * d->H * ((d->W + 3) / 4), d->C, dw_spt(), SEGMI_MAX_GRID);
    return row_geom((long)d->N * d->P * d->Q, d->C, 2, SEGMI_MAX_GRID);
}
static int dw_fwd_impl(const segmi_conv_desc* d, const float* x, const float* w_rsc, float* y, float* stats, segmi_stream_t stream,
                       const float* pre_scale = nullptr, const float* pre_shift = nullptr, int pre_relu = 0) {
    if (!dw_ok(d) || !x || !w_rsc || !y) return SEGMI_ERR_BADARG;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return SEGMI_ERR_BADARG;
    if (pre_scale && !dw_strip(d)) return SEGMI_ERR_BADARG;               // the fused-BatchNorm load exists in the strip kernels only
    if ((d->C & 3) || (d->ldx & 3) || (d->ldy & 3) || d->ldx < d->C || d->ldy < d->C) return SEGMI_ERR_ALIGN;
    if (stats && ((uintptr_t)stats & 15)) return SEGMI_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int D = dw_strip(d);
    const RowGeom g = dw_fwd_geom(d, D);
    if (D) {
#define SEGMI_DW_STRIP(DV, SV, PV, HV) hipLaunchKernelGGL((dw3x3_strip_kernel<DV, false, SV, PV, HV>), g.grid, g.block, 0, st, x, d->ldx, w_rsc, y, d->ldy, d->N, d->H, d->W, d->C, stats, pre_scale, pre_shift, pre_relu)
        if (pre_scale) {
            if (D == 1) { if (stats) SEGMI_DW_STRIP(1, true, true, true); else SEGMI_DW_STRIP(1, false, true, true); }
            else        { if (stats) SEGMI_DW_STRIP(2, true, true, true); else SEGMI_DW_STRIP(2, false, true, true); }
        } else if (dw_hoist()) {
            if (D == 1) { if (stats) SEGMI_DW_STRIP(1, true, false, true); else SEGMI_DW_STRIP(1, false, false, true); }
            else        { if (stats) SEGMI_DW_STRIP(2, true, false, true); else SEGMI_DW_STRIP(2, false, false, true); }
        } else {
            if (D == 1) { if (stats) SEGMI_DW_STRIP(1, true, false, false); else SEGMI_DW_STRIP(1, false, false, false); }
            else        { if (stats) SEGMI_DW_STRIP(2, true, false, false); else SEGMI_DW_STRIP(2, false, false, false); }
        }
#undef SEGMI_DW_STRIP
        return segmi_launch_status();
    }
    if (stats) hipLaunchKernelGGL((dwconv_fwd_kernel<9, true>), g.grid, g.block, 0, st, x, d->ldx, w_rsc, y, d->ldy, dw_geom(d), stats);
    else       hipLaunchKernelGGL((dwconv_fwd_kernel<9, false>), g.grid, g.block, 0, st, x, d->ldx, w_rsc, y, d->ldy, dw_geom(d), (float*)nullptr);
    return segmi_launch_status();
}

int segmi_dwconv2d_fwd(const segmi_conv_desc* d, const float* x, const float* w_rsc, float* y, segmi_stream_t stream) {
    return dw_fwd_impl(d, x, w_rsc, y, nullptr, stream);
}

int segmi_dwconv2d_fwd_stats_parts(const segmi_conv_desc* d) {
    if (!dw_ok(d) || (d->C & 3)) return 0;
    return (int)dw_fwd_geom(d, dw_strip(d)).grid.y;
}

int segmi_dwconv2d_fwd_stats(const segmi_conv_desc* d, const float* x, const float* w_rsc, float* y, float* stats_partials,
                             segmi_stream_t stream) {
    if (!stats_partials) return SEGMI_ERR_BADARG;
    return dw_fwd_impl(d, x, w_rsc, y, stats_partials, stream);
}

int segmi_dwconv2d_pre_ok(const segmi_conv_desc* d) { return (dw_ok(d) && !(d->C & 3) && dw_strip(d)) ? 1 : 0; }

int segmi_dwconv2d_fwd_pre(const segmi_conv_desc* d, const float* z, const float* pre_scale, const float* pre_shift, int pre_relu,
                           const float* w_rsc, float* y, float* stats_partials, segmi_stream_t stream) {
    if (!pre_scale || !pre_shift) return SEGMI_ERR_BADARG;
    return dw_fwd_impl(d, z, w_rsc, y, stats_partials, stream, pre_scale, pre_shift, pre_relu);
}

int segmi_dwconv2d_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_rsc, float* dx, segmi_stream_t stream) {
    if (!dw_ok(d) || !dy || !w_rsc || !dx) return SEGMI_ERR_BADARG;
    if ((d->C & 3) || (d->ldx & 3) || (d->ldy & 3) || d->ldx < d->C || d->ldy < d->C) return SEGMI_ERR_ALIGN;
    const long rows = (long)d->N * d->H * d->W;
    if (const int D = dw_strip(d)) {          // dx = dy (*) rot180(w): the forward strip kernel with the filter read flipped
        RowGeom g = row_geom((long)d->N * d->H * ((d->W + 3) / 4), d->C, dw_spt(), SEGMI_MAX_GRID);
#define SEGMI_DW_DG(DV, HV) hipLaunchKernelGGL((dw3x3_strip_kernel<DV, true, false, false, HV>), g.grid, g.block, 0, (hipStream_t)stream, dy, d->ldy, w_rsc, dx, d->ldx, d->N, d->H, d->W, d->C, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0)
        if (dw_hoist()) { if (D == 1) SEGMI_DW_DG(1, true); else SEGMI_DW_DG(2, true); }
        else            { if (D == 1) SEGMI_DW_DG(1, false); else SEGMI_DW_DG(2, false); }
#undef SEGMI_DW_DG
        return segmi_launch_status();
    }
    RowGeom g = row_geom(rows, d->C, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL((dwconv_dgrad_kernel<9>), g.grid, g.block, 0, (hipStream_t)stream, dy, d->ldy, w_rsc, dx, d->ldx, dw_geom(d));
    return segmi_launch_status();
}

size_t segmi_dwconv2d_wgrad_workspace(const segmi_conv_desc* d) {
    if (!dw_ok(d)) return 0;
    return (size_t)dw_parts((long)d->N * d->P * d->Q, d->C) * d->R * d->S * d->C * sizeof(float);
}

static int dw_wgrad_impl(const segmi_conv_desc* d, const float* x, const float* dy, float* dw_rsc, void* workspace,
                         size_t workspace_bytes, segmi_stream_t stream, const float* pre_scale, const float* pre_shift, int pre_relu) {
    if (!dw_ok(d) || !x || !dy || !dw_rsc) return SEGMI_ERR_BADARG;
    if ((pre_scale == nullptr) != (pre_shift == nullptr) || (pre_scale && !dw_strip(d))) return SEGMI_ERR_BADARG;
    if ((d->C & 3) || (d->ldx & 3) || (d->ldy & 3) || d->ldx < d->C || d->ldy < d->C) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_dwconv2d_wgrad_workspace(d)) return SEGMI_ERR_WORKSPACE;
    const long rows = (long)d->N * d->P * d->Q;
    const int parts = dw_parts(rows, d->C);
    RowGeom g = row_geom(rows, d->C, 1, 1);
    g.grid.y = parts;
    hipStream_t st = (hipStream_t)stream;
    if (const int D = dw_strip(d)) {
#define SEGMI_DW_WG(DV, PV, HV) hipLaunchKernelGGL((dw3x3_strip_wgrad_kernel<DV, PV, HV>), g.grid, g.block, 0, st, x, d->ldx, dy, d->ldy, (float*)workspace, d->N, d->H, d->W, d->C, pre_scale, pre_shift, pre_relu)
        if (pre_scale)      { if (D == 1) SEGMI_DW_WG(1, true, true); else SEGMI_DW_WG(2, true, true); }
        else if (dw_hoist()) { if (D == 1) SEGMI_DW_WG(1, false, true); else SEGMI_DW_WG(2, false, true); }
        else                { if (D == 1) SEGMI_DW_WG(1, false, false); else SEGMI_DW_WG(2, false, false); }
#undef SEGMI_DW_WG
    } else
    hipLaunchKernelGGL((dwconv_wgrad_kernel<9>), g.grid, g.block, 0, st, x, d->ldx, dy, d->ldy, (float*)workspace, dw_geom(d));
    const int n = d->R * d->S * d->C;
    hipLaunchKernelGGL(dw_sum_parts_kernel, dim3(segmi_cdiv(n, 32)), dim3(32, 8), 0, st, (const float*)workspace, parts, n, dw_rsc);
    return segmi_launch_status();
}

int segmi_dwconv2d_wgrad(const segmi_conv_desc* d, const float* x, const float* dy, float* dw_rsc, void* workspace,
                         size_t workspace_bytes, segmi_stream_t stream) {
    return dw_wgrad_impl(d, x, dy, dw_rsc, workspace, workspace_bytes, stream, nullptr, nullptr, 0);
}

int segmi_dwconv2d_wgrad_pre(const segmi_conv_desc* d, const float* z, const float* pre_scale, const float* pre_shift, int pre_relu,
                             const float* dy, float* dw_rsc, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!pre_scale || !pre_shift) return SEGMI_ERR_BADARG;
    return dw_wgrad_impl(d, z, dy, dw_rsc, workspace, workspace_bytes, stream, pre_scale, pre_shift, pre_relu);
}

int segmi_depth_to_space2(const float* src, int ld_src, float* dst, int ld_dst, const float* bias, int N, int H, int W, int K,
                          segmi_stream_t stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || K <= 0) return SEGMI_ERR_BADARG;
    if ((K & 3) || (ld_src & 3) || (ld_dst & 3) || ld_src < 4 * K || ld_dst < K) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom((long)N * H * W, K, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(depth_to_space2_kernel, g.grid, g.block, 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, bias, N, H, W, K);
    return segmi_launch_status();
}

int segmi_space_to_depth2(const float* src, int ld_src, float* dst, int ld_dst, int N, int H, int W, int K, segmi_stream_t stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || K <= 0) return SEGMI_ERR_BADARG;
    if ((K & 3) || (ld_src & 3) || (ld_dst & 3) || ld_src < K || ld_dst < 4 * K) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom((long)N * H * W, K, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(space_to_depth2_kernel, g.grid, g.block, 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, N, H, W, K);
    return segmi_launch_status();
}

}  // extern "C"
