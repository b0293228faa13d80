// Per-pixel losses, forward and backward, over NHWC logits [rows, C] (row stride ld) and int64
// targets [rows].  Wave-cooperative: LPP lanes share one pixel (each lane owns float4 channel
// groups g, g+LPP, ...), reductions over channels are 8-lane xor-shuffles, so a wave streams
// 8 pixels x up to 128 B per instruction, fully coalesced.
//
// CrossEntropyLoss2d  utils/losses.py:24-31  (aten::log_softmax + aten::nll_loss2d, ignore_index,
//                      reduction='mean' over non-ignored pixels; all-ignored -> NaN like torch)
#include "segmi_common.h"
#include "bilinear.h"
#include "rowgeom.h"

// height pass of the separable bilinear backward (pool_resize.hip; C++ linkage, internal)
int segmi_internal_bilinear_bwd_height(const float* tmp, int ldt, float* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                                       int ac, hipStream_t st);

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

// part[block] = {sum of -w_t log p_t over valid pixels, sum of w_t over valid pixels}   (w = 1 without class weights:
// the pixel count).  nn.CrossEntropyLoss(weight=w, reduction='mean') = the quotient of the two (utils/losses.py:24-28).
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                     long rows, int C, long ignore, const float* __restrict__ cw,
                                                     float* __restrict__ lse_out, double* __restrict__ part) {
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
            if (valid) { const float w = cw ? cw[t] : 1.f; lsum += w * (lse - xt); lcnt += w; }
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

// out = {mean = s / c, c, s}: reduction 'mean' reads out[0]; 'sum' and the data-parallel global-batch mean (the caller
// all-reduces c and rescales, segmi/distributed.py) start from the raw sum out[2]
__global__ void ce_finalize_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out) {
    // single wave; nparts <= SEGMI_MAX_GRID
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) { s += part[2 * i]; c += part[2 * i + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if (threadIdx.x == 0) { out[0] = (float)(s / c); out[1] = (float)c; out[2] = (float)s; }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, long rows, int C, long ignore,
                                                     const float* __restrict__ cw, const float* __restrict__ loss_out,
                                                     const float* __restrict__ grad_out, float* __restrict__ dl, int lddl) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    const float gs0 = grad_out[0] / loss_out[1];
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        const float l = lse[r];
        const float* row = logits + r * ld;
        for (int q = g; q < c4n; q += LPP) {
            float4 d = zero4();
            if (valid) {
                const float gs = cw ? gs0 * cw[t] : gs0;
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

// ------------------------------------------------------------------------------------------------
// CrossEntropy on bilinearly UPSAMPLED logits, without materialising them  (SURVEY §8 f2).
// The reference upsamples its stride-8 (PSPNet, models/pspnet.py:85-91) or stride-4 (DeepLab decoder, models/deeplabv3_plus.py:361)
// logits to the input size and feeds the full-resolution tensor to the loss (trainer.py:56-66): at cfg2 that is a 201 MB tensor
// per head written, read twice by log_softmax / nll_loss forward + backward, a 201 MB gradient written and read again by the
// bilinear backward.  Here the loss kernel interpolates the four low-resolution neighbours of each output pixel on the fly
// (they stay in L2: the low-resolution logits are 2.7 MB) — the only full-resolution array is the per-pixel log-sum-exp
// (4 B/pixel) — and the backward recomputes softmax the same way while reducing along the width (pass W of the separable
// bilinear transpose), so the gradient is born at [N, OH, W_lo] and never exists at full resolution.
constexpr int UP_CACHE = 5;   // float4 channel groups a lane keeps in registers: 8 lanes x 5 x 4 = 160 classes
struct UpPix { const float* p00; const float* p01; const float* p10; const float* p11; float a0, a1, b0, b1; };
__device__ __forceinline__ UpPix up_pix(const float* lo, int ld, int n, int H, int W, const Lerp& a, const Lerp& b) {
    const float* base = lo + (long)n * H * W * ld;
    UpPix u;
    u.p00 = base + ((long)a.i0 * W + b.i0) * ld; u.p01 = base + ((long)a.i0 * W + b.i1) * ld;
    u.p10 = base + ((long)a.i1 * W + b.i0) * ld; u.p11 = base + ((long)a.i1 * W + b.i1) * ld;
    u.a0 = a.l0; u.a1 = a.l1; u.b0 = b.l0; u.b1 = b.l1;
    return u;
}
// same expression as bilinear_fwd_kernel (pool_resize.hip)
__device__ __forceinline__ float4 up4(const UpPix& u, int q) {
    const float4 v00 = ld4(u.p00 + q * 4), v01 = ld4(u.p01 + q * 4), v10 = ld4(u.p10 + q * 4), v11 = ld4(u.p11 + q * 4);
    float4 o;
    o.x = u.a0 * (u.b0 * v00.x + u.b1 * v01.x) + u.a1 * (u.b0 * v10.x + u.b1 * v11.x);
    o.y = u.a0 * (u.b0 * v00.y + u.b1 * v01.y) + u.a1 * (u.b0 * v10.y + u.b1 * v11.y);
    o.z = u.a0 * (u.b0 * v00.z + u.b1 * v01.z) + u.a1 * (u.b0 * v10.z + u.b1 * v11.z);
    o.w = u.a0 * (u.b0 * v00.w + u.b1 * v01.w) + u.a1 * (u.b0 * v10.w + u.b1 * v11.w);
    return o;
}

__global__ __launch_bounds__(256) void upce_fwd_kernel(const float* __restrict__ lo, int ld, int N, int H, int W, int C, int OH, int OW, int ac,
                                                       const int64_t* __restrict__ target, long ignore, const float* __restrict__ cw,
                                                       float* __restrict__ lse_out, double* __restrict__ part) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    const long rows = (long)N * OH * OW;
    const float sh = bl_scale(H, OH, ac), sw = bl_scale(W, OW, ac);
    float lsum = 0.f, lcnt = 0.f;
    const long r0 = (long)blockIdx.x * ppb + (threadIdx.x / LPP), stride = (long)gridDim.x * ppb;
    RowWalk3 rw;
    rw.init(r0, stride, OH, OW);
    for (long r = r0; r < rows; r += stride, rw.step()) {
        const int ow = rw.b, oh = rw.a, n = rw.n;
        const UpPix u = up_pix(lo, ld, n, H, W, bl_src(oh, sh, H, ac), bl_src(ow, sw, W, ac));
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        // interpolate once: a lane keeps its (up to UP_CACHE) channel groups in registers for the max and the exp-sum passes;
        // wider class counts (> 32 * UP_CACHE) recompute the tail
        float4 vc[UP_CACHE];
#pragma unroll
        for (int k = 0; k < UP_CACHE; ++k) { const int q = g + k * LPP; vc[k] = q < c4n ? up4(u, q) : zero4(); }
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < UP_CACHE; ++k) {
            const int q = g + k * LPP, c = q * 4;
            if (q < c4n) {
                m = fmaxf(m, vc[k].x);
                if (c + 1 < C) m = fmaxf(m, vc[k].y);
                if (c + 2 < C) m = fmaxf(m, vc[k].z);
                if (c + 3 < C) m = fmaxf(m, vc[k].w);
            }
        }
        for (int q = g + UP_CACHE * LPP; q < c4n; q += LPP) {
            const float4 v = up4(u, q);
            const int c = q * 4;
            m = fmaxf(m, v.x);
            if (c + 1 < C) m = fmaxf(m, v.y);
            if (c + 2 < C) m = fmaxf(m, v.z);
            if (c + 3 < C) m = fmaxf(m, v.w);
        }
        m = grp_max(m);
        float s = 0.f, x_t = 0.f;
#pragma unroll
        for (int k = 0; k < UP_CACHE; ++k) {
            const int q = g + k * LPP, c = q * 4;
            if (q < c4n) {
                const float4 v = vc[k];
                s += expf(v.x - m);
                if (c + 1 < C) s += expf(v.y - m);
                if (c + 2 < C) s += expf(v.z - m);
                if (c + 3 < C) s += expf(v.w - m);
                if (valid && t >= c && t < c + 4) { const int o = (int)(t - c); x_t = o == 0 ? v.x : o == 1 ? v.y : o == 2 ? v.z : v.w; }
            }
        }
        for (int q = g + UP_CACHE * LPP; q < c4n; q += LPP) {
            const float4 v = up4(u, q);
            const int c = q * 4;
            s += expf(v.x - m);
            if (c + 1 < C) s += expf(v.y - m);
            if (c + 2 < C) s += expf(v.z - m);
            if (c + 3 < C) s += expf(v.w - m);
            if (valid && t >= c && t < c + 4) { const int o = (int)(t - c); x_t = o == 0 ? v.x : o == 1 ? v.y : o == 2 ? v.z : v.w; }
        }
        s = grp_sum(s);
        const float xt = grp_sum(x_t);
        const float lse = m + logf(s);
        if (g == 0) {
            lse_out[r] = lse;
            if (valid) { const float w = cw ? cw[t] : 1.f; lsum += w * (lse - xt); lcnt += w; }
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

// Up to 32 classes (VOC 21, Cityscapes 19): ONE lane per output pixel keeps the pixel's 8 interpolated float4 groups in
// registers — no cross-lane reductions, 2-3 instructions per pixel and wave instead of ~25 with 8 lanes per pixel (that form,
// above, stays for wide class counts such as ADE20K's 150).  Neighbouring lanes read the same low-resolution taps (L1 hits).
__global__ __launch_bounds__(256) void upce_fwd_small_kernel(const float* __restrict__ lo, int ld, int N, int H, int W, int C, int OH, int OW, int ac,
                                                             const int64_t* __restrict__ target, long ignore, const float* __restrict__ cw,
                                                             float* __restrict__ lse_out, double* __restrict__ part) {
    constexpr int G = 8;
    const int c4n = (C + 3) >> 2;
    const long rows = (long)N * OH * OW;
    const float sh = bl_scale(H, OH, ac), sw = bl_scale(W, OW, ac);
    float lsum = 0.f, lcnt = 0.f;
    const long r0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    RowWalk3 rw;
    rw.init(r0, stride, OH, OW);
    for (long r = r0; r < rows; r += stride, rw.step()) {
        const int ow = rw.b, oh = rw.a, n = rw.n;
        const UpPix u = up_pix(lo, ld, n, H, W, bl_src(oh, sh, H, ac), bl_src(ow, sw, W, ac));
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        float4 v[G];
        float m = -INFINITY;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            v[q] = zero4();
            if (q < c4n) {
                v[q] = up4(u, q);
                const int c = q * 4;
                m = fmaxf(m, v[q].x);
                if (c + 1 < C) m = fmaxf(m, v[q].y);
                if (c + 2 < C) m = fmaxf(m, v[q].z);
                if (c + 3 < C) m = fmaxf(m, v[q].w);
            }
        }
        float s = 0.f, xt = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (q < c4n) {
                const int c = q * 4;
                s += expf(v[q].x - m);
                if (c + 1 < C) s += expf(v[q].y - m);
                if (c + 2 < C) s += expf(v[q].z - m);
                if (c + 3 < C) s += expf(v[q].w - m);
                if (valid && t >= c && t < c + 4) { const int o = (int)(t - c); xt = o == 0 ? v[q].x : o == 1 ? v[q].y : o == 2 ? v[q].z : v[q].w; }
            }
        const float lse = m + logf(s);
        lse_out[r] = lse;
        if (valid) { const float w = cw ? cw[t] : 1.f; lsum += w * (lse - xt); lcnt += w; }
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

// pass W of the backward: tmp[(n, oh, wl), c] = sum_ow ww(ow -> wl) * w_t * (softmax(n, oh, ow)[c] - [t == c]) * g / denominator
// (thread = one float4 channel group of one (n, oh, wl) row; ~2 * OW/W + 2 candidate output columns each)
__global__ __launch_bounds__(256) void upce_bwd_w_kernel(const float* __restrict__ lo, int ld, int N, int H, int W, int C, int OH, int OW, int ac,
                                                         const int64_t* __restrict__ target, const float* __restrict__ lse, long ignore,
                                                         const float* __restrict__ cw, const float* __restrict__ loss_out,
                                                         const float* __restrict__ grad_out, float* __restrict__ tmp, int ldt) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const int c = c4 * 4;
    const long rows = (long)N * OH * W;
    const float sh = bl_scale(H, OH, ac), sw = bl_scale(W, OW, ac);
    const float gs0 = grad_out[0] / loss_out[1];
    const long r0 = (long)blockIdx.y * blockDim.y + threadIdx.y, stride = (long)gridDim.y * blockDim.y;
    RowWalk3 rw;
    rw.init(r0, stride, OH, W);
    for (long r = r0; r < rows; r += stride, rw.step()) {
        const int wl = rw.b, oh = rw.a, n = rw.n;
        const long tq = (long)n * OH + oh;
        const Lerp a = bl_src(oh, sh, H, ac);
        int lo_c, hi_c;
        bl_range(wl, sw, W, OW, ac, lo_c, hi_c);
        // every output column with a non-zero weight on wl interpolates between wl and ONE of its neighbours: blend the two
        // low-resolution rows (a.i0, a.i1) once for the columns wl-1, wl, wl+1 and reuse them for all ~2*OW/W candidates
        const float* base = lo + (long)n * H * W * ld + c;
        const int wm = max(wl - 1, 0), wp = min(wl + 1, W - 1);
        float4 R[3];
        {
            const int cols[3] = {wm, wl, wp};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 t0 = ld4(base + ((long)a.i0 * W + cols[j]) * ld), t1 = ld4(base + ((long)a.i1 * W + cols[j]) * ld);
                R[j] = make_float4(a.l0 * t0.x + a.l1 * t1.x, a.l0 * t0.y + a.l1 * t1.y, a.l0 * t0.z + a.l1 * t1.z, a.l0 * t0.w + a.l1 * t1.w);
            }
        }
        float4 acc = zero4();
        for (int ow = lo_c; ow <= hi_c; ++ow) {
            const Lerp b = bl_src(ow, sw, W, ac);
            const float wt = (b.i0 == wl ? b.l0 : 0.f) + (b.i1 == wl ? b.l1 : 0.f);
            if (wt == 0.f) continue;
            const long hp = tq * OW + ow;
            const long t = target[hp];
            if (t == ignore || t < 0 || t >= C) continue;
            const float4 r0 = b.i0 == wl ? R[1] : (b.i0 < wl ? R[0] : R[2]);
            const float4 r1 = b.i1 == wl ? R[1] : (b.i1 < wl ? R[0] : R[2]);
            const float4 v = make_float4(b.l0 * r0.x + b.l1 * r1.x, b.l0 * r0.y + b.l1 * r1.y, b.l0 * r0.z + b.l1 * r1.z, b.l0 * r0.w + b.l1 * r1.w);
            const float l = lse[hp];
            const float k = wt * (cw ? gs0 * cw[t] : gs0);
            acc.x += k * (expf(v.x - l) - (t == c ? 1.f : 0.f));
            if (c + 1 < C) acc.y += k * (expf(v.y - l) - (t == c + 1 ? 1.f : 0.f));
            if (c + 2 < C) acc.z += k * (expf(v.z - l) - (t == c + 2 ? 1.f : 0.f));
            if (c + 3 < C) acc.w += k * (expf(v.w - l) - (t == c + 3 ? 1.f : 0.f));
        }
        st4(tmp + r * ldt + c, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// DiceLoss  utils/losses.py:33-50.  The reference first rewrites ignored pixels IN PLACE to target.min()
// (when ignore_index is not inside range(target.min(), target.max()) and at least one pixel is ignored),
// one-hot encodes, and reduces over the WHOLE batch:  1 - (2*sum(p*y) + s) / (sum(p) + sum(y) + s).
// stats = {tmin, tmax, n_ignored, remap flag} are produced on the device (no host sync); the fwd kernel
// performs the same in-place rewrite of `target` so that callers observe the reference's side effect.
__global__ __launch_bounds__(256) void target_stats_partial_kernel(const int64_t* __restrict__ target, long rows, long ignore,
                                                                   long long* __restrict__ part) {
    long long mn = 0x7fffffffffffffffLL, mx = -0x7fffffffffffffffLL - 1, cnt = 0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long long t = target[r];
        mn = t < mn ? t : mn; mx = t > mx ? t : mx; cnt += (t == ignore);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64), c = __shfl_xor(cnt, o, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx; cnt += c;
    }
    __shared__ long long sm[12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave] = mn; sm[4 + wave] = mx; sm[8 + wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = sm[w] < mn ? sm[w] : mn; mx = sm[4 + w] > mx ? sm[4 + w] : mx; }
        part[3 * blockIdx.x] = mn < sm[0] ? mn : sm[0];
        part[3 * blockIdx.x + 1] = mx > sm[4] ? mx : sm[4];
        part[3 * blockIdx.x + 2] = sm[8] + sm[9] + sm[10] + sm[11];
    }
}
__global__ void target_stats_final_kernel(const long long* __restrict__ part, int nparts, long ignore, int64_t* __restrict__ stats) {
    long long mn = 0x7fffffffffffffffLL, mx = -0x7fffffffffffffffLL - 1, cnt = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) {
        mn = part[3 * i] < mn ? part[3 * i] : mn; mx = part[3 * i + 1] > mx ? part[3 * i + 1] : mx; cnt += part[3 * i + 2];
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64), c = __shfl_xor(cnt, o, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx; cnt += c;
    }
    if (threadIdx.x == 0) {
        stats[0] = mn; stats[1] = mx; stats[2] = cnt;
        const bool in_range = ignore >= mn && ignore < mx;          // `ignore_index in range(target.min(), target.max())`
        stats[3] = (!in_range && cnt > 0) ? 1 : 0;
    }
}

// part[block] = {sum p_t, sum_c p_c, sum onehot}
__global__ __launch_bounds__(256) void dice_fwd_kernel(const float* __restrict__ logits, int ld, int64_t* __restrict__ target, long rows,
                                                       int C, long ignore, const int64_t* __restrict__ stats,
                                                       float* __restrict__ lse_out, double* __restrict__ part) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const bool remap = stats[3] != 0;
    const long tmin = stats[0];
    const int c4n = (C + 3) >> 2;
    float si = 0.f, sp = 0.f, st = 0.f;
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        long t = target[r];
        if (remap && t == ignore) {
            t = tmin;
            if (g == 0) target[r] = tmin;     // the reference's in-place rewrite (utils/losses.py:40-42)
        }
        const bool valid = t >= 0 && t < C;
        float xt;
        const float* row = logits + r * ld;
        const float lse = pixel_lse(row, C, g, valid ? t : -1, xt);
        float ps = 0.f;
        for (int q = g; q < c4n; q += LPP) {
            const float4 v = ld4(row + q * 4);
            const int c = q * 4;
            ps += expf(v.x - lse);
            if (c + 1 < C) ps += expf(v.y - lse);
            if (c + 2 < C) ps += expf(v.z - lse);
            if (c + 3 < C) ps += expf(v.w - lse);
        }
        ps = grp_sum(ps);
        if (g == 0) {
            lse_out[r] = lse;
            sp += ps;
            if (valid) { si += expf(xt - lse); st += 1.f; }
        }
    }
    si = wave_sum(si); sp = wave_sum(sp); st = wave_sum(st);
    __shared__ float sm[12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave] = si; sm[4 + wave] = sp; sm[8 + wave] = st; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = (double)sm[0] + sm[1] + sm[2] + sm[3];
        part[3 * blockIdx.x + 1] = (double)sm[4] + sm[5] + sm[6] + sm[7];
        part[3 * blockIdx.x + 2] = (double)sm[8] + sm[9] + sm[10] + sm[11];
    }
}
// sums[3] = {sum p_t, sum_c p_c, sum onehot} of this rank (the data-parallel path all-reduces them before dice_finalize)
__global__ void dice_sum_kernel(const double* __restrict__ part, int nparts, double* __restrict__ sums) {
    double i = 0.0, p = 0.0, t = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) { i += part[3 * k]; p += part[3 * k + 1]; t += part[3 * k + 2]; }
    for (int o = 32; o > 0; o >>= 1) { i += __shfl_xor(i, o, 64); p += __shfl_xor(p, o, 64); t += __shfl_xor(t, o, 64); }
    if (threadIdx.x == 0) { sums[0] = i; sums[1] = p; sums[2] = t; }
}
__global__ void dice_finalize_kernel(const double* __restrict__ part, int nparts, float smooth, float* __restrict__ out) {
    double i = 0.0, p = 0.0, t = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) { i += part[3 * k]; p += part[3 * k + 1]; t += part[3 * k + 2]; }
    for (int o = 32; o > 0; o >>= 1) { i += __shfl_xor(i, o, 64); p += __shfl_xor(p, o, 64); t += __shfl_xor(t, o, 64); }
    if (threadIdx.x == 0) {
        const double den = p + t + smooth;
        out[0] = (float)(1.0 - (2.0 * i + smooth) / den);
        out[1] = (float)i; out[2] = (float)den; out[3] = 0.f;
    }
}
// dz_c = g * (-2/den) * p_t * (delta_ct - p_c); the sum(p) term of the quotient rule vanishes (sum_c p_c == 1)
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                       const float* __restrict__ lse, long rows, int C,
                                                       const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                       float* __restrict__ dl, int lddl) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    const float A = -2.f * grad_out[0] / loss_out[2];
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t >= 0 && t < C;
        const float l = lse[r];
        const float* row = logits + r * ld;
        const float pt = valid ? expf(row[t] - l) : 0.f;
        const float k = A * pt;
        for (int q = g; q < c4n; q += LPP) {
            const float4 v = ld4(row + q * 4);
            const int c = q * 4;
            float4 d;
            d.x = k * ((t == c ? 1.f : 0.f) - expf(v.x - l));
            d.y = c + 1 < C ? k * ((t == c + 1 ? 1.f : 0.f) - expf(v.y - l)) : 0.f;
            d.z = c + 2 < C ? k * ((t == c + 2 ? 1.f : 0.f) - expf(v.z - l)) : 0.f;
            d.w = c + 3 < C ? k * ((t == c + 3 ? 1.f : 0.f) - expf(v.w - l)) : 0.f;
            st4(dl + r * lddl + q * 4, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FocalLoss  utils/losses.py:52-65 (alpha=None): ce = per-pixel cross entropy (0 at ignored pixels),
// loss = mean over ALL pixels of (1 - exp(-ce))^gamma * ce.
// alpha (class weights of the inner nn.CrossEntropyLoss(reduce=False, weight=alpha)): ce = alpha_t * (-log p_t).
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                        long rows, int C, long ignore, float gamma, const float* __restrict__ cw,
                                                        float* __restrict__ lse_out, double* __restrict__ part) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    float lsum = 0.f;
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        float xt;
        const float lse = pixel_lse(logits + r * ld, C, g, valid ? t : -1, xt);
        if (g == 0) {
            lse_out[r] = lse;
            if (valid) {
                const float ce = (cw ? cw[t] : 1.f) * (lse - xt);
                lsum += powf(1.f - expf(-ce), gamma) * ce;
            }
        }
    }
    lsum = wave_sum(lsum);
    __shared__ float sm[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = (double)sm[0] + sm[1] + sm[2] + sm[3]; part[2 * blockIdx.x + 1] = 0.0; }
}
__global__ void focal_finalize_kernel(const double* __restrict__ part, int nparts, double rows, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[2 * i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) { out[0] = (float)(s / rows); out[1] = (float)rows; out[2] = (float)s; }
}
// d loss / d z_c = g/denom * [ gamma (1-pt)^(gamma-1) pt ce + (1-pt)^gamma ] * alpha_t * (p_c - delta_ct),  pt = exp(-ce)
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                        const float* __restrict__ lse, long rows, int C, long ignore, float gamma,
                                                        const float* __restrict__ cw, const float* __restrict__ loss_out,
                                                        const float* __restrict__ grad_out, float* __restrict__ dl, int lddl) {
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    const float gs = grad_out[0] / loss_out[1];
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const long t = target[r];
        const bool valid = t != ignore && t >= 0 && t < C;
        const float l = lse[r];
        const float* row = logits + r * ld;
        float k = 0.f;
        if (valid) {
            const float a = cw ? cw[t] : 1.f;
            const float ce = a * (l - row[t]);
            const float pt = expf(-ce), om = 1.f - pt;
            const float dce = (gamma == 0.f ? 0.f : gamma * powf(om, gamma - 1.f) * pt * ce) + powf(om, gamma);
            k = gs * dce * a;
        }
        for (int q = g; q < c4n; q += LPP) {
            float4 d = zero4();
            if (valid) {
                const float4 v = ld4(row + q * 4);
                const int c = q * 4;
                d.x = k * (expf(v.x - l) - (t == c ? 1.f : 0.f));
                d.y = c + 1 < C ? k * (expf(v.y - l) - (t == c + 1 ? 1.f : 0.f)) : 0.f;
                d.z = c + 2 < C ? k * (expf(v.z - l) - (t == c + 2 ? 1.f : 0.f)) : 0.f;
                d.w = c + 3 < C ? k * (expf(v.w - l) - (t == c + 3 ? 1.f : 0.f)) : 0.f;
            }
            st4(dl + r * lddl + q * 4, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// eval_metrics  utils/metrics.py:42-67 (torch.max over classes -> first maximal index, +1 shift, labeled = 1 <= target+1 <= C,
// pixel accuracy counts and torch.histc intersection / prediction / label areas), fused into one pass over the logits and
// accumulated on the device as integers: acc = {correct, labeled, inter[C], pred_area[C], label_area[C]} (int64, exact,
// order independent).  Removes the per-iteration host synchronisations of trainer.py:84-86.
__global__ __launch_bounds__(256) void seg_metrics_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                          long rows, int C, unsigned long long* __restrict__ acc) {
    extern __shared__ unsigned hist[];   // 2 + 3*C
    const int nh = 2 + 3 * C;
    for (int i = threadIdx.x; i < nh; i += 256) hist[i] = 0;
    __syncthreads();
    const int g = threadIdx.x & (LPP - 1);
    const long ppb = 256 / LPP;
    const int c4n = (C + 3) >> 2;
    for (long r = (long)blockIdx.x * ppb + (threadIdx.x / LPP); r < rows; r += (long)gridDim.x * ppb) {
        const float* row = logits + r * ld;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = g; q < c4n; q += LPP) {
            const float4 v = ld4(row + q * 4);
            const int c = q * 4;
            if (v.x > best) { best = v.x; bi = c; }
            if (c + 1 < C && v.y > best) { best = v.y; bi = c + 1; }
            if (c + 2 < C && v.z > best) { best = v.z; bi = c + 2; }
            if (c + 3 < C && v.w > best) { best = v.w; bi = c + 3; }
        }
#pragma unroll
        for (int o = 1; o < LPP; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (g == 0) {
            const long t = target[r] + 1;               // labels shifted to 1..C like the reference
            const int pred = bi + 1;
            if (t > 0 && t <= C) {
                atomicAdd(&hist[1], 1u);
                atomicAdd(&hist[2 + C + (pred - 1)], 1u);
                atomicAdd(&hist[2 + 2 * C + (int)(t - 1)], 1u);
                if (pred == t) { atomicAdd(&hist[0], 1u); atomicAdd(&hist[2 + (pred - 1)], 1u); }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nh; i += 256)
        if (hist[i]) atomicAdd(&acc[i], (unsigned long long)hist[i]);
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

int segmi_ce_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index,
                 const float* class_weight, float* lse, float* loss_out, void* workspace, size_t workspace_bytes,
                 segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_ce_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, ld, target, rows, C, ignore_index, class_weight, lse,
                       (double*)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, loss_out);
    return segmi_launch_status();
}

int segmi_ce_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C,
                 long ignore_index, const float* class_weight, const float* loss_out, const float* grad_out, float* dlogits,
                 int lddl, segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(rows)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, rows, C,
                       ignore_index, class_weight, loss_out, grad_out, dlogits, lddl);
    return segmi_launch_status();
}

size_t segmi_upsample_ce_workspace(int N, int H, int W, int C, int OH, int OW) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return 0;
    const size_t a = segmi_ce_workspace((long)N * OH * OW);
    const size_t b = (size_t)N * OH * W * ((C + 3) & ~3) * sizeof(float);     // pass-W buffer of the backward
    return a > b ? a : b;
}

int segmi_upsample_ce_fwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                          const int64_t* target, long ignore_index, const float* class_weight, float* lse, float* loss_out,
                          void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits_lo || !target || !lse || !loss_out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    const long rows = (long)N * OH * OW;
    if (!workspace || workspace_bytes < segmi_ce_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    if (C <= 32)
        hipLaunchKernelGGL(upce_fwd_small_kernel, dim3(nb), dim3(256), 0, st, logits_lo, ld, N, H, W, C, OH, OW, align_corners ? 1 : 0, target,
                           ignore_index, class_weight, lse, (double*)workspace);
    else
        hipLaunchKernelGGL(upce_fwd_kernel, dim3(nb), dim3(256), 0, st, logits_lo, ld, N, H, W, C, OH, OW, align_corners ? 1 : 0, target,
                           ignore_index, class_weight, lse, (double*)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, loss_out);
    return segmi_launch_status();
}

int segmi_upsample_ce_bwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                          const int64_t* target, const float* lse, long ignore_index, const float* class_weight,
                          const float* loss_out, const float* grad_out, float* dlogits_lo, int lddl, void* workspace,
                          size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits_lo || !target || !lse || !loss_out || !grad_out || !dlogits_lo || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0)
        return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_upsample_ce_workspace(N, H, W, C, OH, OW) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int ldt = (C + 3) & ~3, ac = align_corners ? 1 : 0;
    float* tmp = (float*)workspace;
    RowGeom g0 = row_geom_dense((long)N * OH * W, C, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(upce_bwd_w_kernel, g0.grid, g0.block, 0, st, logits_lo, ld, N, H, W, C, OH, OW, ac, target, lse, ignore_index,
                       class_weight, loss_out, grad_out, tmp, ldt);
    return segmi_internal_bilinear_bwd_height(tmp, ldt, dlogits_lo, lddl, N, H, W, C, OH, OW, ac, st);
}

/* workspace: 3 doubles per block (dice partials) + 3 int64 per block (target statistics) */
size_t segmi_dice_workspace(long rows) { return (size_t)ce_blocks(rows) * 3 * (sizeof(double) + sizeof(long long)); }

int segmi_target_stats(const int64_t* target, long rows, long ignore_index, int64_t* stats, void* workspace, size_t workspace_bytes,
                       segmi_stream_t stream) {
    if (!target || !stats || rows <= 0) return SEGMI_ERR_BADARG;
    if (!workspace || workspace_bytes < segmi_dice_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    long long* ipart = (long long*)((double*)workspace + 3 * (size_t)nb);
    hipLaunchKernelGGL(target_stats_partial_kernel, dim3(nb), dim3(256), 0, st, target, rows, ignore_index, ipart);
    hipLaunchKernelGGL(target_stats_final_kernel, dim3(1), dim3(64), 0, st, (const long long*)ipart, nb, ignore_index, stats);
    return segmi_launch_status();
}

int segmi_dice_sums(const float* logits, int ld, int64_t* target, long rows, int C, long ignore_index, const int64_t* stats,
                    float* lse, double* sums, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits || !target || !stats || !lse || !sums || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_dice_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    double* dpart = (double*)workspace;
    hipLaunchKernelGGL(dice_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, ld, target, rows, C, ignore_index, stats, lse, dpart);
    hipLaunchKernelGGL(dice_sum_kernel, dim3(1), dim3(64), 0, st, (const double*)dpart, nb, sums);
    return segmi_launch_status();
}

int segmi_dice_finalize(const double* sums, float smooth, float* loss_out, segmi_stream_t stream) {
    if (!sums || !loss_out) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, 1, smooth, loss_out);
    return segmi_launch_status();
}

int segmi_dice_fwd(const float* logits, int ld, int64_t* target, long rows, int C, long ignore_index, float smooth,
                   int64_t* stats, float* lse, float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits || !target || !stats || !lse || !loss_out || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_dice_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    double* dpart = (double*)workspace;
    int rc = segmi_target_stats(target, rows, ignore_index, stats, workspace, workspace_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(dice_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, ld, target, rows, C, ignore_index, (const int64_t*)stats, lse, dpart);
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)dpart, nb, smooth, loss_out);
    return segmi_launch_status();
}

int segmi_dice_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C, const float* loss_out,
                   const float* grad_out, float* dlogits, int lddl, segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(ce_blocks(rows)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, rows, C,
                       loss_out, grad_out, dlogits, lddl);
    return segmi_launch_status();
}

int segmi_focal_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float gamma,
                    const float* alpha, float* lse, float* loss_out, void* workspace, size_t workspace_bytes,
                    segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || rows <= 0 || C <= 0 || !(gamma >= 0.f)) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_ce_workspace(rows)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(rows);
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, ld, target, rows, C, ignore_index, gamma, alpha, lse,
                       (double*)workspace);
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, (double)rows, loss_out);
    return segmi_launch_status();
}

int segmi_focal_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C, long ignore_index,
                    float gamma, const float* alpha, const float* loss_out, const float* grad_out, float* dlogits, int lddl,
                    segmi_stream_t stream) {
    if (!logits || !target || !lse || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(ce_blocks(rows)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, rows, C,
                       ignore_index, gamma, alpha, loss_out, grad_out, dlogits, lddl);
    return segmi_launch_status();
}

int segmi_seg_metrics(const float* logits, int ld, const int64_t* target, long rows, int C, int64_t* acc, segmi_stream_t stream) {
    if (!logits || !target || !acc || rows <= 0 || C <= 0 || C > 4000) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    hipLaunchKernelGGL(seg_metrics_kernel, dim3(ce_blocks(rows)), dim3(256), (size_t)(2 + 3 * C) * sizeof(unsigned), (hipStream_t)stream,
                       logits, ld, target, rows, C, (unsigned long long*)acc);
    return segmi_launch_status();
}

}  // extern "C"
