// Max pooling, adaptive average pooling (PSP pyramid / ASPP image pool) and bilinear resize,
// forward and backward, NHWC fp32.  All backward kernels are written in gather form (each input
// element collects from the outputs that reference it): deterministic, no atomics.
//
// Replaces (reference call sites):
//   aten::max_pool2d_with_indices(+bwd)  models/resnet.py:151, models/deeplabv3_plus.py:24, models/unet.py:27
//   aten::adaptive_avg_pool2d(+bwd)      models/pspnet.py:26, models/deeplabv3_plus.py:274
//   aten::upsample_bilinear2d(+bwd)      models/pspnet.py:35-36,86,91; models/deeplabv3_plus.py:291,328,361;
//                                        models/unet.py:46-47
#include "rowgeom.h"
#include "bilinear.h"

namespace {

// ------------------------------------------------------------------------------------ max pool
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C, int P, int Q,
                                                          int k, int stride, int pad) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const long rows = (long)N * P * Q;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        const int q = (int)(r % Q);
        const long t = r / Q;
        const int p = (int)(t % P), n = (int)(t / P);
        const int h0 = p * stride - pad, w0 = q * stride - pad;
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        bool first = true;
        for (int a = 0; a < k; ++a) {
            const int h = h0 + a;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int b = 0; b < k; ++b) {
                const int w = w0 + b;
                if ((unsigned)w >= (unsigned)W) continue;
                const float4 v = ld4(x + ((long)(n * H + h) * W + w) * ldx + c4 * 4);
                const uint8_t tap = (uint8_t)(a * k + b);
                // aten (cpu): take the first maximum; NaN wins
                if (first || v.x > best.x || v.x != v.x) { best.x = v.x; bi.x = tap; }
                if (first || v.y > best.y || v.y != v.y) { best.y = v.y; bi.y = tap; }
                if (first || v.z > best.z || v.z != v.z) { best.z = v.z; bi.z = tap; }
                if (first || v.w > best.w || v.w != v.w) { best.w = v.w; bi.w = tap; }
                first = false;
            }
        }
        st4(y + r * ldy + c4 * 4, best);
        *reinterpret_cast<uchar4*>(idx + r * (long)(c4n * 4) + c4 * 4) = bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int lddx, int N, int H, int W, int C, int P,
                                                          int Q, int k, int stride, int pad) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const long rows = (long)N * H * W;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        const int w = (int)(r % W);
        const long t = r / W;
        const int h = (int)(t % H), n = (int)(t / H);
        // outputs p with p*stride - pad <= h <= p*stride - pad + k - 1
        int p_lo = h + pad - k + 1; p_lo = p_lo <= 0 ? 0 : (p_lo + stride - 1) / stride;
        int p_hi = (h + pad) / stride; if (p_hi > P - 1) p_hi = P - 1;
        int q_lo = w + pad - k + 1; q_lo = q_lo <= 0 ? 0 : (q_lo + stride - 1) / stride;
        int q_hi = (w + pad) / stride; if (q_hi > Q - 1) q_hi = Q - 1;
        float4 acc = zero4();
        for (int p = p_lo; p <= p_hi; ++p) {
            const int a = h + pad - p * stride;
            for (int q = q_lo; q <= q_hi; ++q) {
                const int b = w + pad - q * stride;
                const uint8_t tap = (uint8_t)(a * k + b);
                const long o = (long)(n * P + p) * Q + q;
                const uchar4 bi = *reinterpret_cast<const uchar4*>(idx + o * (long)(c4n * 4) + c4 * 4);
                const float4 g = ld4(dy + o * lddy + c4 * 4);
                if (bi.x == tap) acc.x += g.x;
                if (bi.y == tap) acc.y += g.y;
                if (bi.z == tap) acc.z += g.z;
                if (bi.w == tap) acc.w += g.w;
            }
        }
        st4(dx + r * lddx + c4 * 4, acc);
    }
}

// ------------------------------------------------------------------------- adaptive average pool
__device__ __forceinline__ int aap_start(int o, int in, int out) { return (int)(((long)o * in) / out); }
__device__ __forceinline__ int aap_end(int o, int in, int out) { return (int)(((long)(o + 1) * in + out - 1) / out); }

// one workgroup per (output bin, channel tile): the ry row-lanes stride over the window pixels
__global__ __launch_bounds__(256) void aap_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N,
                                                      int H, int W, int C, int OH, int OW) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.y * blockDim.x + threadIdx.x;
    const bool cok = c4 < c4n;
    const int bin = blockIdx.x;
    const int ow = bin % OW, t = bin / OW, oh = t % OH, n = t / OH;
    const int h0 = aap_start(oh, H, OH), h1 = aap_end(oh, H, OH);
    const int w0 = aap_start(ow, W, OW), w1 = aap_end(ow, W, OW);
    const int ww = w1 - w0, cnt = (h1 - h0) * ww;
    float4 acc = zero4();
    if (cok)
        for (int i = threadIdx.y; i < cnt; i += blockDim.y) {
            const int h = h0 + i / ww, w = w0 + i % ww;
            const float4 v = ld4(x + ((long)(n * H + h) * W + w) * ldx + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    __shared__ float4 sm[256];
    const int tix = threadIdx.y * blockDim.x + threadIdx.x;
    sm[tix] = acc;
    __syncthreads();
    for (int s = blockDim.y >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.y < s) {
            float4 a = sm[tix], b = sm[tix + s * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; sm[tix] = a;
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        float4 a = sm[tix];
        const float inv = 1.f / (float)cnt;
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        st4(y + (long)bin * ldy + c4 * 4, a);
    }
}

template <bool ACC>
__global__ __launch_bounds__(256) void aap_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                      int N, int H, int W, int C, int OH, int OW) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const long rows = (long)N * H * W;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        const int w = (int)(r % W);
        const long t = r / W;
        const int h = (int)(t % H), n = (int)(t / H);
        const int ohc = (int)(((long)h * OH) / H), owc = (int)(((long)w * OW) / W);
        float4 acc = ACC ? ld4(dx + r * lddx + c4 * 4) : zero4();
        const int ohe = min(OH - 1, (int)(((long)(h + 1) * OH + H - 1) / H));
        const int owe = min(OW - 1, (int)(((long)(w + 1) * OW + W - 1) / W));
        for (int oh = max(ohc - 1, 0); oh <= ohe; ++oh) {
            const int h0 = aap_start(oh, H, OH), h1 = aap_end(oh, H, OH);
            if (h < h0 || h >= h1) continue;
            for (int ow = max(owc - 1, 0); ow <= owe; ++ow) {
                const int w0 = aap_start(ow, W, OW), w1 = aap_end(ow, W, OW);
                if (w < w0 || w >= w1) continue;
                const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                const float4 g = ld4(dy + ((long)(n * OH + oh) * OW + ow) * lddy + c4 * 4);
                acc.x += g.x * inv; acc.y += g.y * inv; acc.z += g.z * inv; acc.w += g.w * inv;
            }
        }
        st4(dx + r * lddx + c4 * 4, acc);
    }
}

// ------------------------------------------------------------------------------------ bilinear
// (source-coordinate helpers Lerp / bl_scale / bl_src / bl_range: bilinear.h, shared with the fused upsample+loss kernels)
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                           int N, int H, int W, int C, int OH, int OW, int ac) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const float sh = bl_scale(H, OH, ac), sw = bl_scale(W, OW, ac);
    const long rows = (long)N * OH * OW;
    const long r0 = (long)blockIdx.y * blockDim.y + threadIdx.y, stride = (long)gridDim.y * blockDim.y;
    RowWalk3 rw;
    rw.init(r0, stride, OH, OW);
    for (long r = r0; r < rows; r += stride, rw.step()) {
        const int ow = rw.b, oh = rw.a, n = rw.n;
        const Lerp a = bl_src(oh, sh, H, ac), b = bl_src(ow, sw, W, ac);
        const float* base = x + (long)n * H * W * ldx + c4 * 4;
        const float4 v00 = ld4(base + ((long)a.i0 * W + b.i0) * ldx), v01 = ld4(base + ((long)a.i0 * W + b.i1) * ldx);
        const float4 v10 = ld4(base + ((long)a.i1 * W + b.i0) * ldx), v11 = ld4(base + ((long)a.i1 * W + b.i1) * ldx);
        float4 o;
        o.x = a.l0 * (b.l0 * v00.x + b.l1 * v01.x) + a.l1 * (b.l0 * v10.x + b.l1 * v11.x);
        o.y = a.l0 * (b.l0 * v00.y + b.l1 * v01.y) + a.l1 * (b.l0 * v10.y + b.l1 * v11.y);
        o.z = a.l0 * (b.l0 * v00.z + b.l1 * v01.z) + a.l1 * (b.l0 * v10.z + b.l1 * v11.z);
        o.w = a.l0 * (b.l0 * v00.w + b.l1 * v01.w) + a.l1 * (b.l0 * v10.w + b.l1 * v11.w);
        st4(y + r * ldy + c4 * 4, o);
    }
}

// Backward = exact transpose of the forward in GATHER form (deterministic, no atomics), done separably:
//   pass W: tmp[n, oh, w, c] = sum_ow ww(ow -> w) * dy[n, oh, ow, c]      (reads dy once, contiguous pixel runs)
//   pass H: dx [n, h,  w, c] = sum_oh wh(oh -> h) * tmp[n, oh, w, c]
// (the one-pass form evaluated ~400 candidate weights per input pixel for an 8x upsample and ran 10x off the HBM roofline).
// AXIS 0: reduce along the width (src [n*OH+oh][OW] -> dst [..][W]); AXIS 1: along the height.
template <int AXIS>
__global__ __launch_bounds__(256) void bilinear_bwd_axis_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                                int N, int H, int W, int C, int OH, int OW, int ac) {
    const int c4n = (C + 3) / 4;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const long rows = AXIS == 0 ? (long)N * OH * W : (long)N * H * W;
    const float scale = AXIS == 0 ? bl_scale(W, OW, ac) : bl_scale(H, OH, ac);
    const long r0 = (long)blockIdx.y * blockDim.y + threadIdx.y, stride = (long)gridDim.y * blockDim.y;
    RowWalk3 rw;                                  // r = (n * A + a) * W + w with A = OH (AXIS 0) or H (AXIS 1)
    rw.init(r0, stride, AXIS == 0 ? OH : H, W);
    for (long r = r0; r < rows; r += stride, rw.step()) {
        const int w = rw.b;
        const long t = (long)rw.n * rw.A + rw.a;  // AXIS 0: n*OH + oh ; AXIS 1: n*H + h
        int lo, hi;
        float4 acc = zero4();
        if (AXIS == 0) {
            bl_range(w, scale, W, OW, ac, lo, hi);
            const float* base = src + t * OW * lds + c4 * 4;
            for (int ow = lo; ow <= hi; ++ow) {
                const Lerp b = bl_src(ow, scale, W, ac);
                const float wt = (b.i0 == w ? b.l0 : 0.f) + (b.i1 == w ? b.l1 : 0.f);
                if (wt != 0.f) {
                    const float4 g = ld4(base + (long)ow * lds);
                    acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
                }
            }
        } else {
            const int h = rw.a;
            const long n = rw.n;
            bl_range(h, scale, H, OH, ac, lo, hi);
            const float* base = src + (n * OH * W + w) * lds + c4 * 4;
            for (int oh = lo; oh <= hi; ++oh) {
                const Lerp a = bl_src(oh, scale, H, ac);
                const float wt = (a.i0 == h ? a.l0 : 0.f) + (a.i1 == h ? a.l1 : 0.f);
                if (wt != 0.f) {
                    const float4 g = ld4(base + (long)oh * W * lds);
                    acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
                }
            }
        }
        st4(dst + r * ldd + c4 * 4, acc);
    }
}

// ------------------------------------------------------------------------------------ fused pyramid pooling
// The PSP module (models/pspnet.py:25-37) pools the SAME [N, 2048, 64, 64] map (268 MB at cfg2) with bins 1, 2, 3 and 6: four
// reads forward, and backward four full-size gradient maps that autograd then adds.  Fused: the union of all window
// boundaries cuts the map into <= 24 x 24 cells; pass 1 reduces x to per-cell sums in ONE read, pass 2 assembles every bin of
// every pyramid level from cells (a few KB); backward writes dx ONCE from the four small dy tensors.
constexpr int PYR_MAX_LEVELS = 4, PYR_MAX_BIN = 8, PYR_MAX_SEG = 2 * PYR_MAX_LEVELS * PYR_MAX_BIN + 1;
struct PyrAxis {
    int nseg;
    int brk[PYR_MAX_SEG + 1];                       // segment k = [brk[k], brk[k+1])
    int lo[PYR_MAX_LEVELS][PYR_MAX_BIN], hi[PYR_MAX_LEVELS][PYR_MAX_BIN];   // window i of level l = segments [lo, hi)
};
struct PyrGeom {
    int N, H, W, C, nl;
    int bins[PYR_MAX_LEVELS];
    PyrAxis ah, aw;
};
struct PyrPtrs {
    float* y[PYR_MAX_LEVELS];
    const float* dy[PYR_MAX_LEVELS];
    int ld[PYR_MAX_LEVELS];
};

// cells[n][sh][sw][C] = sum over the cell's pixels; grid (channel tiles, nsh*nsw, N)
__global__ __launch_bounds__(256) void pyramid_cells_kernel(const float* __restrict__ x, int ldx, float* __restrict__ cells, PyrGeom g) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 * 4 < g.C;
    const int sh = blockIdx.y / g.aw.nseg, sw = blockIdx.y % g.aw.nseg, n = blockIdx.z;
    const int h0 = g.ah.brk[sh], h1 = g.ah.brk[sh + 1], w0 = g.aw.brk[sw], w1 = g.aw.brk[sw + 1];
    const int cw = w1 - w0, cnt = (h1 - h0) * cw;
    float4 acc = zero4();
    if (cok)
        for (int i = threadIdx.y; i < cnt; i += blockDim.y) {
            const int h = h0 + i / cw, w = w0 + i % cw;
            const float4 v = ld4(x + ((long)(n * g.H + h) * g.W + w) * ldx + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    __shared__ float4 sm[256];
    const int tix = threadIdx.y * blockDim.x + threadIdx.x;
    sm[tix] = acc;
    __syncthreads();
    for (int s = blockDim.y >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.y < s) {
            float4 a = sm[tix], b = sm[tix + s * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            sm[tix] = a;
        }
        __syncthreads();
    }
    const int Cp = (g.C + 3) & ~3;
    if (threadIdx.y == 0 && cok) st4(cells + ((long)(n * g.ah.nseg + sh) * g.aw.nseg + sw) * Cp + c4 * 4, sm[tix]);
}

// y_l[n, i, j, c] = (sum of the cells of window (i, j) of level l) / window area; grid (channel tiles, sum_l b_l^2, N)
__global__ __launch_bounds__(256) void pyramid_assemble_kernel(const float* __restrict__ cells, PyrGeom g, PyrPtrs p) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= g.C || threadIdx.y) return;
    int item = blockIdx.y, l = 0;
    while (item >= g.bins[l] * g.bins[l]) { item -= g.bins[l] * g.bins[l]; ++l; }
    const int b = g.bins[l], i = item / b, j = item % b, n = blockIdx.z;
    const int Cp = (g.C + 3) & ~3;
    float4 acc = zero4();
    for (int sh = g.ah.lo[l][i]; sh < g.ah.hi[l][i]; ++sh)
        for (int sw = g.aw.lo[l][j]; sw < g.aw.hi[l][j]; ++sw) {
            const float4 v = ld4(cells + ((long)(n * g.ah.nseg + sh) * g.aw.nseg + sw) * Cp + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    const float inv = 1.f / (float)((g.ah.brk[g.ah.hi[l][i]] - g.ah.brk[g.ah.lo[l][i]]) * (g.aw.brk[g.aw.hi[l][j]] - g.aw.brk[g.aw.lo[l][j]]));
    st4(p.y[l] + ((long)(n * b + i) * b + j) * p.ld[l] + c4 * 4, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
}

// dx[n,h,w,c] = sum_l sum_{windows (i,j) of level l containing (h,w)} dy_l[n,i,j,c] / area.  All pixels of a cell (the map cut at
// every window boundary of every level: <= 24 x 24 cells) receive the SAME value, so it is computed once per (n, cell, channel)
// — the nested window search — and the full-size pass only looks its cell up and streams the gradient out (a per-pixel search
// made this write-only kernel instruction-bound: 250 us for 268 MB at cfg2, 13 % of the HBM rate).
__global__ __launch_bounds__(256) void pyramid_bwd_cells_kernel(float* __restrict__ cellgrad, PyrGeom g, PyrPtrs p) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= g.C || threadIdx.y) return;
    const int cell = blockIdx.y, n = blockIdx.z;
    const int sh = cell / g.aw.nseg, sw = cell - sh * g.aw.nseg;
    float4 acc = zero4();
    for (int l = 0; l < g.nl; ++l) {
        const int b = g.bins[l];
        for (int i = 0; i < b; ++i) {
            if (sh < g.ah.lo[l][i] || sh >= g.ah.hi[l][i]) continue;
            const int hh = g.ah.brk[g.ah.hi[l][i]] - g.ah.brk[g.ah.lo[l][i]];
            for (int j = 0; j < b; ++j) {
                if (sw < g.aw.lo[l][j] || sw >= g.aw.hi[l][j]) continue;
                const float inv = 1.f / (float)(hh * (g.aw.brk[g.aw.hi[l][j]] - g.aw.brk[g.aw.lo[l][j]]));
                const float4 v = ld4(p.dy[l] + ((long)(n * b + i) * b + j) * p.ld[l] + c4 * 4);
                acc.x += v.x * inv; acc.y += v.y * inv; acc.z += v.z * inv; acc.w += v.w * inv;
            }
        }
    }
    const int Cp = (g.C + 3) & ~3;
    st4(cellgrad + ((long)(n * g.ah.nseg + sh) * g.aw.nseg + sw) * Cp + c4 * 4, acc);
}

__global__ __launch_bounds__(256) void pyramid_bwd_expand_kernel(const float* __restrict__ cellgrad, float* __restrict__ dx, int lddx, PyrGeom g) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= g.C) return;
    const int Cp = (g.C + 3) & ~3;
    const long rows = (long)g.N * g.H * g.W;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        const int w = (int)(r % g.W);
        const long t = r / g.W;
        const int h = (int)(t % g.H), n = (int)(t / g.H);
        int sh = 0, sw = 0;                                  // the pixel's cell (no divisions: <= 24 breakpoints per axis)
        while (g.ah.brk[sh + 1] <= h) ++sh;
        while (g.aw.brk[sw + 1] <= w) ++sw;
        st4(dx + r * lddx + c4 * 4, ld4(cellgrad + ((long)(n * g.ah.nseg + sh) * g.aw.nseg + sw) * Cp + c4 * 4));
    }
}

static bool pyr_axis(int in, int nl, const int* bins, PyrAxis* a) {
    int pts[PYR_MAX_SEG + 1], np = 0;
    for (int l = 0; l < nl; ++l)
        for (int i = 0; i < bins[l]; ++i) {
            pts[np++] = (int)(((long)i * in) / bins[l]);
            pts[np++] = (int)(((long)(i + 1) * in + bins[l] - 1) / bins[l]);
        }
    for (int i = 1; i < np; ++i) {                 // insertion sort + unique (<= 64 points)
        int v = pts[i], j = i - 1;
        while (j >= 0 && pts[j] > v) { pts[j + 1] = pts[j]; --j; }
        pts[j + 1] = v;
    }
    int nu = 0;
    for (int i = 0; i < np; ++i)
        if (nu == 0 || pts[i] != a->brk[nu - 1]) a->brk[nu++] = pts[i];
    a->nseg = nu - 1;
    if (a->nseg < 1 || a->brk[0] != 0 || a->brk[nu - 1] != in) return false;
    for (int l = 0; l < nl; ++l)
        for (int i = 0; i < bins[l]; ++i) {
            const int s = (int)(((long)i * in) / bins[l]), e = (int)(((long)(i + 1) * in + bins[l] - 1) / bins[l]);
            int lo = 0, hi = 0;
            while (a->brk[lo] != s) ++lo;
            hi = lo;
            while (a->brk[hi] != e) ++hi;
            a->lo[l][i] = lo; a->hi[l][i] = hi;
        }
    return true;
}
static bool pyr_geom(int N, int H, int W, int C, int nl, const int* bins, PyrGeom* g) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || nl <= 0 || nl > PYR_MAX_LEVELS || !bins) return false;
    for (int l = 0; l < nl; ++l)
        if (bins[l] <= 0 || bins[l] > PYR_MAX_BIN || bins[l] > H || bins[l] > W) return false;
    g->N = N; g->H = H; g->W = W; g->C = C; g->nl = nl;
    for (int l = 0; l < PYR_MAX_LEVELS; ++l) g->bins[l] = l < nl ? bins[l] : 0;
    return pyr_axis(H, nl, bins, &g->ah) && pyr_axis(W, nl, bins, &g->aw);
}

bool ldok(int ld, int C) { return ld >= ((C + 3) & ~3) && (ld & 3) == 0; }

}  // namespace

extern "C" {

int segmi_maxpool_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int N, int H, int W, int C, int P,
                      int Q, int k, int stride, int pad, segmi_stream_t stream) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0)
        return SEGMI_ERR_BADARG;
    if (!ldok(ldx, C) || !ldok(ldy, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom((long)N * P * Q, C, 2, SEGMI_MAX_GRID * 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, idx, N, H, W, C, P, Q, k, stride, pad);
    return segmi_launch_status();
}

int segmi_maxpool_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int N, int H, int W, int C,
                      int P, int Q, int k, int stride, int pad, segmi_stream_t stream) {
    if (!dy || !dx || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0)
        return SEGMI_ERR_BADARG;
    if (!ldok(lddy, C) || !ldok(lddx, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom((long)N * H * W, C, 2, SEGMI_MAX_GRID * 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, dy, lddy, idx, dx, lddx, N, H, W, C, P, Q, k, stride, pad);
    return segmi_launch_status();
}

int segmi_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH, int OW,
                               segmi_stream_t stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    if ((long)N * OH * OW > 0x7fffffffL) return SEGMI_ERR_BADARG;
    if (!ldok(ldx, C) || !ldok(ldy, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom(1, C, 1, 1);
    g.grid = dim3((unsigned)(N * OH * OW), g.grid.x);  // item on x (2^31 limit), channel tile on y
    hipLaunchKernelGGL(aap_fwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, N, H, W, C, OH, OW);
    return segmi_launch_status();
}

int segmi_adaptive_avgpool_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int OH,
                               int OW, int accumulate, segmi_stream_t stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    if (!ldok(lddy, C) || !ldok(lddx, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom((long)N * H * W, C, 2, SEGMI_MAX_GRID * 4);
    if (accumulate) hipLaunchKernelGGL((aap_bwd_kernel<true>), g.grid, g.block, 0, (hipStream_t)stream, dy, lddy, dx, lddx, N, H, W, C, OH, OW);
    else            hipLaunchKernelGGL((aap_bwd_kernel<false>), g.grid, g.block, 0, (hipStream_t)stream, dy, lddy, dx, lddx, N, H, W, C, OH, OW);
    return segmi_launch_status();
}

int segmi_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH, int OW,
                       int align_corners, segmi_stream_t stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    if (!ldok(ldx, C) || !ldok(ldy, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom_dense((long)N * OH * OW, C, 2, SEGMI_MAX_GRID * 4);
    hipLaunchKernelGGL(bilinear_fwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, N, H, W, C, OH, OW, align_corners ? 1 : 0);
    return segmi_launch_status();
}

}  // extern "C"

// Height pass of the separable bilinear backward on a width-reduced buffer tmp[N, OH, W, ldt] — internal (C++ linkage, not part
// of the C ABI): the fused upsample + cross-entropy backward of loss.hip produces tmp itself.
int segmi_internal_bilinear_bwd_height(const float* tmp, int ldt, float* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                                       int ac, hipStream_t st) {
    RowGeom g1 = row_geom_dense((long)N * H * W, C, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL((bilinear_bwd_axis_kernel<1>), g1.grid, g1.block, 0, st, tmp, ldt, dx, lddx, N, H, W, C, OH, OW, ac);
    return segmi_launch_status();
}

extern "C" {

size_t segmi_bilinear_bwd_workspace(int N, int H, int W, int C, int OH, int OW) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return 0;
    return (size_t)N * OH * W * ((C + 3) & ~3) * sizeof(float);
}

int segmi_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                       int align_corners, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    if (!ldok(lddy, C) || !ldok(lddx, C)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_bilinear_bwd_workspace(N, H, W, C, OH, OW) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    const int ldt = (C + 3) & ~3;
    float* tmp = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int ac = align_corners ? 1 : 0;
    RowGeom g0 = row_geom_dense((long)N * OH * W, C, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL((bilinear_bwd_axis_kernel<0>), g0.grid, g0.block, 0, st, dy, lddy, tmp, ldt, N, H, W, C, OH, OW, ac);
    RowGeom g1 = row_geom_dense((long)N * H * W, C, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL((bilinear_bwd_axis_kernel<1>), g1.grid, g1.block, 0, st, (const float*)tmp, ldt, dx, lddx, N, H, W, C, OH, OW, ac);
    return segmi_launch_status();
}

size_t segmi_pyramid_pool_workspace(int N, int H, int W, int C, int nlevels, const int* bins) {
    PyrGeom g;
    if (!pyr_geom(N, H, W, C, nlevels, bins, &g)) return 0;
    return (size_t)N * g.ah.nseg * g.aw.nseg * ((C + 3) & ~3) * sizeof(float);
}

int segmi_pyramid_pool_fwd(const float* x, int ldx, int N, int H, int W, int C, int nlevels, const int* bins, float* const* y,
                           const int* ldy, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    PyrGeom g;
    if (!x || !y || !ldy || !pyr_geom(N, H, W, C, nlevels, bins, &g) || N > 65535) return SEGMI_ERR_BADARG;
    if (!ldok(ldx, C)) return SEGMI_ERR_ALIGN;
    PyrPtrs p;
    int items = 0;
    for (int l = 0; l < PYR_MAX_LEVELS; ++l) {
        p.y[l] = l < nlevels ? y[l] : nullptr; p.dy[l] = nullptr; p.ld[l] = l < nlevels ? ldy[l] : 0;
        if (l < nlevels) {
            if (!y[l]) return SEGMI_ERR_BADARG;
            if (!ldok(ldy[l], C)) return SEGMI_ERR_ALIGN;
            items += bins[l] * bins[l];
        }
    }
    if (!workspace || workspace_bytes < segmi_pyramid_pool_workspace(N, H, W, C, nlevels, bins) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RowGeom rg = row_geom(1, C, 1, 1);
    hipLaunchKernelGGL(pyramid_cells_kernel, dim3(rg.grid.x, (unsigned)(g.ah.nseg * g.aw.nseg), (unsigned)N), rg.block, 0, st, x, ldx, (float*)workspace, g);
    hipLaunchKernelGGL(pyramid_assemble_kernel, dim3(rg.grid.x, (unsigned)items, (unsigned)N), rg.block, 0, st, (const float*)workspace, g, p);
    return segmi_launch_status();
}

int segmi_pyramid_pool_bwd(const float* const* dy, const int* lddy, float* dx, int lddx, int N, int H, int W, int C, int nlevels,
                           const int* bins, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    PyrGeom g;
    if (!dy || !lddy || !dx || !pyr_geom(N, H, W, C, nlevels, bins, &g) || N > 65535) return SEGMI_ERR_BADARG;
    if (!ldok(lddx, C)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_pyramid_pool_workspace(N, H, W, C, nlevels, bins) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    PyrPtrs p;
    for (int l = 0; l < PYR_MAX_LEVELS; ++l) {
        p.y[l] = nullptr; p.dy[l] = l < nlevels ? dy[l] : nullptr; p.ld[l] = l < nlevels ? lddy[l] : 0;
        if (l < nlevels && (!dy[l] || !ldok(lddy[l], C))) return dy[l] ? SEGMI_ERR_ALIGN : SEGMI_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    RowGeom cg = row_geom(1, C, 1, 1);
    hipLaunchKernelGGL(pyramid_bwd_cells_kernel, dim3(cg.grid.x, (unsigned)(g.ah.nseg * g.aw.nseg), (unsigned)N), cg.block, 0, st, (float*)workspace, g, p);
    RowGeom rg = row_geom((long)N * H * W, C, 2, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(pyramid_bwd_expand_kernel, rg.grid, rg.block, 0, st, (const float*)workspace, dx, lddx, g);
    return segmi_launch_status();
}

}  // extern "C"
