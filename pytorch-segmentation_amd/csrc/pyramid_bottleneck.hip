// Factored form of the PSP bottleneck convolution (models/pspnet.py:25-38 of the reference).
//
// The reference upsamples the four pyramid branches p_b [N, C/4, b, b] (b = 1, 2, 3, 6) bilinearly (align_corners=True) to the
// feature size, concatenates them with the features (2048 + 4*512 = 4096 channels) and runs ONE 3x3 convolution 4096 -> 512
// over the 64x64 map: 618 GMAC per cfg2 step, 38 % of the model — and HALF of it multiplies upsampled copies of at most 36
// distinct pixels.  Both operations are linear, so for a pyramid branch
//
//   conv3x3(up(p))[h, w, k] = sum_{r,s} sum_{(i,j)} B[(h+r-1, w+s-1) -> (i,j)] * T[(i,j), (r,s), k],
//   T[(i,j), (r,s), k]      = sum_c W[k, r, s, c] * p[(i,j), c]                      (a [N*b*b, C/4] x [C/4, 9K] GEMM)
//
// with B the bilinear weights (zero when the tap falls outside the map: the convolution's zero padding).  The 77 GMAC of each
// branch become a 0.02-0.7 GMAC GEMM on the MFMA path (segmi_conv2d_fwd as a 1x1 convolution with 9K output channels) plus the
// separable interpolation below; the gradient w.r.t. T is the transposed interpolation of dy, from which dp and dW follow as
// the dgrad / wgrad of that same 1x1 convolution.  The convolution proper only sees the 2048 feature channels.  Exact in real
// arithmetic; in fp32 it is a different summation order, like any tiling change (parity: tests/test_pspnet_gpu.py and the
// BASELINE-shape audits run through it).  The upsampled branches and the 4096-channel concat buffer (537 MB at cfg2) are never
// materialised.
//
// Kernels here (HBM/L2-bound, one float4 of output channels per thread, separable in h / w):
//   pyr_up_w   V[n, (ii, r), w, k] = sum_{s, corner j} Bw[(w+s-1) -> j] * T_stage[n, i, j, (r, s), k]         ii = row node of any stage
//   pyr_up_h   y[n, h, w, k]       = sum_stage sum_{r, corner i} Bh[(h+r-1) -> i] * V[n, (ii, r), w, k]
//   pyr_dn_w   U[n, h, (jj, s), k] = sum_w Bw[(w+s-1) -> j] * dy[n, h, w, k]                                    jj = column node of any stage
//   pyr_dn_h   G_stage[n, i, j, (r, s), k] = sum_h Bh[(h+r-1) -> i] * U[n, h, (jj, s), k]
//   filter_slice / filter_unslice: channel slices of the KRSC filter as contiguous GEMM operands (rows (k, rs) or (rs, k)).
#include "bilinear.h"
#include "rowgeom.h"

namespace {

constexpr int PB_MAX = 4;
struct PbGeom {
    int N, H, W, K, nl, nrow, ncol;              // nrow = sum of bh (row nodes of all stages), ncol = sum of bw
    int bh[PB_MAX], bw[PB_MAX], offh[PB_MAX], offw[PB_MAX];
};
struct PbPtrs { const float* t[PB_MAX]; float* g[PB_MAX]; };

__device__ __forceinline__ int pb_stage(const PbGeom& g, const int (&off)[PB_MAX], int ii) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < PB_MAX; ++q) s += (q < g.nl && ii >= off[q]) ? 1 : 0;
    return s;
}
__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) { a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w; }

// rows: (n, ii, r, w)
__global__ __launch_bounds__(256) void pyr_up_w_kernel(PbGeom g, PbPtrs p, float* __restrict__ V) {
    const int k4n = g.K >> 2;
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 >= k4n) return;
    const long rows = (long)g.N * g.nrow * 3 * g.W;
    const int ldt = 9 * g.K;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int w = (int)(row % g.W);
        long q = row / g.W;
        const int r = (int)(q % 3); q /= 3;
        const int ii = (int)(q % g.nrow), n = (int)(q / g.nrow);
        const int s = pb_stage(g, g.offh, ii), b = g.bw[s], i = ii - g.offh[s];
        const float sc = bl_scale(b, g.W, 1);
        const float* base = p.t[s] + ((long)(n * g.bh[s] + i) * b) * ldt + k4 * 4;
        float4 acc = zero4();
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int wp = w + sx - 1;
            if (wp < 0 || wp >= g.W) continue;                     // zero padding of the convolution
            const Lerp L = bl_src(wp, sc, b, 1);
            const float* tp = base + (r * 3 + sx) * g.K;
            fma4(acc, L.l0, ld4(tp + (long)L.i0 * ldt));
            if (L.l1 != 0.f) fma4(acc, L.l1, ld4(tp + (long)L.i1 * ldt));
        }
        st4(V + row * g.K + k4 * 4, acc);
    }
}

// rows: (n, h, w)
__global__ __launch_bounds__(256) void pyr_up_h_kernel(PbGeom g, const float* __restrict__ V, float* __restrict__ y, int ldy) {
    const int k4n = g.K >> 2;
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 >= k4n) return;
    const long rows = (long)g.N * g.H * g.W;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int w = (int)(row % g.W);
        const long q = row / g.W;
        const int h = (int)(q % g.H), n = (int)(q / g.H);
        float4 acc = zero4();
        for (int s = 0; s < g.nl; ++s) {
            const int b = g.bh[s];
            const float sc = bl_scale(b, g.H, 1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int hp = h + r - 1;
                if (hp < 0 || hp >= g.H) continue;
                const Lerp L = bl_src(hp, sc, b, 1);
                const float* vp = V + ((((long)n * g.nrow + g.offh[s]) * 3 + r) * g.W + w) * g.K + k4 * 4;
                fma4(acc, L.l0, ld4(vp + (long)L.i0 * 3 * g.W * g.K));
                if (L.l1 != 0.f) fma4(acc, L.l1, ld4(vp + (long)L.i1 * 3 * g.W * g.K));
            }
        }
        st4(y + row * ldy + k4 * 4, acc);
    }
}

// rows: (n, h, jj, s)   U[(n*H + h), (jj*3 + s), K]
__global__ __launch_bounds__(256) void pyr_dn_w_kernel(PbGeom g, const float* __restrict__ dy, int lddy, float* __restrict__ U) {
    const int k4n = g.K >> 2;
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 >= k4n) return;
    const long rows = (long)g.N * g.H * g.ncol * 3;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int sx = (int)(row % 3);
        long q = row / 3;
        const int jj = (int)(q % g.ncol);
        const long nh = q / g.ncol;                                   // n*H + h
        const int s = pb_stage(g, g.offw, jj), b = g.bw[s], j = jj - g.offw[s];
        const float sc = bl_scale(b, g.W, 1);
        int lo, hi;
        bl_range(j, sc, b, g.W, 1, lo, hi);                           // interpolated positions wp that can touch node j
        const float* base = dy + nh * g.W * lddy + k4 * 4;
        float4 acc = zero4();
        for (int wp = lo; wp <= hi; ++wp) {
            const int w = wp - (sx - 1);                              // output pixel whose tap sx lands on wp
            if (w < 0 || w >= g.W) continue;
            const Lerp L = bl_src(wp, sc, b, 1);
            const float wt = (L.i0 == j ? L.l0 : 0.f) + (L.i1 == j ? L.l1 : 0.f);
            if (wt != 0.f) fma4(acc, wt, ld4(base + (long)w * lddy));
        }
        st4(U + row * g.K + k4 * 4, acc);
    }
}

// rows: (n, node (ii, j) of any stage, r, s)
__global__ __launch_bounds__(256) void pyr_dn_h_kernel(PbGeom g, const float* __restrict__ U, PbPtrs p) {
    const int k4n = g.K >> 2;
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 >= k4n) return;
    int nodes = 0;
    for (int s = 0; s < g.nl; ++s) nodes += g.bh[s] * g.bw[s];
    const long rows = (long)g.N * nodes * 9;
    const int ldt = 9 * g.K;
    for (long row = (long)blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += (long)gridDim.y * blockDim.y) {
        const int rs = (int)(row % 9), r = rs / 3, sx = rs - 3 * r;
        long q = row / 9;
        int node = (int)(q % nodes);
        const int n = (int)(q / nodes);
        int s = 0;
        while (node >= g.bh[s] * g.bw[s]) { node -= g.bh[s] * g.bw[s]; ++s; }
        const int b = g.bh[s], bwid = g.bw[s], i = node / bwid, j = node - i * bwid;
        const float sc = bl_scale(b, g.H, 1);
        int lo, hi;
        bl_range(i, sc, b, g.H, 1, lo, hi);
        float4 acc = zero4();
        for (int hp = lo; hp <= hi; ++hp) {
            const int h = hp - (r - 1);
            if (h < 0 || h >= g.H) continue;
            const Lerp L = bl_src(hp, sc, b, 1);
            const float wt = (L.i0 == i ? L.l0 : 0.f) + (L.i1 == i ? L.l1 : 0.f);
            if (wt != 0.f) fma4(acc, wt, ld4(U + ((((long)n * g.H + h) * g.ncol + g.offw[s] + j) * 3 + sx) * g.K + k4 * 4));
        }
        st4(p.g[s] + ((long)(n * b + i) * bwid + j) * ldt + rs * g.K + k4 * 4, acc);
    }
}

// rs_major: out row = rs*K + k, else k*RS + rs; UNSLICE writes the slice back into the full filter gradient
template <bool UNSLICE>
__global__ __launch_bounds__(256) void filter_slice_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int RS, int Ctot, int c0,
                                                           int Cs, int rs_major) {
    const int c4n = Cs >> 2;
    const long total = (long)K * RS * c4n;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c4 = (int)(t % c4n);
        const long row = t / c4n;                                      // row of the slice
        const int k = rs_major ? (int)(row % K) : (int)(row / RS), rs = rs_major ? (int)(row / K) : (int)(row % RS);
        const long full = ((long)k * RS + rs) * Ctot + c0 + c4 * 4, part = row * Cs + c4 * 4;
        if (UNSLICE) st4(dst + full, ld4(src + part));
        else         st4(dst + part, ld4(src + full));
    }
}

bool pb_geom(int N, int H, int W, int K, int nl, const int* bins_h, const int* bins_w, PbGeom* g) {
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || (K & 3) || nl < 1 || nl > PB_MAX || !bins_h || !bins_w) return false;
    g->N = N; g->H = H; g->W = W; g->K = K; g->nl = nl; g->nrow = 0; g->ncol = 0;
    for (int s = 0; s < PB_MAX; ++s) { g->bh[s] = g->bw[s] = 1; g->offh[s] = g->offw[s] = 0; }
    for (int s = 0; s < nl; ++s) {
        if (bins_h[s] < 1 || bins_h[s] > 256 || bins_w[s] < 1 || bins_w[s] > 256) return false;
        g->bh[s] = bins_h[s]; g->bw[s] = bins_w[s];
        g->offh[s] = g->nrow; g->nrow += bins_h[s];
        g->offw[s] = g->ncol; g->ncol += bins_w[s];
    }
    return true;
}

}  // namespace

extern "C" {

int segmi_filter_slice(const float* w_krsc, int K, int RS, int Ctot, int c0, int Cs, int rs_major, float* out, segmi_stream_t stream) {
    if (!w_krsc || !out || K <= 0 || RS <= 0 || Ctot <= 0 || c0 < 0 || Cs <= 0 || c0 + Cs > Ctot) return SEGMI_ERR_BADARG;
    if ((Ctot & 3) || (c0 & 3) || (Cs & 3)) return SEGMI_ERR_ALIGN;
    const long total = (long)K * RS * (Cs >> 2);
    hipLaunchKernelGGL((filter_slice_kernel<false>), dim3((unsigned)min((total + 255) / 256, (long)SEGMI_MAX_GRID * 4)), dim3(256), 0, (hipStream_t)stream,
                       w_krsc, out, K, RS, Ctot, c0, Cs, rs_major ? 1 : 0);
    return segmi_launch_status();
}

int segmi_filter_unslice(const float* grad_slice, int K, int RS, int Ctot, int c0, int Cs, int rs_major, float* dw_krsc, segmi_stream_t stream) {
    if (!grad_slice || !dw_krsc || K <= 0 || RS <= 0 || Ctot <= 0 || c0 < 0 || Cs <= 0 || c0 + Cs > Ctot) return SEGMI_ERR_BADARG;
    if ((Ctot & 3) || (c0 & 3) || (Cs & 3)) return SEGMI_ERR_ALIGN;
    const long total = (long)K * RS * (Cs >> 2);
    hipLaunchKernelGGL((filter_slice_kernel<true>), dim3((unsigned)min((total + 255) / 256, (long)SEGMI_MAX_GRID * 4)), dim3(256), 0, (hipStream_t)stream,
                       grad_slice, dw_krsc, K, RS, Ctot, c0, Cs, rs_major ? 1 : 0);
    return segmi_launch_status();
}

size_t segmi_pyramid_up_workspace(int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w) {
    PbGeom g;
    if (!pb_geom(N, H, W, K, nlevels, bins_h, bins_w, &g)) return 0;
    const size_t v = (size_t)N * g.nrow * 3 * W, u = (size_t)N * H * g.ncol * 3;      // forward / backward intermediate rows
    return (v > u ? v : u) * K * sizeof(float);
}

int segmi_pyramid_up_fwd(const float* const* T, int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w, float* y,
                         int ldy, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    PbGeom g;
    if (!T || !y || !pb_geom(N, H, W, K, nlevels, bins_h, bins_w, &g)) return SEGMI_ERR_BADARG;
    if ((ldy & 3) || ldy < K) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_pyramid_up_workspace(N, H, W, K, nlevels, bins_h, bins_w) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    PbPtrs p;
    for (int s = 0; s < PB_MAX; ++s) { p.t[s] = s < nlevels ? T[s] : nullptr; p.g[s] = nullptr; if (s < nlevels && !T[s]) return SEGMI_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    float* V = (float*)workspace;
    RowGeom a = row_geom((long)N * g.nrow * 3 * W, K, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(pyr_up_w_kernel, a.grid, a.block, 0, st, g, p, V);
    RowGeom b = row_geom((long)N * H * W, K, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(pyr_up_h_kernel, b.grid, b.block, 0, st, g, (const float*)V, y, ldy);
    return segmi_launch_status();
}

int segmi_pyramid_up_bwd(const float* dy, int lddy, int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w,
                         float* const* G, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    PbGeom g;
    if (!dy || !G || !pb_geom(N, H, W, K, nlevels, bins_h, bins_w, &g)) return SEGMI_ERR_BADARG;
    if ((lddy & 3) || lddy < K) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_pyramid_up_workspace(N, H, W, K, nlevels, bins_h, bins_w) || ((uintptr_t)workspace & 15)) return SEGMI_ERR_WORKSPACE;
    PbPtrs p;
    int nodes = 0;
    for (int s = 0; s < PB_MAX; ++s) { p.t[s] = nullptr; p.g[s] = s < nlevels ? G[s] : nullptr; if (s < nlevels && !G[s]) return SEGMI_ERR_BADARG; }
    for (int s = 0; s < nlevels; ++s) nodes += bins_h[s] * bins_w[s];
    hipStream_t st = (hipStream_t)stream;
    float* U = (float*)workspace;
    RowGeom a = row_geom((long)N * H * g.ncol * 3, K, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(pyr_dn_w_kernel, a.grid, a.block, 0, st, g, dy, lddy, U);
    RowGeom b = row_geom((long)N * nodes * 9, K, 1, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(pyr_dn_h_kernel, b.grid, b.block, 0, st, g, (const float*)U, p);
    return segmi_launch_status();
}

}  // extern "C"
