// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions (padding == dilation, "same" size): forward, data gradient and filter
// gradient — the default algorithm of the eligible layers (segmi/ops.py: min(C, K) >= 256, sub-grids of >= 8 pixels).
// Replaces, for those layers, the same call sites as conv_igemm.hip (aten::conv2d / convolution_backward of
// models/resnet.py:84-86 conv2 of every Bottleneck, models/pspnet.py:27-30 bottleneck, models/deeplabv3_plus.py:264-284 ASPP,
// :307-318 decoder, models/unet.py:15-18) with 2.25x fewer multiplications in the SAME arithmetic:
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, d = its 4x4 input patch
//
// The transform constants are 0, +-1, +-1/2, so nothing is rounded that an fp32 addition would not round; only the order of
// the channel summation per output changes (16 transform-domain sums combined by A instead of 9 tap sums).  Measured on the
// CPU twin (tools/probes/winograd_numerics.py): per-layer error 0.9-2.8x the direct fp32 convolution's, PSPNet-R50 logits
// 1.2-1.7x (1.1-1.8e-4 of max|logit|), UNet parameter gradients as close to fp64 as the direct path.
//
// Dataflow (HBM layouts, fp32):
//   wino_filter_kernel   g [O,3,3,I]            -> U [16][O][I]          (per step: the filters change)
//   wino_input_kernel    x [N,H,W,ldx]          -> V [16][Tpad][Cin]     T = N * dil^2 * th * tw tiles, Tpad = round_up(T, 32) (zero rows);
//                                                  the forward pass may write V into a caller buffer kept for the filter gradient
//   segmi_internal_gemm_batched                   M_xi [T, Cout] = V_xi [T, Cin] x U_xi [Cout, Cin]^T, xi < 16, ONE launch of the
//                                                  LDS-DMA implicit-GEMM kernel (blockIdx.y = xi), under the process-wide conv arithmetic
//   wino_output_kernel   M [16][T][ldm]         -> y [N,H,W,ldy]  (+ bias) (+ y)
//   filter gradient:     wino_dy_kernel dy -> W [16][Tpad][Kp];  ONE batched launch of the filter-gradient kernel (blockIdx.z = xi)
//                        dU_xi = W_xi^T V_xi into split partials;  wino_filter_grad_kernel sums the splits and applies G^T . G
// A dilation d decomposes into d*d dense problems on the sub-grids (h % d, w % d): tile (n, i, j, ty, tx) covers the output
// pixels h = (2 ty + a) d + i, w = (2 tx + b) d + j, a, b < 2, and reads rows (2 ty - 1 + u) d + i, u < 4.
// The transforms are HBM streams (V is 4x the input, M 4x the output); the contraction is MFMA-bound and 2.25x shorter.
#include "conv_internal.h"
#include "rowgeom.h"
#include "welford.h"
#include <cstdio>

namespace {

struct WinoGeom {
    int N, H, W, d, th, tw;
    long T;
};

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

struct TileIdx { int n, i, j, ty, tx; };
__device__ __forceinline__ TileIdx tile_of(long t, const WinoGeom& g) {
    TileIdx q;
    q.tx = (int)(t % g.tw); t /= g.tw;
    q.ty = (int)(t % g.th); t /= g.th;
    q.j = (int)(t % g.d); t /= g.d;
    q.i = (int)(t % g.d);
    q.n = (int)(t / g.d);
    return q;
}

// U[xi][o][i] = (G g[o,:,:,i] G^T)[xi];  flip: the 180-degree rotated filter (data gradient: g = the [C,3,3,Kp] re-laid filter)
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ g, int O, int I, int flip, float* __restrict__ U) {
    const long n = (long)O * I;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long o = idx / I;
        const int i = (int)(idx - o * I);
        float w[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) w[r][s] = g[((o * 3 + (flip ? 2 - r : r)) * 3 + (flip ? 2 - s : s)) * I + i];
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = w[0][s];
            t[1][s] = 0.5f * (w[0][s] + w[1][s] + w[2][s]);
            t[2][s] = 0.5f * (w[0][s] - w[1][s] + w[2][s]);
            t[3][s] = w[2][s];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
            U[(a * 4 + 0) * n + idx] = u0;
            U[(a * 4 + 1) * n + idx] = u1;
            U[(a * 4 + 2) * n + idx] = u2;
            U[(a * 4 + 3) * n + idx] = u3;
        }
    }
}

// V[xi][t][c] = (B^T d B)[xi], d = the 4x4 input patch of tile t (zero outside the image)
// (planes hold Tpad >= T rows; rows T..Tpad-1 are written as zeros: the filter-gradient contraction walks whole 32-row chunks)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, int C, WinoGeom g, long Tpad, float* __restrict__ V) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= C) return;
    const long plane = Tpad * (long)C;
    for (long t = (long)blockIdx.y * blockDim.y + threadIdx.y; t < Tpad; t += (long)gridDim.y * blockDim.y) {
        if (t >= g.T) {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) st4(V + xi * plane + t * C + c4 * 4, zero4());
            continue;
        }
        const TileIdx q = tile_of(t, g);
        float4 r[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sx = 2 * q.tx - 1 + v, w = sx * g.d + q.j;
            const bool wok = sx >= 0 && w < g.W;
            float4 dcol[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sy = 2 * q.ty - 1 + u, h = sy * g.d + q.i;
                const bool ok = wok && sy >= 0 && h < g.H;
                dcol[u] = ok ? ld4(x + (((long)q.n * g.H + h) * g.W + w) * ldx + c4 * 4) : zero4();
            }
            r[0][v] = sub4(dcol[0], dcol[2]);
            r[1][v] = add4(dcol[1], dcol[2]);
            r[2][v] = sub4(dcol[2], dcol[1]);
            r[3][v] = sub4(dcol[1], dcol[3]);
        }
        float* o = V + t * C + c4 * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            st4(o + (u * 4 + 0) * plane, sub4(r[u][0], r[u][2]));
            st4(o + (u * 4 + 1) * plane, add4(r[u][1], r[u][2]));
            st4(o + (u * 4 + 2) * plane, sub4(r[u][2], r[u][1]));
            st4(o + (u * 4 + 3) * plane, sub4(r[u][1], r[u][3]));
        }
    }
}

// y[tile] = A^T m A (+ bias) (+ y), m = M[:, t, k]
// STATS: the BN-statistics partials of y while its values are in registers (the output transform is the last kernel that holds
// them: its 2x2 tile is the only place a Winograd layer's y exists before HBM) — per thread a Welford run over the outputs it
// writes, a Chan tree over the block's thread rows, ONE {n, mean, M2} partial per (blockIdx.y, channel) in the layout of the
// implicit-GEMM kernel's epilogue (conv_igemm.hip): stats[blockIdx.y][3][K].  K % 4 == 0, no accumulate (host side).
template <bool STATS>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mm, int ldm, int K, WinoGeom g,
                                                          const float* __restrict__ bias, float* __restrict__ y, int ldy, int accumulate,
                                                          float* __restrict__ stats) {
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool kok = k4 * 4 < ((K + 3) & ~3);
    if (!STATS && !kok) return;
    Wf4 wf;
    wf_init(wf);
    if (kok) {
    const long plane = g.T * (long)ldm;
    float4 bv = zero4();
    if (bias) {
        const int k = k4 * 4;
        bv.x = k < K ? bias[k] : 0.f; bv.y = k + 1 < K ? bias[k + 1] : 0.f;
        bv.z = k + 2 < K ? bias[k + 2] : 0.f; bv.w = k + 3 < K ? bias[k + 3] : 0.f;
    }
    for (long t = (long)blockIdx.y * blockDim.y + threadIdx.y; t < g.T; t += (long)gridDim.y * blockDim.y) {
        const TileIdx q = tile_of(t, g);
        const float* m = Mm + t * ldm + k4 * 4;
        float4 s[2][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 m0 = ld4(m + (0 * 4 + v) * plane), m1 = ld4(m + (1 * 4 + v) * plane);
            const float4 m2 = ld4(m + (2 * 4 + v) * plane), m3 = ld4(m + (3 * 4 + v) * plane);
            s[0][v] = add4(add4(m0, m1), m2);
            s[1][v] = sub4(sub4(m1, m2), m3);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = (2 * q.ty + a) * g.d + q.i;
            if (h >= g.H) continue;
            const float4 o0 = add4(add4(s[a][0], s[a][1]), s[a][2]);
            const float4 o1 = sub4(sub4(s[a][1], s[a][2]), s[a][3]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int w = (2 * q.tx + b) * g.d + q.j;
                if (w >= g.W) continue;
                float* dst = y + (((long)q.n * g.H + h) * g.W + w) * ldy + k4 * 4;
                float4 o = add4(b ? o1 : o0, bv);
                if (accumulate) o = add4(o, ld4(dst));
                st4(dst, o);
                if (STATS) wf_push(wf, o);
            }
        }
    }
    }
    if (STATS) {
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
        if (threadIdx.y == 0 && kok) {
            const Wf4 a = sm[t];
            float* o = stats + (long)blockIdx.y * 3 * K + k4 * 4;
            st4(o, make_float4(a.n, a.n, a.n, a.n));
            st4(o + K, a.mean);
            st4(o + 2 * K, a.m2);
        }
    }
}

// W[xi][t][k] = (A dy A^T)[xi], dy = the 2x2 output-gradient tile t (zero outside the image); rows T..Tpad-1 zero
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int lddy, int Kp, WinoGeom g, long Tpad, float* __restrict__ Wt) {
    const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 * 4 >= Kp) return;
    const long plane = Tpad * (long)Kp;
    for (long t = (long)blockIdx.y * blockDim.y + threadIdx.y; t < Tpad; t += (long)gridDim.y * blockDim.y) {
        float* o = Wt + t * Kp + k4 * 4;
        if (t >= g.T) {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) st4(o + xi * plane, zero4());
            continue;
        }
        const TileIdx q = tile_of(t, g);
        float4 e[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int h = (2 * q.ty + a) * g.d + q.i, w = (2 * q.tx + b) * g.d + q.j;
                e[a][b] = (h < g.H && w < g.W) ? ld4(dy + (((long)q.n * g.H + h) * g.W + w) * lddy + k4 * 4) : zero4();
            }
        // r = A e (4x2), A = [[1,0],[1,1],[1,-1],[0,-1]]
        float4 r[4][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            r[0][b] = e[0][b];
            r[1][b] = add4(e[0][b], e[1][b]);
            r[2][b] = sub4(e[0][b], e[1][b]);
            r[3][b] = sub4(zero4(), e[1][b]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            st4(o + (u * 4 + 0) * plane, r[u][0]);
            st4(o + (u * 4 + 1) * plane, add4(r[u][0], r[u][1]));
            st4(o + (u * 4 + 2) * plane, sub4(r[u][0], r[u][1]));
            st4(o + (u * 4 + 3) * plane, sub4(zero4(), r[u][1]));
        }
    }
}

// dg[k][r][s][c] = (G^T dU[:, k, c] G)[r][s],  dU = the sum of the nsplit partial slices ws[s][xi][k][c] in slice order
// (deterministic): the split-K reduction of the 16 batched contractions and the filter-gradient transform in ONE pass
__global__ __launch_bounds__(256) void wino_filter_grad_kernel(const float* __restrict__ ws, int nsplit, int K, int C, float* __restrict__ dg) {
    const long n = (long)K * C;
    const long slice = 16 * n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long k = idx / C;
        const int c = (int)(idx - k * C);
        float u[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) u[a][b] = ws[(a * 4 + b) * n + idx];
        for (int sp = 1; sp < nsplit; ++sp) {
            const float* w = ws + sp * slice + idx;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) u[a][b] += w[(a * 4 + b) * n];
        }
        float t[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
            t[1][b] = 0.5f * (u[1][b] - u[2][b]);
            t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float* o = dg + ((k * 3 + r) * 3) * C + c;
            o[0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[C] = 0.5f * (t[r][1] - t[r][2]);
            o[2 * (long)C] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

bool wino_geom(int N, int H, int W, int d, WinoGeom* g) {
    if (N <= 0 || H <= 0 || W <= 0 || d <= 0) return false;
    g->N = N; g->H = H; g->W = W; g->d = d;
    g->th = (segmi_cdiv(H, d) + 1) / 2;
    g->tw = (segmi_cdiv(W, d) + 1) / 2;
    g->T = (long)N * d * d * g->th * g->tw;
    return g->T > 0 && g->T < (1L << 31);
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// Measurement hook (segmi_conv2d_winograd_trace): a pair of caller-owned HIP events recorded around the CONTRACTION launch of the
// next Winograd call of this thread, so that a profiler bracketing the whole call can separate the MFMA-bound batched GEMM from
// the HBM-bound transforms around it.  One-shot: consumed (cleared) by the call that uses it.
thread_local hipEvent_t g_trace_begin = nullptr, g_trace_end = nullptr;
struct TraceScope {
    hipStream_t st; hipEvent_t e;
    explicit TraceScope(hipStream_t s) : st(s), e(g_trace_end) {
        if (g_trace_begin) hipEventRecord(g_trace_begin, st);
        g_trace_begin = g_trace_end = nullptr;
    }
    ~TraceScope() { if (e) hipEventRecord(e, st); }
};

struct WinoPlan { WinoGeom g; long Tpad; int Cin, Cout, ldm; size_t u_bytes, v_bytes, m_bytes; };

// op 0: y = conv(x, w); op 1: dx = conv^T(dy, w).  Cin/Cout are the contraction / output widths of the pass.
bool wino_plan(const segmi_conv_desc* d, int op, WinoPlan* pl) {
    if (!d || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != d->dil || d->P != d->H || d->Q != d->W) return false;
    if (d->N <= 0 || d->C <= 0 || d->K <= 0 || (d->C & 3)) return false;
    if (!wino_geom(d->N, d->H, d->W, d->dil, &pl->g)) return false;
    const int Kp = (d->K + 3) & ~3;
    pl->Cin = op == 0 ? d->C : Kp;
    pl->Cout = op == 0 ? d->K : d->C;
    pl->ldm = (pl->Cout + 3) & ~3;
    const long T = pl->g.T;
    pl->Tpad = (T + 31) & ~31L;       // V planes hold whole 32-row chunks (zero rows past T): the filter gradient can contract a kept V
    if (pl->Tpad * pl->Cin * 4 >= 0xFFFFFF00L || T * pl->ldm * 4 >= 0xFFFFFF00L || (long)pl->Cout * pl->Cin * 4 >= 0xFFFFFF00L) return false;
    pl->u_bytes = align256((size_t)16 * pl->Cout * pl->Cin * sizeof(float));
    pl->v_bytes = align256((size_t)16 * pl->Tpad * pl->Cin * sizeof(float));
    pl->m_bytes = align256((size_t)16 * T * pl->ldm * sizeof(float));
    return true;
}

// partials the output transform emits when asked for BN statistics: <= WINO_STATS_PARTS thread-row blocks walk the tiles
constexpr int WINO_STATS_PARTS = 256;
int wino_stats_parts(const WinoPlan& pl) {
    const RowGeom rg = row_geom(pl.g.T, pl.ldm, 1, WINO_STATS_PARTS);
    return (int)rg.grid.y;
}

int wino_run(const WinoPlan& pl, const float* src, int lds, const float* filt, int flip, const float* bias, float* dst, int ldd,
             int accumulate, float* v_keep, void* workspace, size_t workspace_bytes, hipStream_t st, float* stats = nullptr) {
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < pl.u_bytes + pl.v_bytes + pl.m_bytes) return SEGMI_ERR_WORKSPACE;
    if (v_keep && ((uintptr_t)v_keep & 15)) return SEGMI_ERR_ALIGN;
    float* U = (float*)workspace;
    float* V = v_keep ? v_keep : (float*)((char*)workspace + pl.u_bytes);      // caller keeps the transformed input for the filter gradient
    float* Mm = (float*)((char*)workspace + pl.u_bytes + pl.v_bytes);
    const long T = pl.g.T;
    {
        const long n = (long)pl.Cout * pl.Cin;
        long nb = (n + 255) / 256;
        if (nb > SEGMI_MAX_GRID * 4) nb = SEGMI_MAX_GRID * 4;
        hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)nb), dim3(256), 0, st, filt, pl.Cout, pl.Cin, flip, U);
    }
    {
        RowGeom rg = row_geom(pl.Tpad, pl.Cin, 1, SEGMI_MAX_GRID * 4);
        hipLaunchKernelGGL(wino_input_kernel, rg.grid, rg.block, 0, st, src, lds, pl.Cin, pl.g, pl.Tpad, V);
    }
    int rc;
    {
        TraceScope tr(st);
        rc = segmi_internal_gemm_batched(V, pl.Cin, U, Mm, pl.ldm, (int)T, pl.Cin, pl.Cout, 16, pl.Tpad * pl.Cin, (long)pl.Cout * pl.Cin, T * pl.ldm, st);
    }
    if (rc != SEGMI_OK) return rc;
    {
        if (stats) {
            RowGeom rg = row_geom(T, pl.ldm, 1, WINO_STATS_PARTS);
            hipLaunchKernelGGL(wino_output_kernel<true>, rg.grid, rg.block, 0, st, (const float*)Mm, pl.ldm, pl.Cout, pl.g, bias, dst, ldd, 0, stats);
        } else {
            RowGeom rg = row_geom(T, pl.ldm, 1, SEGMI_MAX_GRID * 4);
            hipLaunchKernelGGL(wino_output_kernel<false>, rg.grid, rg.block, 0, st, (const float*)Mm, pl.ldm, pl.Cout, pl.g, bias, dst, ldd, accumulate, (float*)nullptr);
        }
    }
    return segmi_launch_status();
}

}  // namespace

extern "C" {

int segmi_conv2d_winograd_ok(const segmi_conv_desc* d, int op) {
    WinoPlan pl;
    if (op != 0 && op != 1) return 0;
    if (!wino_plan(d, op, &pl)) return 0;
    return segmi_internal_gemm_ok(pl.g.T, pl.Cin, pl.Cin, pl.Cout) ? 1 : 0;
}

size_t segmi_conv2d_winograd_workspace(const segmi_conv_desc* d, int op) {
    WinoPlan pl;
    if ((op != 0 && op != 1) || !wino_plan(d, op, &pl)) return 0;
    return pl.u_bytes + pl.v_bytes + pl.m_bytes;
}

size_t segmi_conv2d_winograd_v_bytes(const segmi_conv_desc* d) {
    WinoPlan pl;
    return wino_plan(d, 0, &pl) ? (size_t)16 * pl.Tpad * pl.Cin * sizeof(float) : 0;
}

int segmi_conv2d_winograd_fwd_stats_parts(const segmi_conv_desc* d) {
    WinoPlan pl;
    if (!wino_plan(d, 0, &pl) || (d->K & 3)) return 0;
    return wino_stats_parts(pl);
}

int segmi_conv2d_winograd_fwd(const segmi_conv_desc* d, const float* x, const float* w_krsc, const float* bias, float* y,
                              int accumulate, float* v_keep, float* stats_partials, void* workspace, size_t workspace_bytes,
                              segmi_stream_t stream) {
    WinoPlan pl;
    if (!x || !w_krsc || !y || !wino_plan(d, 0, &pl)) return SEGMI_ERR_BADARG;
    if (stats_partials && (accumulate || (d->K & 3) || ((uintptr_t)stats_partials & 15))) return SEGMI_ERR_BADARG;
    if ((d->ldx & 3) || d->ldx < d->C || (d->ldy & 3) || d->ldy < pl.ldm || ((uintptr_t)x & 15) || ((uintptr_t)w_krsc & 15) ||
        ((uintptr_t)y & 15))
        return SEGMI_ERR_ALIGN;
    return wino_run(pl, x, d->ldx, w_krsc, 0, bias, y, d->ldy, accumulate, v_keep, workspace, workspace_bytes, (hipStream_t)stream, stats_partials);
}

int segmi_conv2d_winograd_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                                void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    WinoPlan pl;
    if (!dy || !w_crsk || !dx || !wino_plan(d, 1, &pl)) return SEGMI_ERR_BADARG;
    if ((d->ldy & 3) || d->ldy < pl.Cin || (d->ldx & 3) || d->ldx < pl.ldm || ((uintptr_t)dy & 15) || ((uintptr_t)w_crsk & 15) ||
        ((uintptr_t)dx & 15))
        return SEGMI_ERR_ALIGN;
    return wino_run(pl, dy, d->ldy, w_crsk, 1, nullptr, dx, d->ldx, accumulate, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- filter gradient: dg = G^T [ sum_tiles (A dy A^T) (.) (B^T d B) ] G; the 16 contractions over the tiles are 1x1 filter
// gradients (x := V_xi [T, C], dy := W_xi [T, Kp]) and run as ONE batched launch of the direct filter-gradient kernel
// (blockIdx.z = xi, one deterministic pixel split planned for 16x the tiles); the split reduction is folded into the G^T . G pass
struct WinoWgradPlan { WinoGeom g; long Tpad; int C, K, Kp, nsplit; size_t v_bytes, w_bytes, s_bytes; };

static bool wino_wgrad_plan(const segmi_conv_desc* d, WinoWgradPlan* pl) {
    WinoPlan f;
    if (!wino_plan(d, 0, &f)) return false;
    pl->g = f.g;
    pl->Tpad = (f.g.T + 31) & ~31L;
    pl->C = d->C; pl->K = d->K; pl->Kp = (d->K + 3) & ~3;
    if (pl->Tpad >= (1L << 31) || pl->Tpad * pl->C * 4 >= 0xFFFFFF00L || pl->Tpad * pl->Kp * 4 >= 0xFFFFFF00L) return false;
    pl->nsplit = segmi_internal_wgrad_batched_splits((int)pl->Tpad, pl->C, pl->K, 16);
    if (pl->nsplit < 1) return false;
    pl->v_bytes = align256((size_t)16 * pl->Tpad * pl->C * sizeof(float));
    pl->w_bytes = align256((size_t)16 * pl->Tpad * pl->Kp * sizeof(float));
    pl->s_bytes = align256((size_t)pl->nsplit * 16 * pl->K * pl->C * sizeof(float));
    return true;
}

int segmi_conv2d_winograd_wgrad_ok(const segmi_conv_desc* d) {
    WinoWgradPlan pl;
    return wino_wgrad_plan(d, &pl) ? 1 : 0;
}

size_t segmi_conv2d_winograd_wgrad_workspace(const segmi_conv_desc* d) {
    WinoWgradPlan pl;
    if (!wino_wgrad_plan(d, &pl)) return 0;
    return pl.v_bytes + pl.w_bytes + pl.s_bytes;
}

int segmi_conv2d_winograd_wgrad(const segmi_conv_desc* d, const float* x, const float* v_kept, const float* dy, float* dw_krsc,
                                void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    WinoWgradPlan pl;
    if ((!x && !v_kept) || !dy || !dw_krsc || !wino_wgrad_plan(d, &pl)) return SEGMI_ERR_BADARG;
    if ((d->ldx & 3) || d->ldx < d->C || (d->ldy & 3) || d->ldy < pl.Kp || ((uintptr_t)x & 15) || ((uintptr_t)v_kept & 15) ||
        ((uintptr_t)dy & 15) || ((uintptr_t)dw_krsc & 15))
        return SEGMI_ERR_ALIGN;
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < pl.v_bytes + pl.w_bytes + pl.s_bytes)
        return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const float* V = v_kept ? v_kept : (const float*)workspace;    // the forward pass's transformed input (segmi_conv2d_winograd_fwd v_keep)
    float* Wt = (float*)((char*)workspace + pl.v_bytes);
    float* sk = (float*)((char*)workspace + pl.v_bytes + pl.w_bytes);
    if (!v_kept) {
        RowGeom rg = row_geom(pl.Tpad, pl.C, 1, SEGMI_MAX_GRID * 4);
        hipLaunchKernelGGL(wino_input_kernel, rg.grid, rg.block, 0, st, x, d->ldx, pl.C, pl.g, pl.Tpad, (float*)workspace);
    }
    {
        RowGeom rg = row_geom(pl.Tpad, pl.Kp, 1, SEGMI_MAX_GRID * 4);
        hipLaunchKernelGGL(wino_dy_kernel, rg.grid, rg.block, 0, st, dy, d->ldy, pl.Kp, pl.g, pl.Tpad, Wt);
    }
    int rc;
    {
        TraceScope tr(st);
        rc = segmi_internal_wgrad_batched(V, Wt, sk, (int)pl.Tpad, pl.C, pl.K, 16, st);
    }
    if (rc != SEGMI_OK) return rc;
    {
        const long n = (long)pl.K * pl.C;
        long nb = (n + 255) / 256;
        if (nb > SEGMI_MAX_GRID * 4) nb = SEGMI_MAX_GRID * 4;
        hipLaunchKernelGGL(wino_filter_grad_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)sk, pl.nsplit, pl.K, pl.C, dw_krsc);
    }
    return segmi_launch_status();
}

int segmi_conv2d_winograd_wgrad_variant(const segmi_conv_desc* d, char* buf, size_t len) {
    WinoWgradPlan pl;
    if (!buf || len < 112 || !wino_wgrad_plan(d, &pl)) return SEGMI_ERR_BADARG;
    char k[80];
    if (segmi_internal_wgrad_batched_variant((int)pl.Tpad, pl.C, pl.K, 16, k, sizeof k) != SEGMI_OK) return SEGMI_ERR_BADARG;
    snprintf(buf, len, "winograd_f2x2_3x3 wgrad: 16 x %s", k);
    return SEGMI_OK;
}

int segmi_conv2d_winograd_trace(void* ev_begin, void* ev_end) {
    g_trace_begin = (hipEvent_t)ev_begin;
    g_trace_end = (hipEvent_t)ev_end;
    return SEGMI_OK;
}

long segmi_conv2d_winograd_tiles(const segmi_conv_desc* d) {
    WinoPlan pl;
    return wino_plan(d, 0, &pl) ? pl.g.T : 0;
}

int segmi_conv2d_winograd_variant(const segmi_conv_desc* d, int op, char* buf, size_t len) {
    WinoPlan pl;
    if (!buf || len < 96 || (op != 0 && op != 1) || !wino_plan(d, op, &pl)) return SEGMI_ERR_BADARG;
    char gemm[64];
    if (segmi_internal_gemm_variant((int)pl.g.T, pl.Cin, pl.Cout, gemm, sizeof gemm) != SEGMI_OK) return SEGMI_ERR_BADARG;
    snprintf(buf, len, "winograd_f2x2_3x3 %s: 16 x %s", op == 0 ? "fwd" : "dgrad", gemm);
    return SEGMI_OK;
}

}  // extern "C"
