// Dense conv2d forward / data-gradient / weight-gradient as implicit GEMM on the fp32 matrix
// cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain).
//
// Replaces aten::conv2d + aten::convolution_backward for every nn.Conv2d of the reference
// (models/resnet.py:80-87,137-145,184-185; models/pspnet.py:18,27,61,65,69;
//  models/deeplabv3_plus.py:21,80,94,143,146,256,275,279,306,312-319; models/unet.py:15,18,77).
//
// Data layout: activations NHWC (pixel stride ld), filters KRSC.  With that layout both GEMM
// operands of fprop/dgrad are contiguous along the reduction (channel) axis, so a tile row is one
// 64/128-byte run of a pixel; padding and dilation are pure index math on the row's base pixel.
//
//   fprop : Y[m, k]   = sum_{r,s,c} X[pix(m) * stride - pad + (r,s)*dil, c] * W[k,r,s,c]      (gather kernel)
//   dgrad : dX[m, c]  = sum_{r,s,k} dY[(pix(m) + pad - (r,s)*dil)/stride, k] * Wt[c,r,s,k]     (gather kernel)
//   wgrad : dW[k,r,s,c] = sum_m dY[m,k] * X[pix(m)*stride - pad + (r,s)*dil, c]               (wgrad kernel)
//
// Tiling: 256 threads = 4 waves (one per SIMD); a wave owns TM x TN MFMA tiles of 32x32.
// Operand tiles are staged global -> VGPR -> LDS (register staging so out-of-image taps can be
// zero-filled), double-buffered, one barrier per K-chunk.  LDS rows are padded by 4 floats so the
// ds_read_b128 fragment reads are bank-conflict free (MI355X_MICROARCH.md, LDS table).
#include "segmi_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

enum { MODE_FPROP = 0, MODE_DGRAD = 1 };

struct GatherParams {
    const float* src;  // [N, Hs, Ws, Cs] pixel stride lds
    const float* wgt;  // [Cd, R, S, Cs]
    const float* bias; // [Cd] or null
    float* dst;        // [N, Hd, Wd, Cd] pixel stride ldd
    int N, Hs, Ws, Cs, lds;
    int Hd, Wd, Cd, ldd;
    int R, S, stride, pad, dil;
    int M;  // N*Hd*Wd
    int accumulate;
    int tiles_m, tiles_n;
};

template <int BM, int BN, int BK, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_gather_kernel(GatherParams p) {
    constexpr int LDT = BK + 4;                      // padded LDS row (floats)
    constexpr int KQ = BK / 4;                       // float4 per tile row
    constexpr int RPP = 256 / KQ;                    // tile rows loaded per pass
    constexpr int A_IT = (BM + RPP - 1) / RPP;
    constexpr int B_IT = (BN + RPP - 1) / RPP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM][LDT]
    float* Bs = smem + 2 * BM * LDT;       // [2][BN][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    // tile id: n-tile fastest so concurrently resident workgroups of an XCD share the A rows in L2
    const unsigned ntiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned t = xcd_swizzle(blockIdx.x, ntiles);
    const int tn = t % p.tiles_n, tm = t / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kq = tid % KQ, lrow = tid / KQ;

    // ---- per-thread A row state (fixed for the whole K loop)
    int a_img[A_IT], a_bh[A_IT], a_bw[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int row = lrow + i * RPP;
        int m = m0 + row;
        bool ok = (row < BM) && (m < p.M);
        int mm = ok ? m : 0;
        int hw = p.Hd * p.Wd;
        int n = mm / hw, rem = mm - n * hw;
        int hd = rem / p.Wd, wd = rem - hd * p.Wd;
        a_img[i] = n * p.Hs * p.Ws;
        if (MODE == MODE_FPROP) { a_bh[i] = hd * p.stride - p.pad; a_bw[i] = wd * p.stride - p.pad; }
        else                    { a_bh[i] = hd + p.pad;            a_bw[i] = wd + p.pad; }
        a_ok[i] = ok;
    }
    bool b_ok[B_IT];
    long b_off[B_IT];
    const int RS = p.R * p.S;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int row = lrow + i * RPP;
        int k = n0 + row;
        b_ok[i] = (row < BN) && (k < p.Cd);
        b_off[i] = (long)(b_ok[i] ? k : 0) * RS * p.Cs;
    }

    float4 ra[A_IT], rb[B_IT];

    auto gload = [&](int r, int s, int c0) {
        const int c = c0 + kq * 4;
        const bool cok = c < p.Cs;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hs, ws;
            bool ok = a_ok[i] && cok;
            if (MODE == MODE_FPROP) {
                hs = a_bh[i] + r * p.dil; ws = a_bw[i] + s * p.dil;
            } else {
                int th = a_bh[i] - r * p.dil, tw = a_bw[i] - s * p.dil;
                if (p.stride == 1) { hs = th; ws = tw; }
                else {
                    ok = ok && th >= 0 && tw >= 0 && (th % p.stride == 0) && (tw % p.stride == 0);
                    hs = th / p.stride; ws = tw / p.stride;
                }
            }
            ok = ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
            ra[i] = ok ? ld4(p.src + (long)(a_img[i] + hs * p.Ws + ws) * p.lds + c) : zero4();
        }
        const int tap = r * p.S + s;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            bool ok = b_ok[i] && cok;
            rb[i] = ok ? ld4(p.wgt + b_off[i] + (long)tap * p.Cs + c) : zero4();
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int row = lrow + i * RPP;
            if (A_IT * RPP == BM || row < BM) st4(As + (buf * BM + row) * LDT + kq * 4, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = lrow + i * RPP;
            if (B_IT * RPP == BN || row < BN) st4(Bs + (buf * BN + row) * LDT + kq * 4, rb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunk = (p.Cs + BK - 1) / BK;
    const int T = nchunk * RS;
    int r = 0, s = 0, c0 = 0;  // channel chunk outer, taps inner: a pixel row's 9 taps reuse L1/L2 lines
    auto advance = [&]() {
        if (++s == p.S) { s = 0; if (++r == p.R) { r = 0; c0 += BK; } }
    };

    gload(r, s, c0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    const int lrow32 = lane & 31, lhalf = lane >> 5;
    for (int it = 0; it < T; ++it) {
        const bool more = it + 1 < T;
        if (more) { advance(); gload(r, s, c0); }
        const float* Ab = As + buf * BM * LDT;
        const float* Bb = Bs + buf * BN * LDT;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ld4(Ab + (wm0 + i * 32 + lrow32) * LDT + kk * 8 + lhalf * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = ld4(Bb + (wn0 + j * 32 + lrow32) * LDT + kk * 8 + lhalf * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    // Channels Cd..round_up(Cd,4)-1 (the 16-byte padding of the pixel row) are written as zeros so that
    // consumers may read whole float4 groups.
    const int cd4 = min((p.Cd + 3) & ~3, p.ldd);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = n0 + wn0 + j * 32 + lrow32;
        const bool kreal = k < p.Cd, kok = k < cd4;
        const float bv = (kreal && p.bias) ? p.bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                if (kok && m < p.M) {
                    float* o = p.dst + (long)m * p.ldd + k;
                    float v = acc[i][j][e] + bv;
                    if (p.accumulate) v += *o;
                    *o = kreal ? v : 0.f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* x;   // [N,H,W,C] ldx
    const float* dy;  // [N,P,Q,K] ldy
    float* out;       // dw or workspace: [nsplit][K][R][S][C]
    int N, H, W, C, ldx;
    int P, Q, K, ldy;
    int R, S, stride, pad, dil;
    int M;            // N*P*Q
    int tiles_k, tiles_c;
    int chunks_per_split;  // in units of BKP pixels
    long split_stride;     // K*R*S*C
};

template <int BM, int BN, int BKP, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AQ = BM / 4, BQ = BN / 4;          // float4 per tile row
    constexpr int A_RPP = 256 / AQ, B_RPP = 256 / BQ;
    constexpr int A_IT = BKP / A_RPP, B_IT = BKP / B_RPP;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(A_IT >= 1 && B_IT >= 1, "tile passes");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BKP][BM]
    float* Bs = smem + 2 * BKP * BM;      // [2][BKP][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int lrow32 = lane & 31, lhalf = lane >> 5;

    // tile = (tap, k-tile, c-tile)
    int tidx = blockIdx.x;
    const int tc = tidx % p.tiles_c; tidx /= p.tiles_c;
    const int tk = tidx % p.tiles_k; tidx /= p.tiles_k;
    const int tap = tidx;
    const int r = tap / p.S, s = tap - r * p.S;
    const int k0 = tk * BM, c0 = tc * BN;
    const int split = blockIdx.y;
    const int mbeg = split * p.chunks_per_split * BKP;
    const int mend = min(p.M, mbeg + p.chunks_per_split * BKP);

    const int a_col = (tid % AQ) * 4, a_row = tid / AQ;
    const int b_col = (tid % BQ) * 4, b_row = tid / BQ;
    const bool a_cok = (k0 + a_col) < p.K;
    const bool b_cok = (c0 + b_col) < p.C;
    const int PQ = p.P * p.Q;

    float4 ra[A_IT], rb[B_IT];
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int m = mb + a_row + i * A_RPP;
            ra[i] = (a_cok && m < mend) ? ld4(p.dy + (long)m * p.ldy + k0 + a_col) : zero4();
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int m = mb + b_row + i * B_RPP;
            bool ok = b_cok && m < mend;
            int mm = ok ? m : 0;
            int n = mm / PQ, rem = mm - n * PQ;
            int pp = rem / p.Q, qq = rem - pp * p.Q;
            int hs = pp * p.stride - p.pad + r * p.dil, ws = qq * p.stride - p.pad + s * p.dil;
            ok = ok && (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
            rb[i] = ok ? ld4(p.x + ((long)(n * p.H + hs) * p.W + ws) * p.ldx + c0 + b_col) : zero4();
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st4(As + (buf * BKP + a_row + i * A_RPP) * BM + a_col, ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) st4(Bs + (buf * BKP + b_row + i * B_RPP) * BN + b_col, rb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int buf = 0;
    if (mbeg < mend) {
        gload(mbeg);
        sstore(0);
        __syncthreads();
        for (int mb = mbeg; mb < mend; mb += BKP) {
            const bool more = mb + BKP < mend;
            if (more) gload(mb + BKP);
            const float* Ab = As + buf * BKP * BM;
            const float* Bb = Bs + buf * BKP * BN;
#pragma unroll
            for (int kk = 0; kk < BKP / 2; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ab[(kk * 2 + lhalf) * BM + wm0 + i * 32 + lrow32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bb[(kk * 2 + lhalf) * BN + wn0 + j * 32 + lrow32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (more) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    float* out = p.out + (long)split * p.split_stride;
    const int RS = p.R * p.S;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + wn0 + j * 32 + lrow32;
        const bool cok = c < p.C;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                if (cok && k < p.K) out[((long)k * RS + tap) * p.C + c] = acc[i][j][e];
            }
        }
    }
}

__global__ void splitk_reduce_kernel(const float* ws, float* out, long n4, int nsplit, long stride4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    const float4* w = reinterpret_cast<const float4*>(ws);
    for (; i < n4; i += step) {
        float4 a = w[i];
        for (int s = 1; s < nsplit; ++s) {
            float4 b = w[i + s * stride4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

// w[K][R][S][C] -> wt[C][R][S][Kpad]  (k >= K zero-filled)
__global__ void krsc_to_crsk_kernel(const float* w, float* wt, int K, int RS, int C, int Kpad) {
    // 32x32 LDS transpose of the (k, c) plane for one tap
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
    for (int j = ty; j < 32; j += 8) {
        int k = k0 + j, c = c0 + tx;
        tile[j][tx] = (k < K && c < C) ? w[((long)k * RS + tap) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, k = k0 + tx;
        if (c < C && k < Kpad) wt[((long)c * RS + tap) * Kpad + k] = tile[tx][j];
    }
}

// column sums of a [rows, C] matrix (bias gradients): stage 1 partials, stage 2 finalize
__global__ void colsum_partial_kernel(const float* x, int ld, long rows, int C, float* part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;  // 4 row lanes
    float acc = 0.f;
    if (c < C)
        for (long r = (long)blockIdx.y * 4 + rl; r < rows; r += (long)gridDim.y * 4) acc += x[r * ld + c];
    __shared__ float sm[4][64];
    sm[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < C) part[(long)blockIdx.y * C + c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* part, int nparts, int C, float* out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int i = 0; i < nparts; ++i) acc += part[(long)i * C + c];
    out[c] = acc;
}

// ---------------------------------------------------------------------------------- host side
template <int BM, int BN, int BK, int WM, int WN, int MODE>
int launch_gather(GatherParams& p, hipStream_t st) {
    p.tiles_m = segmi_cdiv(p.M, BM);
    p.tiles_n = segmi_cdiv(p.Cd, BN);
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    auto kern = conv_gather_kernel<BM, BN, BK, WM, WN, MODE>;
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles_m * p.tiles_n), dim3(256), lds, st, p);
    return segmi_launch_status();
}

int g_bk = 0;  // 0 = auto; tuning hook (SEGMI_CONV_BK)
int conv_bk() {
    if (g_bk == 0) {
        const char* e = getenv("SEGMI_CONV_BK");
        g_bk = (e && atoi(e) == 16) ? 16 : 32;
    }
    return g_bk;
}

template <int MODE>
int dispatch_gather(GatherParams& p, hipStream_t st) {
    const bool bk32 = conv_bk() == 32 && p.Cs >= 32;
    if (p.Cd > 64) return bk32 ? launch_gather<128, 128, 32, 2, 2, MODE>(p, st) : launch_gather<128, 128, 16, 2, 2, MODE>(p, st);
    if (p.Cd > 32) return bk32 ? launch_gather<128, 64, 32, 2, 2, MODE>(p, st) : launch_gather<128, 64, 16, 2, 2, MODE>(p, st);
    return bk32 ? launch_gather<128, 32, 32, 4, 1, MODE>(p, st) : launch_gather<128, 32, 16, 4, 1, MODE>(p, st);
}

bool desc_ok(const segmi_conv_desc* d) {
    if (!d) return false;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0) return false;
    if (d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->P <= 0 || d->Q <= 0) return false;
    if (d->P != (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1) return false;
    if (d->Q != (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1) return false;
    if ((long)d->N * d->H * d->W >= (1L << 31) || (long)d->N * d->P * d->Q >= (1L << 31)) return false;
    return true;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct WgradPlan { int bm, bn, tiles_k, tiles_c, nsplit, chunks_per_split; };
constexpr int WG_BKP = 16;
WgradPlan plan_wgrad(const segmi_conv_desc* d) {
    WgradPlan pl;
    pl.bm = d->K > 64 ? 128 : 64;
    pl.bn = d->C > 64 ? 128 : 64;
    pl.tiles_k = segmi_cdiv(d->K, pl.bm);
    pl.tiles_c = segmi_cdiv(d->C, pl.bn);
    const long tiles = (long)pl.tiles_k * pl.tiles_c * d->R * d->S;
    const long M = (long)d->N * d->P * d->Q;
    const long chunks = (M + WG_BKP - 1) / WG_BKP;
    long want = (4L * SEGMI_NUM_CU + tiles - 1) / tiles;        // ~4 workgroups per CU
    long max_split = chunks / 16 > 0 ? chunks / 16 : 1;          // >= 256 pixels per split
    long ns = want < max_split ? want : max_split;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
    pl.chunks_per_split = (int)((chunks + ns - 1) / ns);
    pl.nsplit = (int)((chunks + pl.chunks_per_split - 1) / pl.chunks_per_split);
    return pl;
}

template <int BM, int BN>
int launch_wgrad(WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    const size_t lds = (size_t)2 * WG_BKP * (BM + BN) * sizeof(float);
    dim3 grid((unsigned)(pl.tiles_k * pl.tiles_c * p.R * p.S), (unsigned)pl.nsplit);
    hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WG_BKP, 2, 2>), grid, dim3(256), lds, st, p);
    return segmi_launch_status();
}

}  // namespace

extern "C" {

int segmi_conv2d_fwd(const segmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                     int accumulate, segmi_stream_t stream) {
    if (!desc_ok(d) || !x || !w || !y) return SEGMI_ERR_BADARG;
    if ((d->C & 3) || (d->ldx & 3) || d->ldx < d->C || d->ldy < d->K || !aligned16(x) || !aligned16(w)) return SEGMI_ERR_ALIGN;
    GatherParams p;
    p.src = x; p.wgt = w; p.bias = bias; p.dst = y;
    p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.Cs = d->C; p.lds = d->ldx;
    p.Hd = d->P; p.Wd = d->Q; p.Cd = d->K; p.ldd = d->ldy;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.M = d->N * d->P * d->Q; p.accumulate = accumulate;
    return dispatch_gather<MODE_FPROP>(p, (hipStream_t)stream);
}

int segmi_conv2d_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                       segmi_stream_t stream) {
    if (!desc_ok(d) || !dy || !w_crsk || !dx) return SEGMI_ERR_BADARG;
    // the reduction axis is K here: the caller pads it to a multiple of 4 (Kpad = round_up(K,4) <= ldy)
    const int Kpad = (d->K + 3) & ~3;
    if ((d->ldy & 3) || d->ldy < Kpad || d->ldx < d->C || !aligned16(dy) || !aligned16(w_crsk)) return SEGMI_ERR_ALIGN;
    GatherParams p;
    p.src = dy; p.wgt = w_crsk; p.bias = nullptr; p.dst = dx;
    p.N = d->N; p.Hs = d->P; p.Ws = d->Q; p.Cs = Kpad; p.lds = d->ldy;
    p.Hd = d->H; p.Wd = d->W; p.Cd = d->C; p.ldd = d->ldx;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.M = d->N * d->H * d->W; p.accumulate = accumulate;
    return dispatch_gather<MODE_DGRAD>(p, (hipStream_t)stream);
}

size_t segmi_conv2d_wgrad_workspace(const segmi_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    WgradPlan pl = plan_wgrad(d);
    if (pl.nsplit <= 1) return 0;
    return (size_t)pl.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
}

int segmi_conv2d_wgrad(const segmi_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                       size_t workspace_bytes, segmi_stream_t stream) {
    if (!desc_ok(d) || !x || !dy || !dw) return SEGMI_ERR_BADARG;
    if ((d->C & 3) || (d->ldx & 3) || (d->ldy & 3) || d->ldx < d->C || d->ldy < ((d->K + 3) & ~3) || !aligned16(x) ||
        !aligned16(dy) || !aligned16(dw))
        return SEGMI_ERR_ALIGN;
    WgradPlan pl = plan_wgrad(d);
    const size_t need = segmi_conv2d_wgrad_workspace(d);
    if (need && (!workspace || workspace_bytes < need || !aligned16(workspace))) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    WgradParams p;
    p.x = x; p.dy = dy; p.out = pl.nsplit > 1 ? (float*)workspace : dw;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.ldx = d->ldx;
    p.P = d->P; p.Q = d->Q; p.K = d->K; p.ldy = d->ldy;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.M = d->N * d->P * d->Q;
    p.tiles_k = pl.tiles_k; p.tiles_c = pl.tiles_c; p.chunks_per_split = pl.chunks_per_split;
    p.split_stride = (long)d->K * d->R * d->S * d->C;
    int rc;
    if (pl.bm == 128 && pl.bn == 128) rc = launch_wgrad<128, 128>(p, pl, st);
    else if (pl.bm == 128) rc = launch_wgrad<128, 64>(p, pl, st);
    else if (pl.bn == 128) rc = launch_wgrad<64, 128>(p, pl, st);
    else rc = launch_wgrad<64, 64>(p, pl, st);
    if (rc != SEGMI_OK) return rc;
    if (pl.nsplit > 1) {
        const long n4 = p.split_stride / 4;  // C % 4 == 0
        int grid = (int)((n4 + 255) / 256);
        if (grid > SEGMI_MAX_GRID) grid = SEGMI_MAX_GRID;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)workspace, dw, n4, pl.nsplit, n4);
        rc = segmi_launch_status();
    }
    return rc;
}

int segmi_conv2d_variant(const segmi_conv_desc* d, int op, char* buf, size_t len) {
    if (!desc_ok(d) || !buf || len == 0 || op < 0 || op > 2) return SEGMI_ERR_BADARG;
    if (op == 2) {
        WgradPlan pl = plan_wgrad(d);
        snprintf(buf, len, "conv_wgrad_kernel<%d, %d, %d, 2, 2> splitk=%d", pl.bm, pl.bn, WG_BKP, pl.nsplit);
        return SEGMI_OK;
    }
    const int Cs = op == 0 ? d->C : ((d->K + 3) & ~3), Cd = op == 0 ? d->K : d->C;
    const int bk = (conv_bk() == 32 && Cs >= 32) ? 32 : 16;
    const int bn = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
    snprintf(buf, len, "conv_gather_kernel<128, %d, %d, %s, %d>", bn, bk, bn == 32 ? "4, 1" : "2, 2", op);
    return SEGMI_OK;
}

int segmi_filter_krsc_to_crsk(const float* w, float* wt, int K, int R, int S, int C, int Kpad, segmi_stream_t stream) {
    if (!w || !wt || K <= 0 || R <= 0 || S <= 0 || C <= 0 || Kpad < K) return SEGMI_ERR_BADARG;
    dim3 grid(segmi_cdiv(C, 32), segmi_cdiv(Kpad, 32), R * S);
    hipLaunchKernelGGL(krsc_to_crsk_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, K, R * S, C, Kpad);
    return segmi_launch_status();
}

static int colsum_parts(long rows) {
    long p = (rows + 255) / 256;
    if (p < 1) p = 1;
    if (p > 512) p = 512;
    return (int)p;
}
size_t segmi_colsum_workspace(long rows, int C) { return (size_t)colsum_parts(rows) * C * sizeof(float); }

int segmi_colsum(const float* x, int ld, long rows, int C, float* out, void* workspace, size_t workspace_bytes,
                 segmi_stream_t stream) {
    if (!x || !out || rows <= 0 || C <= 0 || ld < C) return SEGMI_ERR_BADARG;
    if (!workspace || workspace_bytes < segmi_colsum_workspace(rows, C)) return SEGMI_ERR_WORKSPACE;
    const int parts = colsum_parts(rows);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(segmi_cdiv(C, 64), parts), dim3(256), 0, st, x, ld, rows, C, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(segmi_cdiv(C, 256)), dim3(256), 0, st, (const float*)workspace, parts, C, out);
    return segmi_launch_status();
}

}  // extern "C"
