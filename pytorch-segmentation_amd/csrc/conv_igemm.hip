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
#include <type_traits>
#include "segmi_common.h"
#include "conv_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    // split of the reduction (channel chunks x taps) over blockIdx.y for problems with too few output tiles to fill the chip
    // (LDS-DMA kernel only): partial tiles go to `ws` [ksplit][M][ldd] and are summed by splitk_reduce_kernel
    int ksplit, its_per_split;
    float* ws;
    // fprop with Cs == 4 (the RGB stems, channels padded 3 -> 4): the reduction axis is re-indexed as j = tap*4 + c so that a
    // 32-wide chunk holds 8 taps x 4 channels instead of one tap's 4 channels + 28 zeros (7x7 stem: 7 chunks instead of 49)
    int pack4;
    // dgrad with stride > 1, decomposed by output parity class (ph, pw): the rows of this launch are the pixels
    // (n, hc*os + ph, wc*os + pw); only the taps whose source index is integral for the class are visited (a 3x3 stride-2
    // dgrad evaluates 1/2/2/4 of its 9 taps per class instead of 9 zero-padded ones), with per-tap source offsets and filter
    // tap ids from the tables below.  R = 1, S = ntaps for the loop logic; wRS = the filter's real tap count.
    int sub, Hc, Wc, ph, pw, os, wRS;
    int tab_r[16], tab_s[16], tab_w[16];
    // batch > 1 (LDS-DMA kernel, ksplit == 1): blockIdx.y selects one of `batch` independent problems of identical shape whose
    // operands lie bs_src / bs_wgt / bs_dst floats apart (the 16 transform-domain GEMMs of a Winograd convolution)
    int batch;
    long bs_src, bs_wgt, bs_dst;
    // fprop feeding a BatchNorm (LDS-DMA kernel, ksplit == 1, batch == 1, no accumulate): every workgroup also writes the Welford
    // partial {count, mean, M2} of ITS output tile, per channel, to stats[tile_m][3][stats_cp] — the layout bn_stats_merge_kernel
    // folds (csrc/bn.hip) — so the statistics pass of the BN layer never reads the tensor again
    float* stats;
    int stats_cp;
    int dbg;    // ablation hook (SEGMI_CONV_DBG, tools/experiments): 1 no epilogue stores, 2 no operand traffic after the first chunk, 4 no epilogue, 8 every chunk re-reads the first one (cache-hot operands), 16 one workgroup per CU, 32 no two-level summation
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
// LDS-DMA variant of the gather kernel (the fast path): operand tiles go global -> LDS directly with
// `buffer_load_dwordx4 ... offen lds` (no VGPR staging, no ds_write pass, no per-load branches):
//   * a wave-instruction deposits 64 lanes x 16 B = 1 KiB lane-linearly, i.e. 8 tile rows of BK = 32
//     floats (128 B) each, so LDS rows are UNPADDED; bank conflicts of the ds_read_b128 fragment reads
//     are avoided by an XOR swizzle of the 16-byte k-group, applied on the SOURCE address of the DMA and
//     again on the read (slot = kgroup ^ ((row >> 1) & 7): 16 consecutive rows hit 16 distinct 4-bank groups);
//   * out-of-image taps, rows past M / Cd and channels past Cs are given a byte offset beyond the buffer
//     descriptor's range: the hardware then writes ZEROS to LDS (tools/probes/lds_dma_oob.hip verifies
//     both properties on gfx950).  Padding and dilation stay pure index math, with no divergence.
// Requires both operands to be smaller than 4 GiB (32-bit buffer offsets); larger tensors take the
// register-staged kernel above.
// The DMA is issued from inline asm on purpose: hipcc does not count asm memory operations, so it does
// not put an `s_waitcnt vmcnt(0)` between the DMA of stage t+1 and the ds_reads of stage t (with the
// builtin it does, serialising load latency and MFMA work).  We count them ourselves: one
// `s_waitcnt vmcnt(0)` + barrier at the end of each stage.  M0 (LDS destination base) is written in
// the same statement that consumes it.
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));   // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);                              // num_records (bytes)
    r.w = 0x00020000;                                                              // raw buffer, 32-bit data format
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_void*)p; }
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned voffset, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voffset), "s"(rsrc), "s"(lds_dst) : "memory");
}

// the same with a scalar byte offset on top of the lane's (soffset field of the instruction)
__device__ __forceinline__ void dma16s(const i32x4& rsrc, unsigned voffset, unsigned soffset, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_dst) : "memory");
}

// FAST (R*S <= 32 taps, fprop or unit-stride dgrad): the source pixel of tap (r,s) is affine in the tap, so each DMA row
// keeps ONE base offset plus a 32-bit tap-validity mask computed once per workgroup; the per-chunk address work drops to
// an add, a bit test and a select per load (the issue phase is what keeps a wave off the matrix pipe: 124 -> ~60 VALU per chunk).
// PW (FAST, one tap, no parity classes — the 1x1 layers and the batched Winograd contractions): a lane's offsets do
// not change from chunk to chunk, the chunk's channel offset rides in the instruction's scalar offset, and a piece of the K loop is
// the DMA instruction alone.  Measured with the address arithmetic in place but no DMA (tools/experiments/r06t.sh,
// profiles/r06_conv_loop_ablation.txt): the ~6 VALU instructions per piece, not the loads, were what the matrix pipe waited for.
template <int BM, int BN, int WM, int WN, int MODE, bool FAST, bool PW = false>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(GatherParams p, unsigned src_bytes, unsigned wgt_bytes, unsigned dst_bytes) {
    constexpr int BK = 32;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;      // wave-instructions per wave per tile (8 rows each, 4 waves)
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile shape");

    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int bidy = p.batch > 1 ? 0 : (int)blockIdx.y;  // split-K slice, unless blockIdx.y is the batch index
    const long bat = p.batch > 1 ? (long)blockIdx.y : 0;
    const float* const src_base = p.src + bat * p.bs_src;     // (locals: writing to the by-value parameter block would move it to scratch)
    const float* const wgt_base = p.wgt + bat * p.bs_wgt;
    float* const dst_base = p.dst + bat * p.bs_dst;
    constexpr int STAGE = (BM + BN) * BK;             // floats per pipeline stage: A = BM rows x 32 fp32, B = BN rows x 32 fp32
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    // Tile order: each XCD owns a contiguous range of tile ids; ids run over groups of GN = 8 n-tiles, inside a group m-major
    // with the n-tile fastest.  The 64 workgroups resident on an XCD are then an 8 x 8 block of tiles sharing 8 A panels and
    // 8 filter panels in that XCD's L2 (with all n-tiles of a wide output in flight the 75 MB CRSK filter of the PSP
    // bottleneck's dgrad was re-streamed from the Infinity Cache once per pair of m-tiles: 9.5 GB per launch).
    const unsigned ntiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned t = xcd_swizzle(blockIdx.x, ntiles);
    constexpr unsigned GN = 8;
    const unsigned per_group = GN * (unsigned)p.tiles_m;
    const unsigned grp = t / per_group, in_grp = t - grp * per_group;
    const unsigned gw = min(GN, (unsigned)p.tiles_n - grp * GN);          // width of this (possibly last, narrower) group
    const int tm = in_grp / gw, tn = grp * GN + in_grp % gw;
    const int m0 = tm * BM, n0 = tn * BN;

    const i32x4 src_rsrc = make_rsrc(src_base, src_bytes);
    const i32x4 wgt_rsrc = make_rsrc(wgt_base, wgt_bytes);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + (unsigned)wave * (8 * BK * 4);

    // DMA role of this lane: row (wave*8 + lane/8) of every 32-row group, 16-byte slot lane%8
    const int rl = wave * 8 + (lane >> 3);
    const int kg = (lane & 7) ^ ((rl >> 1) & 7);        // k-group this lane fetches (source-side swizzle)

    int a_pix[A_IT], a_bh[A_IT], a_bw[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + i * 32 + rl;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const bool sub = MODE == MODE_DGRAD && FAST && p.sub;
        const int Hr = sub ? p.Hc : p.Hd, Wr = sub ? p.Wc : p.Wd;   // row space of this launch
        const int hw = Hr * Wr;
        const int n = mm / hw, rem = mm - n * hw;
        const int hd = rem / Wr, wd = rem - hd * Wr;
        a_pix[i] = n * p.Hs * p.Ws;
        if (MODE == MODE_FPROP) { a_bh[i] = hd * p.stride - p.pad; a_bw[i] = wd * p.stride - p.pad; }
        else if (sub)           { a_bh[i] = hd;                    a_bw[i] = wd; }
        else                    { a_bh[i] = hd + p.pad;            a_bw[i] = wd + p.pad; }
        a_ok[i] = ok;
    }
    const int RS = p.R * p.S;
    const bool subm = MODE == MODE_DGRAD && FAST && p.sub;
    int a_base[A_IT];
    unsigned a_mask[A_IT];
    if (FAST) {
        const int sgn = MODE == MODE_FPROP ? 1 : -1;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            a_base[i] = (a_pix[i] + a_bh[i] * p.Ws + a_bw[i]) * p.lds;
            unsigned mk = 0;
            for (int t2 = 0; t2 < RS; ++t2) {
                const int r2 = t2 / p.S, s2 = t2 - r2 * p.S;
                const int hs = subm ? a_bh[i] + p.tab_r[t2] : a_bh[i] + sgn * r2 * p.dil;
                const int ws = subm ? a_bw[i] + p.tab_s[t2] : a_bw[i] + sgn * s2 * p.dil;
                if (a_ok[i] && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws) mk |= 1u << t2;
            }
            a_mask[i] = mk;
        }
    }
    unsigned b_off[B_IT];                                // element offset of the filter row, OOB if k >= Cd
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int k = n0 + i * 32 + rl;
        b_off[i] = k < p.Cd ? (unsigned)k * (unsigned)((subm ? p.wRS : RS) * p.Cs) : OOB;
    }

    static_assert(!PW || FAST, "the pointwise form is a special case of the tap-mask form");
    unsigned pw_a[A_IT], pw_b[B_IT];                     // PW: byte offsets of the lane's 16 bytes in chunk 0 (OOB: row past M / Cd)
    const bool pw_tail_ok = kg * 4 < (p.Cs & 31);        // PW, Cs % 32 != 0: does the lane's channel group exist in the last, partial chunk
    if (PW) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) pw_a[i] = (a_mask[i] & 1u) ? (unsigned)(a_base[i] + kg * 4) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) pw_b[i] = b_off[i] != OOB ? (b_off[i] + (unsigned)(kg * 4)) * 4u : OOB;
    }

    auto issue = [&](int r, int s, int c0, int buf) {
        const unsigned As = lds0 + (unsigned)buf * (STAGE * 4), Bs = As + BM * BK * 4;   // LDS byte addresses (wave's 8-row slice)
        const int c = c0 + kg * 4;
        const bool cok = c < p.Cs;
        if (FAST) {
            const int tap = r * p.S + s;
            const int tap_off = (subm ? (p.tab_r[tap] * p.Ws + p.tab_s[tap]) * p.lds
                                      : (MODE == MODE_FPROP ? 1 : -1) * (r * p.dil * p.Ws + s * p.dil) * p.lds) + c;   // wave-uniform + lane's channel
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = cok && ((a_mask[i] >> tap) & 1u);
                dma16(src_rsrc, ok ? (unsigned)(a_base[i] + tap_off) * 4u : OOB, As + i * (32 * BK * 4));
            }
            if (subm) {
                const unsigned tapc2 = (unsigned)(p.tab_w[tap] * p.Cs + c);
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    dma16(wgt_rsrc, (cok && b_off[i] != OOB) ? (b_off[i] + tapc2) * 4u : OOB, Bs + i * (32 * BK * 4));
                return;
            }
        } else if (MODE == MODE_FPROP && p.pack4) {
            const int tapl = (c0 >> 2) + kg;                 // this lane's tap: 8 taps x 4 channels per chunk
            const bool tok = tapl < RS;
            const int r2 = tapl / p.S, s2 = tapl - r2 * p.S;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int hs = a_bh[i] + r2 * p.dil, ws = a_bw[i] + s2 * p.dil;
                const bool ok = a_ok[i] && tok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
                dma16(src_rsrc, ok ? (unsigned)(a_pix[i] + hs * p.Ws + ws) * (unsigned)p.lds * 4u : OOB, As + i * (32 * BK * 4));
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                dma16(wgt_rsrc, (tok && b_off[i] != OOB) ? (b_off[i] + (unsigned)tapl * 4u) * 4u : OOB, Bs + i * (32 * BK * 4));
            return;
        } else
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hs, ws;
            bool ok = a_ok[i] && cok;
            if (MODE == MODE_FPROP) {
                hs = a_bh[i] + r * p.dil; ws = a_bw[i] + s * p.dil;
            } else {
                const int th = a_bh[i] - r * p.dil, tw = a_bw[i] - s * p.dil;
                if (p.stride == 1) { hs = th; ws = tw; }
                else {
                    ok = ok && th >= 0 && tw >= 0 && (th % p.stride == 0) && (tw % p.stride == 0);
                    hs = th / p.stride; ws = tw / p.stride;
                }
            }
            ok = ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
            const unsigned off = ((unsigned)(a_pix[i] + hs * p.Ws + ws) * (unsigned)p.lds + (unsigned)c) * 4u;
            dma16(src_rsrc, ok ? off : OOB, As + i * (32 * BK * 4));
        }
        const unsigned tapc = (unsigned)((r * p.S + s) * p.Cs + c);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bool ok = cok && b_off[i] != OOB;
            dma16(wgt_rsrc, ok ? (b_off[i] + tapc) * 4u : OOB, Bs + i * (32 * BK * 4));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool pk = MODE == MODE_FPROP && !FAST && p.pack4;
    const int RSl = pk ? 1 : (RS > 0 ? RS : 1);          // taps folded into the chunk axis when packed (RS == 0: empty parity class)
    const int nchunk = ((pk ? RS * 4 : p.Cs) + BK - 1) / BK;
    // (round 6) Taps that are out of the image for EVERY row of this tile are skipped: a dilation-18 3x3 layer on a 33x33 map (DeepLab's
    // ASPP, the dilations of which are too wide for the Winograd sub-grids) reaches its upper taps from 15 of 33 image rows only, and a
    // 64-row tile covers two image rows — the zero chunks were a quarter of those launches.  tapmask = OR of the rows' tap-validity
    // masks over the workgroup; the iteration space is (channel chunk) x (set bits of tapmask).  Unsplit tap-mask launches only.
    unsigned tapmask = 0xFFFFFFFFu;
    const bool skiptaps = FAST && !PW && !subm && p.ksplit <= 1 && RS > 1;
    if (FAST && !PW) {
        __shared__ unsigned s_tapmask[4];
        if (skiptaps) {                                      // (workgroup-uniform)
            unsigned mk = 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) mk |= a_mask[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mk |= (unsigned)__shfl_xor((int)mk, off);
            if (lane == 0) s_tapmask[wave] = mk;
            __syncthreads();
            tapmask = __builtin_amdgcn_readfirstlane(s_tapmask[0] | s_tapmask[1] | s_tapmask[2] | s_tapmask[3]);
        }
    }
    const int nvalid = skiptaps ? __builtin_popcount(tapmask & (RS >= 32 ? 0xFFFFFFFFu : ((1u << RS) - 1u))) : RSl;
    const int Tall = nchunk * nvalid;
    const int it0 = bidy * p.its_per_split;              // split-K slice of the (chunk, tap) iteration space (whole range if ksplit == 1)
    const int T = min(Tall, it0 + p.its_per_split);
    const int Sl = p.S > 0 ? p.S : 1;
    int c0 = (it0 / RSl) * BK, r = (it0 % RSl) / Sl, s = (it0 % RSl) % Sl;  // channel chunk outer, taps inner: a pixel row's taps reuse L1/L2 lines
    int tcur = 0;                                        // skiptaps: current tap index (a set bit of tapmask)
    auto next_tap = [&](int from) {                      // first set bit of tapmask at or after `from`, RS if none
        const unsigned rest = from < 32 ? (tapmask >> from) << from : 0u;
        const int t = rest ? __builtin_ctz(rest) : 32;
        return t < RS ? t : RS;
    };
    if (skiptaps) { tcur = next_tap(0); if (tcur >= RS) tcur = 0; r = tcur / Sl; s = tcur - r * Sl; }
    auto advance = [&]() {
        if (pk) { c0 += BK; return; }
        if (skiptaps) {
            int t = next_tap(tcur + 1);
            if (t >= RS) { t = next_tap(0); c0 += BK; }
            tcur = t; r = t / Sl; s = t - r * Sl;
            return;
        }
        if (++s == p.S) { s = 0; if (++r == p.R) { r = 0; c0 += BK; } }
    };

    if (it0 < T) issue(r, s, c0, 0);                     // T == 0: a parity class without taps (its pixels are zeros)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lrow32 = lane & 31, lhalf = lane >> 5;
    const int swz = (lrow32 >> 1) & 7;                   // read-side swizzle (rows wm0 + i*32 + lrow32: same low bits)
    {
    // fp32 MFMA: the matrix instruction adds its 2 products onto the accumulator in k order, i.e. an fp32 chain as long as the
    // reduction (up to 9 x 2048 terms) with a rounding error that grows like sqrt(length) — measured 2-5x the error of the
    // reference's CPU kernels on the same operands (tools/probes/conv_error_vs_fp64.py, profiles/r04_conv_error_vs_fp64*.txt).
    // Two-level summation: every FLUSH chunks (FLUSH * 32 reduction elements) the running tile is added into a second
    // accumulator and restarted, so both chains stay short; reductions of <= FLUSH chunks (C <= 512 for a 1x1 layer) never
    // flush.  Cost: the matrix pipe drains once per flush (64 VALU adds + 64 moves per lane behind the MFMA -> VALU hazard).
    // Measured: flushing every 8 chunks brings the kernels' error from 2-5x to 1.2-1.5x torch-CPU's and the network-level gradient
    // distance from fp64 from 1.29x to 1.10x of the reference's own (cfg2); every 16 chunks only reaches 1.7-2.1x (and never
    // triggers on a C = 512 Winograd contraction); rotating ONE tile per chunk gives errors BELOW torch-CPU's but costs 16 % of the
    // kernel (MFMA -> VALU hazard stalls every chunk).
    constexpr int FLUSH = 8;
    f32x16 total[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) total[i][j][e] = 0.f;
    // (groups of FLUSH chunks: the inner loop is the plain pipelined K loop, untouched by the flush logic — a flush test inside it
    //  cost the ds_read / MFMA interleave 4 extra s_waitcnt per chunk and 3.7 % of the kernel)
    // (round 6) operand fragments double-buffered by hand, like the filter-gradient kernel's: the ds_reads of k-step kk+1 are in
    // flight while the MFMAs of step kk issue (the compiler's own schedule reused the A registers of the wave's first tile row for
    // the second and waited out the LDS latency behind every 8 MFMAs), and the second half of a chunk's last step is held back
    // behind the barrier, where it covers the ds_read latency of the next chunk's first fragments.
    // FAST: the DMA of the chunk after the next leaves in NP pieces of one wave-instruction each (~7 VALU + the load), one piece
    // behind every second MFMA of the chunk's first steps — in the 64-cycle shadow of a matrix instruction instead of ~300 cycles in
    // which the wave issues none.  A tile's last chunk issues the pieces with out-of-range offsets (zero fill, no memory traffic)
    // so that the chunk body stays one basic block.
    constexpr int NP = A_IT + B_IT;
    // Fragment reads: ds_read_b128 from inline asm with IMMEDIATE offsets (stage, tile row) on one of four lane addresses per
    // operand (one per k-step: the XOR swizzle is not additive), waits counted by hand — the compiler's own address arithmetic was
    // 3 VALU per k-step, and VALU between the MFMAs is what the matrix pipe waits for.  The stage is a compile-time argument of
    // the chunk body (two copies, even / odd chunk).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 fa[2][TM], fb[2][TN];
    unsigned fa_addr[4], fb_addr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned slotb = (unsigned)(((kk * 2 + lhalf) ^ swz) * 16);
        fa_addr[kk] = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + (unsigned)((wm0 + lrow32) * BK * 4) + slotb;
        fb_addr[kk] = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + (unsigned)((BM + wn0 + lrow32) * BK * 4) + slotb;
    }
    auto fetch = [&](int stg, int kk, int sb) {              // (stg, kk, sb: constants once the chunk body is inlined)
        const int SO = stg * STAGE * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[sb][i]) : "v"(fa_addr[kk]), "n"(SO + i * 32 * BK * 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[sb][j]) : "v"(fb_addr[kk]), "n"(SO + j * 32 * BK * 4));
    };
    auto landed = [&](int sb) {                              // behind an s_waitcnt: all but the newest TM + TN reads are in registers
        if (TM + TN == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (TM + TN == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[sb][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[sb][j]));
    };
    int q_tap = 0, q_tapoff = 0, q_c0 = 0;
    unsigned q_wtap = 0;
    bool q_have = false, q_tail = false;
    auto prep = [&](bool have) {                             // chunk-level scalars of the pieces (after advance())
        q_have = have && !(p.dbg & 2); q_c0 = (p.dbg & 8) ? 0 : c0; q_tap = r * p.S + s;
        q_tail = c0 + BK > p.Cs;
        if (subm) { q_tapoff = (p.tab_r[q_tap] * p.Ws + p.tab_s[q_tap]) * p.lds; q_wtap = (unsigned)(p.tab_w[q_tap] * p.Cs); }
        else      { q_tapoff = (MODE == MODE_FPROP ? 1 : -1) * (r * p.dil * p.Ws + s * p.dil) * p.lds; q_wtap = (unsigned)(q_tap * p.Cs); }
    };
    auto piece = [&](int idx, int dstbuf) {
        const unsigned As = lds0 + (unsigned)dstbuf * (STAGE * 4), Bs = As + BM * BK * 4;
        if (PW) {
            if (q_have) {                                      // (scalar branch around one instruction; no chunk behind a tile's last)
                const unsigned soff = (unsigned)q_c0 * 4u;
                if (!q_tail) {
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) if (i == idx) dma16s(src_rsrc, pw_a[i], soff, As + i * (32 * BK * 4));
#pragma unroll
                    for (int i = 0; i < B_IT; ++i) if (A_IT + i == idx) dma16s(wgt_rsrc, pw_b[i], soff, Bs + i * (32 * BK * 4));
                } else {                                       // the partial chunk of a channel count that is no multiple of 32
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) if (i == idx) dma16s(src_rsrc, pw_tail_ok ? pw_a[i] : OOB, soff, As + i * (32 * BK * 4));
#pragma unroll
                    for (int i = 0; i < B_IT; ++i) if (A_IT + i == idx) dma16s(wgt_rsrc, pw_tail_ok ? pw_b[i] : OOB, soff, Bs + i * (32 * BK * 4));
                }
            }
            return;
        }
        const int c = q_c0 + kg * 4;
        const unsigned live = ((unsigned)q_have & (unsigned)(c < p.Cs)) & 1u;      // (bitwise on purpose: `&&` became exec-masked branches)
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (i == idx) {
                const unsigned ok = (a_mask[i] >> q_tap) & live;
                dma16(src_rsrc, ok ? (unsigned)(a_base[i] + q_tapoff + c) * 4u : OOB, As + i * (32 * BK * 4));
            }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (A_IT + i == idx) {
                const unsigned ok = live & (unsigned)(b_off[i] != OOB);
                dma16(wgt_rsrc, ok ? (b_off[i] + q_wtap + (unsigned)c) * 4u : OOB, Bs + i * (32 * BK * 4));
            }
    };
    auto mfmas = [&](int sb, int q0, int q1, int& pc, int dstbuf) {   // reduction elements q0 .. q1-1 of fragment buffer sb
        int cnt = 0;
#pragma unroll
        for (int q = q0; q < q1; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i][q], fb[sb][j][q], acc[i][j], 0, 0, 0);
                    if (FAST && (++cnt & 1) == 0 && pc < NP) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(pc++, dstbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
    };
    int it = it0;
    bool flushed = false;                                    // (reductions of <= FLUSH chunks never flush: their tiles skip the final add)
    auto chunk = [&](const int B) __attribute__((always_inline)) {   // one K chunk out of stage B (a constant at both call sites)
        const int STG = B, OTH = B ^ 1;
        static_assert(BK / 8 == 4, "four k-steps per chunk");
        const bool have = it + 1 < T;
        if (have) advance();
        if (FAST) prep(have);
        else if (have) issue(r, s, c0, B ^ 1);
        int pc = 0;
        fetch(STG, 1, 1);
        landed(0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 4, pc, B ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(STG, 2, 0);
        landed(1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 4, pc, B ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(STG, 3, 1);
        landed(0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 4, pc, B ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the chunk's last fragments: every read of this stage is in registers
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[1][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[1][j]));
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 2, pc, B ^ 1);
        static_assert(!FAST || NP <= 7 * TM * TN, "every piece has a slot before the barrier");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage has landed in LDS
        __syncthreads();                                     // ... for every wave, and this stage is free again
        fetch(OTH, 0, 0);                                    // (after the tile's last chunk: a dead read of the other stage)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 2, 4, pc, B);                               // (pc == NP here: no piece behind the barrier)
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(0, 0, 0);
    while (it < T) {
    const int gend = min(T, it + FLUSH);                     // (FLUSH is even: every group starts on stage 0)
    while (it < gend) {
        chunk(0);
        if (++it >= gend) break;
        chunk(1);
        ++it;
    }
        if (it < T && !(p.dbg & 32)) {
            flushed = true;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { total[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.f; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the dead read behind the last chunk, before its registers are reused)
    if (flushed) {
        asm volatile("" ::: "memory");                       // (a real branch: if-converted, the add AND a select per element ran always)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += total[i][j][e];
    }
    }

    // ---- epilogue (same C/D mapping as the register-staged kernel)
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) dst_base[0] = 0.f; return; }
    const int cd4 = min((p.Cd + 3) & ~3, p.ldd);
    if (p.ksplit > 1) {                                  // partial tile -> workspace slice of this split (no bias / accumulate)
        float* out = p.ws + (long)bidy * p.M * p.ldd;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn0 + j * 32 + lrow32;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                    if (k < cd4 && m < p.M) out[(long)m * p.ldd + k] = k < p.Cd ? acc[i][j][e] : 0.f;
                }
        }
        return;
    }
    // Full tiles leave through LDS: the MFMA accumulator layout holds ONE channel per lane (column lane%32, 16 rows), which
    // would be 64 single-dword stores per lane — store-issue bound (MI355X_MICROARCH.md: ~7 B/clk/CU), a quarter of the
    // runtime of short-reduction (1x1, C = 256) tiles.  Each wave transposes its 64x32 column block in its own LDS slice
    // (the pipeline stages are free now) and emits 16-byte stores of 4 consecutive channels: 16 instead of 64 per lane.
    constexpr int EP = 36;                                // padded row pitch (floats) of the staging slice
    constexpr int NQ = TM * 4;                            // 8-row store steps of the wave's TM*32 rows
    float* stage = smem + wave * (TM * 32 * EP);          // TM*32 rows x 32 columns per pass
    const int er = lane >> 3, ec = (lane & 7) * 4;        // store role: row er (+8 per step), channels ec..ec+3
    // Byte offset of every store step's 16 bytes in column pass 0, once for all passes (OOB: row past M).  The tile leaves through
    // buffer stores on a descriptor of the destination (< 4 GiB, checked by the dispatcher): the pass's channel offset is the
    // instruction's scalar offset, rows past M are out-of-range offsets the hardware drops — no 64-bit address arithmetic and no
    // exec-masked branch per store (round 6: the epilogue's VALU instructions compete with the partner workgroup's MFMAs for issue).
    const i32x4 dst_rsrc = make_rsrc(dst_base, dst_bytes);
    unsigned voff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int m = m0 + wm0 + q * 8 + er;
        unsigned pix = (unsigned)m;
        if (subm) {                                       // parity-class launch: row -> (n, hc*os + ph, wc*os + pw)
            const int hw = p.Hc * p.Wc;
            const int n = m / hw, rem = m - n * hw;
            const int hc = rem / p.Wc, wc = rem - hc * p.Wc;
            pix = (unsigned)((n * p.Hd + (hc * p.os + p.ph)) * p.Wd + (wc * p.os + p.pw));
        }
        voff[q] = m < p.M ? (pix * (unsigned)p.ldd + (unsigned)ec) * 4u : OOB;
    }
    // accumulate (dgrad into the gradient the residual branch already wrote): ALL of the tile's previous values are requested
    // here, before the LDS transposition — issued one by one between the stores, as a load-wait-add-store chain per step, the
    // 16 HBM round trips of a tile were serialised and dominated short-reduction (1x1) layers
    float4 prev[TN][NQ];
    if (p.accumulate) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn0 + j * 32 + ec;
#pragma unroll
            for (int q = 0; q < NQ; ++q) prev[j][q] = (k < cd4 && voff[q] != OOB) ? ld4(dst_base + (size_t)(voff[q] >> 2) + (k - ec)) : zero4();
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf) * EP + lrow32] = acc[i][j][e];
        const int k = n0 + wn0 + j * 32 + ec;
        const bool kok = k < cd4;
        float4 bv = zero4();
        if (p.bias && kok) {
            bv.x = k < p.Cd ? p.bias[k] : 0.f; bv.y = k + 1 < p.Cd ? p.bias[k + 1] : 0.f;
            bv.z = k + 2 < p.Cd ? p.bias[k + 2] : 0.f; bv.w = k + 3 < p.Cd ? p.bias[k + 3] : 0.f;
        }
        float4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = ld4(stage + (q * 8 + er) * EP + ec);   // same wave wrote it: LDS is in order per wave
        if (p.bias) {                                        // (uniform real branches: BN-fed layers have no bias, forward passes no accumulate)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < NQ; ++q) { v[q].x += bv.x; v[q].y += bv.y; v[q].z += bv.z; v[q].w += bv.w; }
        }
        if (p.accumulate) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < NQ; ++q) { v[q].x += prev[j][q].x; v[q].y += prev[j][q].y; v[q].z += prev[j][q].z; v[q].w += prev[j][q].w; }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // pin the finished value HERE, in straight-line code: sunk into the predicated store blocks below, every block would
            // wait for its own operand with s_waitcnt vmcnt(0) — which on gfx9 also waits for the PREVIOUS STORE to be acknowledged
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
        }
        if (!(p.dbg & 1)) {
            const unsigned soff = (unsigned)(n0 + wn0 + j * 32) * 4u;                 // (wave-uniform)
            const bool ragged = n0 + wn0 + j * 32 + 32 > cd4;                         // (wave-uniform: only the last n-tile of a ragged K)
            typedef float f32x4s __attribute__((ext_vector_type(4)));
            if (!ragged) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4s d; d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(d), "v"(voff[q]), "s"(dst_rsrc), "s"(soff) : "memory");
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4s d; d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(d), "v"(kok ? voff[q] : OOB), "s"(dst_rsrc), "s"(soff) : "memory");
                }
            }
        }
        if (MODE == MODE_FPROP && p.stats) {
            // BN statistics of the tile while its values are in registers: a lane holds NQ rows x 4 channels — exact two-pass
            // {count, mean, M2} per lane, Chan merge across the 8 lanes that share the channel group (lane bits 3..5), then the
            // wave's 32-channel partial goes to LDS for the cross-wave merge below
            float cnt = 0.f;
            float4 sum = zero4();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (voff[q] != OOB) { cnt += 1.f; sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
            float4 mean = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv), m2 = zero4();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (voff[q] != OOB) {
                    const float dx = v[q].x - mean.x, dy = v[q].y - mean.y, dz = v[q].z - mean.z, dw = v[q].w - mean.w;
                    m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
                }
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                const float nb = __shfl_xor(cnt, off);
                const float4 mb = make_float4(__shfl_xor(mean.x, off), __shfl_xor(mean.y, off), __shfl_xor(mean.z, off), __shfl_xor(mean.w, off));
                const float4 qb = make_float4(__shfl_xor(m2.x, off), __shfl_xor(m2.y, off), __shfl_xor(m2.z, off), __shfl_xor(m2.w, off));
                const float nn = cnt + nb;
                const float f = nn > 0.f ? nb / nn : 0.f, g2 = cnt * f;
                float d;
                d = mb.x - mean.x; mean.x += d * f; m2.x += qb.x + d * d * g2;
                d = mb.y - mean.y; mean.y += d * f; m2.y += qb.y + d * d * g2;
                d = mb.z - mean.z; mean.z += d * f; m2.z += qb.z + d * d * g2;
                d = mb.w - mean.w; mean.w += d * f; m2.w += qb.w + d * d * g2;
                cnt = nn;
            }
            if (er == 0) {
                float* sst = smem + 4 * (TM * 32 * EP) + (wave / WN) * (3 * BN) + wn0 + j * 32 + ec;
                st4(sst, make_float4(cnt, cnt, cnt, cnt));
                st4(sst + BN, mean);
                st4(sst + 2 * BN, m2);
            }
        }
    }
    if (MODE == MODE_FPROP && p.stats) {             // (workgroup-uniform)
        static_assert(4 * (BM / WM) * EP + WM * 3 * BN <= 2 * (BM + BN) * BK, "statistics slice fits behind the staging slices");
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cd) {
            const float* sst = smem + 4 * (TM * 32 * EP) + tid;
            float n = sst[0], m = sst[BN], q2 = sst[2 * BN];
#pragma unroll
            for (int w = 1; w < WM; ++w) {            // fixed order: deterministic
                const float nb = sst[w * 3 * BN], mb = sst[w * 3 * BN + BN], qb = sst[w * 3 * BN + 2 * BN];
                const float nn = n + nb;
                if (nb > 0.f) {
                    const float d = mb - m, f = nb / nn;
                    m += d * f;
                    q2 += qb + d * d * n * f;
                    n = nn;
                }
            }
            float* o = p.stats + (long)tm * 3 * p.stats_cp + n0 + tid;
            o[0] = n; o[p.stats_cp] = m; o[2 * p.stats_cp] = q2;
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
    int pack4;             // C == 4: the N axis of the GEMM is j = tap*4 + c (all taps in one tile) instead of one tile per tap
    // batch > 1 (LDS-DMA kernel): blockIdx.z selects one of `batch` independent problems of identical shape whose operands /
    // results lie bs_x / bs_dy / bs_out floats apart (the 16 transform-domain contractions of a Winograd filter gradient)
    int batch;
    long bs_x, bs_dy, bs_out;
    int skiprows;  // 1 (generic form, Q >= 32): pixel chunks whose image rows are all outside the image for the workgroup's tap are skipped
    int flat;  // 1: (tile, split) from the linear workgroup id through xcd_swizzle, split-major (see conv_wgrad_dma_kernel); 0: blockIdx.y = split
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
    // Pixel coordinates of each B row this thread loads, advanced incrementally chunk by chunk (the
    // m -> (n, p, q) divisions are paid once here, not per chunk).
    int b_n[B_IT], b_p[B_IT], b_q[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int m = mbeg + b_row + i * B_RPP;
        const int mm = m < p.M ? m : 0;
        b_n[i] = mm / PQ;
        const int rem = mm - b_n[i] * PQ;
        b_p[i] = rem / p.Q;
        b_q[i] = rem - b_p[i] * p.Q;
    }
    const int tap_h = -p.pad + r * p.dil, tap_w = -p.pad + s * p.dil;
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int m = mb + a_row + i * A_RPP;
            ra[i] = (a_cok && m < mend) ? ld4(p.dy + (long)m * p.ldy + k0 + a_col) : zero4();
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int m = mb + b_row + i * B_RPP;
            const int hs = b_p[i] * p.stride + tap_h, ws = b_q[i] * p.stride + tap_w;
            const bool ok = b_cok && m < mend && (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
            rb[i] = ok ? ld4(p.x + ((long)(b_n[i] * p.H + hs) * p.W + ws) * p.ldx + c0 + b_col) : zero4();
            // advance this row's pixel by one chunk (BKP pixels)
            b_q[i] += BKP;
            while (b_q[i] >= p.Q) {
                b_q[i] -= p.Q;
                if (++b_p[i] == p.P) { b_p[i] = 0; ++b_n[i]; }
            }
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

// LDS-DMA variant of the weight-gradient kernel.  Tile rows are pixels (the GEMM reduction axis) and
// run along channels, so a row of BM (BN) floats is BM/4 lanes x 16 B of one DMA wave-instruction and the
// unpadded [BKP][BM] layout is already conflict-free for the per-lane ds_read_b32 operand reads
// (lanes read consecutive channels of one pixel).  Out-of-image taps / rows past the split's pixel range /
// channels past K (C) get an out-of-range offset and arrive as zeros.
// ROWQ (Q % 32 == 0): a 32-pixel chunk never straddles an output row, so (n, p, q0) of the chunk are wave-uniform scalars
// advanced with SALU, and a lane only adds its fixed in-chunk column: ~20 VALU per chunk instead of ~130 (the m -> (n,p,q)
// bookkeeping per lane and per load is what kept the generic path's waves off the matrix pipe: 125 vs 136 TF/s of fprop).
// PW (1x1, stride 1, no padding — the pointwise layers and the batched Winograd filter gradients): pixel m of dy IS pixel
// m of x, a lane's DMA offsets never change and the chunk's pixel offset is the instruction's scalar offset.
template <int BM, int BN, bool ROWQ, bool PW = false>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(WgradParams p, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int BKP = 32, WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LPR = BM / 4, B_LPR = BN / 4;               // lanes per tile row
    constexpr int A_RPI = 64 / A_LPR, B_RPI = 64 / B_LPR;       // tile rows per wave-instruction
    constexpr int A_IT = BKP / A_RPI / 4, B_IT = BKP / B_RPI / 4;  // wave-instructions per wave per stage
    constexpr unsigned OOB = 0xFFFFFFFFu;
    constexpr int STAGE = BKP * (BM + BN);
    static_assert(A_IT >= 1 && B_IT >= 1, "tile passes");

    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int lrow32 = lane & 31, lhalf = lane >> 5;

    // tile = (c-tile, k-tile, tap) with the tap fastest, and each XCD owning a contiguous range of tile ids:
    // an XCD's L2 then holds only ITS channel slices of x (read once per tap from L2, not from HBM) plus dy.
    // Workgroups are dispatched x-fastest and dealt round-robin to the 8 XCDs.  The (tile, split) pair is taken from the LINEAR
    // workgroup id through xcd_swizzle, split-major: an XCD then owns a contiguous range of pixel splits with ALL their tiles, so
    // the 64 workgroups resident on it walk the same pixel rows of x and dy together and every operand byte enters exactly one
    // L2 (with blockIdx.y = split and only the tiles swizzled, the 16 x 4 tiles of one split of a 512 -> 2048 1x1 layer sat on all
    // 8 XCDs: dy was fetched by 4 of them, x by 2 — 726 MB per launch for 335 MB of operands, profiles/r03_cfg2_conv_traffic_f32.json)
    const int RSn = p.pack4 ? 1 : p.R * p.S;
    const unsigned lin = p.flat ? xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y)
                                : xcd_swizzle(blockIdx.x, gridDim.x) + gridDim.x * blockIdx.y;
    int tidx = (int)(lin % gridDim.x);
    const int tap = tidx % RSn; tidx /= RSn;
    const int tk = tidx % p.tiles_k; tidx /= p.tiles_k;
    const int tc = tidx;
    const int k0 = tk * BM, c0 = tc * BN;
    const int split = (int)(lin / gridDim.x);
    const int mbeg = split * p.chunks_per_split * BKP;
    const int mend = min(p.M, mbeg + p.chunks_per_split * BKP);
    const int PQ = p.P * p.Q;

    const long bat = p.batch > 1 ? (long)blockIdx.z : 0;
    const i32x4 dy_rsrc = make_rsrc(p.dy + bat * p.bs_dy, dy_bytes);
    const i32x4 x_rsrc = make_rsrc(p.x + bat * p.bs_x, x_bytes);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));

    // DMA roles: A (dy) row = (i*4 + wave)*A_RPI + lane/A_LPR, channel group lane%A_LPR; same for B (x)
    const int a_rl = lane / A_LPR, a_col = (lane % A_LPR) * 4;
    const int b_rl = lane / B_LPR, b_col = (lane % B_LPR) * 4;
    const bool a_cok = (k0 + a_col) < ((p.K + 3) & ~3);          // dy rows own zero-filled channel padding up to 4
    // B columns: channels c0 + b_col of the tile's tap, or (pack4) tap (c0 + b_col)/4 with all 4 channels — per-lane tap then
    const int ltap = p.pack4 ? (c0 + b_col) >> 2 : tap;
    const int r = ltap / p.S, s = ltap - r * p.S;
    const bool b_cok = p.pack4 ? ltap < p.R * p.S : (c0 + b_col) < p.C;
    const int b_ch = p.pack4 ? 0 : c0 + b_col;                   // channel offset inside the source pixel
    int b_n[B_IT], b_p[B_IT], b_q[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int m = mbeg + (i * 4 + wave) * B_RPI + b_rl;
        const int mm = m < p.M ? m : 0;
        b_n[i] = mm / PQ;
        const int rem = mm - b_n[i] * PQ;
        b_p[i] = rem / p.Q;
        b_q[i] = rem - b_p[i] * p.Q;
    }
    const int tap_h = -p.pad + r * p.dil, tap_w = -p.pad + s * p.dil;

    // ROWQ state: chunk-uniform scalars + per-lane constants
    int cn = 0, cp = 0, cq = 0;
    unsigned a_const[A_IT];
    int b_ws[B_IT];
    if (ROWQ) {
        const int mm = mbeg < p.M ? mbeg : 0;
        cn = __builtin_amdgcn_readfirstlane(mm / PQ);
        const int rem = mm - cn * PQ;
        cp = __builtin_amdgcn_readfirstlane(rem / p.Q);
        cq = __builtin_amdgcn_readfirstlane(rem - cp * p.Q);
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            a_const[i] = a_cok ? ((unsigned)(((i * 4 + wave) * A_RPI + a_rl) * p.ldy + k0 + a_col)) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) b_ws[i] = ((i * 4 + wave) * B_RPI + b_rl) * p.stride + tap_w;
    }

    // One DMA wave-instruction of a stage (idx < A_IT: dy rows, else x rows).  PW: the lane's offset never changes and the chunk's
    // pixel offset rides in the instruction's scalar offset (no VALU at all); ROWQ: chunk-uniform scalars + a few VALU per x load;
    // generic: per-lane (n, p, q) bookkeeping.  The chunk-level scalars of ROWQ / PW are set by prep() before the first piece.
    unsigned pw_b[B_IT];
    if (PW) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            a_const[i] = a_cok ? ((unsigned)(((i * 4 + wave) * A_RPI + a_rl) * p.ldy + k0 + a_col)) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            pw_b[i] = b_cok ? ((unsigned)(((i * 4 + wave) * B_RPI + b_rl) * p.ldx + b_ch)) * 4u : OOB;
    }
    unsigned q_achunk = 0, q_bchunk = 0;
    int q_rowbase = 0, q_wq = 0, q_mb = 0;
    bool q_hok = false, q_tail = false;
    auto prep = [&](int mb, int jump = 0) {                      // jump: pixels skipped since the last issued chunk (generic form)
        q_mb = mb;
        q_tail = mb + BKP > mend;                                // (PW: the one chunk with rows past the last pixel)
        if (!PW && !ROWQ && jump) {                              // (scalar branch)
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                b_q[i] += jump;
                while (b_q[i] >= p.Q) {
                    b_q[i] -= p.Q;
                    if (++b_p[i] == p.P) { b_p[i] = 0; ++b_n[i]; }
                }
            }
        }
        if (PW) { q_achunk = (unsigned)mb * (unsigned)p.ldy * 4u; q_bchunk = (unsigned)mb * (unsigned)p.ldx * 4u; return; }
        if (ROWQ) {
            q_achunk = (unsigned)mb * (unsigned)p.ldy * 4u;                                     // scalar
            const int hs = cp * p.stride + tap_h;                                               // scalar
            q_hok = (unsigned)hs < (unsigned)p.H;
            q_rowbase = ((cn * p.H + hs) * p.W) * p.ldx + b_ch;
            q_wq = cq * p.stride;
            cq += BKP;
            if (cq >= p.Q) { cq = 0; if (++cp == p.P) { cp = 0; ++cn; } }
        }
    };
    auto piece = [&](int idx, int buf) {
        const unsigned As = lds0 + (unsigned)buf * (STAGE * 4), Bs = As + BKP * BM * 4;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (i == idx) {
                const unsigned dst = As + (i * 4 + wave) * A_RPI * (BM * 4);
                if (PW && q_tail) dma16s(dy_rsrc, q_mb + (i * 4 + wave) * A_RPI + a_rl < mend ? a_const[i] : OOB, q_achunk, dst);
                else if (PW || ROWQ) dma16s(dy_rsrc, a_const[i], q_achunk, dst);
                else {
                    const int m = q_mb + (i * 4 + wave) * A_RPI + a_rl;
                    const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)(k0 + a_col)) * 4u;
                    dma16(dy_rsrc, (a_cok && m < mend) ? off : OOB, dst);
                }
            }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (A_IT + i == idx) {
                const unsigned dst = Bs + (i * 4 + wave) * B_RPI * (BN * 4);
                if (PW && q_tail) dma16s(x_rsrc, q_mb + (i * 4 + wave) * B_RPI + b_rl < mend ? pw_b[i] : OOB, q_bchunk, dst);
                else if (PW) dma16s(x_rsrc, pw_b[i], q_bchunk, dst);
                else if (ROWQ) {
                    const int ws = q_wq + b_ws[i];
                    const unsigned ok = (unsigned)q_hok & (unsigned)b_cok & (unsigned)((unsigned)ws < (unsigned)p.W);
                    dma16(x_rsrc, ok ? (unsigned)(q_rowbase + ws * p.ldx) * 4u : OOB, dst);
                } else {
                    const int m = q_mb + (i * 4 + wave) * B_RPI + b_rl;
                    const int hs = b_p[i] * p.stride + tap_h, ws = b_q[i] * p.stride + tap_w;
                    const bool ok = b_cok && m < mend && (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
                    const unsigned off = ((unsigned)((b_n[i] * p.H + hs) * p.W + ws) * (unsigned)p.ldx + (unsigned)b_ch) * 4u;
                    dma16(x_rsrc, ok ? off : OOB, dst);
                    b_q[i] += BKP;
                    while (b_q[i] >= p.Q) {
                        b_q[i] -= p.Q;
                        if (++b_p[i] == p.P) { b_p[i] = 0; ++b_n[i]; }
                    }
                }
            }
    };
    constexpr int NP = A_IT + B_IT;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // (round 6) Generic form, dilated 3x3 on small maps (DeepLab's ASPP: dilation 18 on 33x33 reaches its upper taps from 15 of the
    // 33 image rows): a 32-pixel chunk covers at most two image rows when Q >= 32, and a chunk both rows of which are outside the
    // image for this workgroup's tap contributes zeros — skipped.  (sp, sq) = image row / column of the candidate chunk's first
    // pixel, carried in scalars; next_chunk() moves the candidate to the first chunk at or after it that has work.
    const bool skip = !PW && !ROWQ && p.skiprows;
    int sp = 0, sq = 0;
    if (skip) {
        const int mm = mbeg < p.M ? mbeg : 0;
        const int rem = mm - (mm / PQ) * PQ;
        sp = __builtin_amdgcn_readfirstlane(rem / p.Q);
        sq = __builtin_amdgcn_readfirstlane(rem - sp * p.Q);
    }
    auto next_chunk = [&](int mb) {
        if (!skip) return mb;
        while (mb < mend) {
            const int p1 = sp + 1 == p.P ? 0 : sp + 1;
            const bool v0 = (unsigned)(sp * p.stride + tap_h) < (unsigned)p.H;
            const bool v1 = sq + BKP > p.Q && (unsigned)(p1 * p.stride + tap_h) < (unsigned)p.H;
            if (v0 || v1) break;
            mb += BKP; sq += BKP;
            if (sq >= p.Q) { sq -= p.Q; sp = p1; }
        }
        return mb;
    };
    auto step_chunk = [&]() { sq += BKP; if (sq >= p.Q) { sq -= p.Q; if (++sp == p.P) sp = 0; } };
    const int mfirst = next_chunk(mbeg);
    if (mfirst < mend) {
        prep(mfirst, mfirst - mbeg);
#pragma unroll
        for (int q = 0; q < NP; ++q) piece(q, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        // (round 6) Operand reads are ds_read_b32 with IMMEDIATE offsets, issued from inline asm and counted by hand: the compiler
        // merged a k-step's two A (B) reads into ds_read2_b32, whose 8-bit offsets cannot reach past the first k-steps, and paid two
        // v_add per k-step for it — VALU instructions between the MFMAs are what the matrix pipe waits for (profiles/
        // r06_conv_loop_ablation.txt), LDS instructions are not.  Four fragment slots: k-step kk+2 is in flight while kk issues;
        // the last two k-steps of a chunk are held back behind the barrier, where they cover the first reads of the next stage;
        // the DMA of the chunk after the next leaves in NP pieces behind the first k-steps (no VALU in the PW form).
        constexpr int KS = BKP / 2;                                  // k-steps per chunk
        static_assert(KS % 4 == 0 && NP <= KS - 2, "slot rotation / piece slots");
        const unsigned a_lane = lds0 + (unsigned)((lhalf * BM + wm0 + lrow32) * 4);
        const unsigned b_lane = lds0 + (unsigned)(BKP * BM * 4 + (lhalf * BN + wn0 + lrow32) * 4);
        float fa[4][TM], fb[4][TN];
        auto fetch = [&](unsigned abase, unsigned bbase, int kk, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[slot][i]) : "v"(abase), "n"(kk * 2 * BM * 4 + i * 128));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[slot][j]) : "v"(bbase), "n"(kk * 2 * BN * 4 + j * 128));
        };
        auto landed = [&](int slot) {                                 // (after an s_waitcnt: ties the MFMAs to it)
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[slot][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[slot][j]));
        };
        auto mfmas = [&](int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
        };
        fetch(a_lane, b_lane, 0, 0);
        fetch(a_lane, b_lane, 1, 1);
        for (int mb = mfirst, mnext; mb < mend; mb = mnext) {
            if (skip) step_chunk();
            mnext = next_chunk(mb + BKP);
            const bool have = mnext < mend;
            if (have) prep(mnext, mnext - (mb + BKP));
            const unsigned abase = a_lane + (unsigned)buf * (STAGE * 4), bbase = b_lane + (unsigned)buf * (STAGE * 4);
#pragma unroll
            for (int kk = 0; kk < KS - 2; ++kk) {
                fetch(abase, bbase, kk + 2, (kk + 2) & 3);
                if ((TM + TN) * 2 == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if ((TM + TN) * 2 == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                landed(kk & 3);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kk & 3);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < NP && have) piece(kk, buf ^ 1);              // (scalar branch; a split's last chunk has nothing to fetch)
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every read of this stage is in registers
            landed((KS - 2) & 3); landed((KS - 1) & 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // next stage has landed
            __syncthreads();
            buf ^= 1;
            {
                const unsigned an = a_lane + (unsigned)buf * (STAGE * 4), bn = b_lane + (unsigned)buf * (STAGE * 4);
                fetch(an, bn, 0, 0);                                     // (after the split's last chunk: dead reads)
                fetch(an, bn, 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas((KS - 2) & 3);
            mfmas((KS - 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the dead reads, before the epilogue reuses the registers)
    }

    // epilogue through LDS like the fprop kernel: 16-byte stores of 4 consecutive input channels
    float* out = p.out + (long)split * p.split_stride + bat * p.bs_out;
    const int RS = p.R * p.S;
    constexpr int EP = 36;
    float* stage = smem + wave * (TM * 32 * EP);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf) * EP + lrow32] = acc[i][j][e];
        const int c = c0 + wn0 + j * 32 + ec;
        // all of the wave's rows first, pinned in straight-line code, then the predicated stores back to back (a store block that
        // waits for its own LDS operand with s_waitcnt vmcnt(0) also waits for the previous store: see the fprop epilogue)
        float4 v[TM * 4];
#pragma unroll
        for (int q = 0; q < TM * 4; ++q) {
            v[q] = ld4(stage + (q * 8 + er) * EP + ec);
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
        }
#pragma unroll
        for (int q = 0; q < TM * 4; ++q) {
            const int k = k0 + wm0 + q * 8 + er;
            if (c < (p.pack4 ? RS * 4 : p.C) && k < p.K) st4(out + ((long)k * RS + tap) * p.C + c, v[q]);   // pack4: tap == 0, c runs over tap*4 + ch
        }
    }
}

// out = sum over the nsplit partial slices, in slice order (deterministic).  Four slices are fetched per step so that a
// thread has four independent 16-byte loads in flight (a one-load-per-iteration loop is latency bound: 1.8 ms per cfg2 step).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, float* out, long n4, int nsplit, long stride4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    const float4* w = reinterpret_cast<const float4*>(ws);
    for (; i < n4; i += step) {
        float4 a = w[i];
        int s = 1;
        // (round 6) eight slices per step first: the 14- and 16-way splits of the Xception / bottleneck filter gradients took four
        // dependent round trips of four loads; the summation order (slice order) is unchanged
        for (; s + 7 < nsplit; s += 8) {
            float4 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = w[i + (long)(s + u) * stride4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
        }
        for (; s + 3 < nsplit; s += 4) {
            const float4 b0 = w[i + (long)s * stride4], b1 = w[i + (long)(s + 1) * stride4];
            const float4 b2 = w[i + (long)(s + 2) * stride4], b3 = w[i + (long)(s + 3) * stride4];
            a.x = (((a.x + b0.x) + b1.x) + b2.x) + b3.x; a.y = (((a.y + b0.y) + b1.y) + b2.y) + b3.y;
            a.z = (((a.z + b0.z) + b1.z) + b2.z) + b3.z; a.w = (((a.w + b0.w) + b1.w) + b2.w) + b3.w;
        }
        for (; s < nsplit; ++s) {
            const float4 b = w[i + (long)s * stride4];
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

// The same transposition for MANY filters in one launch: a training step transposes every convolution filter once for its
// data-gradient pass (66 launches of ~5 us for PSPNet-R50, latency- not bandwidth-bound).  tab[i].tile_begin is the running
// sum of the tensors' 32x32 tile counts; a block finds its tensor by bisection.
__global__ __launch_bounds__(256) void krsc_to_crsk_multi_kernel(const segmi_filter_tx* __restrict__ tab, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
    }
    const segmi_filter_tx t = tab[lo];
    const int RS = t.R * t.S;
    const int tc = (t.C + 31) >> 5, tk = (t.Kpad + 31) >> 5;
    int r = b - t.tile_begin;
    const int c0 = (r % tc) * 32; r /= tc;
    const int k0 = (r % tk) * 32;
    const int tap = r / tk;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + tx;
        tile[j][tx] = (k < t.K && c < t.C) ? t.w_krsc[((long)k * RS + tap) * t.C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, k = k0 + tx;
        if (c < t.C && k < t.Kpad) t.w_crsk[((long)c * RS + tap) * t.Kpad + k] = tile[tx][j];
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
// block = (32 channels, 8 partial lanes): four independent loads per lane and step, LDS tree over the lanes (one thread per
// channel walking up to 512 partials serially cost 40 us per bias gradient: every load waited for the previous add)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 32 + threadIdx.x;
    float a = 0.f;
    if (c < C) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = threadIdx.y;
        for (; i + 24 < nparts; i += 32) {
            a0 += part[(long)i * C + c]; a1 += part[(long)(i + 8) * C + c];
            a2 += part[(long)(i + 16) * C + c]; a3 += part[(long)(i + 24) * C + c];
        }
        for (; i < nparts; i += 8) a0 += part[(long)i * C + c];
        a = (a0 + a1) + (a2 + a3);
    }
    __shared__ float sm[8][33];
    sm[threadIdx.y][threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][threadIdx.x];
        out[c] = t;
    }
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

// the K loop's pointwise form (conv_dma_kernel<..., PW>)
static bool dma_pointwise(bool fast, int RS, int Cs, int sub) { (void)Cs; return fast && RS == 1 && !sub; }

template <int BM, int BN, int WM, int WN, int MODE>
int launch_dma(GatherParams& p, unsigned src_bytes, unsigned wgt_bytes, unsigned dst_bytes, hipStream_t st) {
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SEGMI_CONV_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    p.tiles_m = segmi_cdiv(p.M, BM);
    p.tiles_n = segmi_cdiv(p.Cd, BN);
    size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
    if (p.dbg & 16) {                                          // ablation: one workgroup per CU (LDS request of 100 KB)
        lds = 100 * 1024;
        static bool once = false;
        if (!once) { once = true;
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<BM, BN, WM, WN, MODE, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<BM, BN, WM, WN, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<BM, BN, WM, WN, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
    }
    p.pack4 = (MODE == MODE_FPROP && p.Cs == 4 && p.R * p.S > 1 && p.ksplit <= 1) ? 1 : 0;
    const bool fast = !p.pack4 && p.R * p.S <= 32 && (MODE == MODE_FPROP || p.stride == 1);
    const int Tall = p.pack4 ? segmi_cdiv(p.R * p.S * 4, 32) : segmi_cdiv(p.Cs, 32) * p.R * p.S;
    if (p.ksplit <= 1) { p.ksplit = 1; p.its_per_split = Tall > 0 ? Tall : 1; }
    const dim3 grid((unsigned)p.tiles_m * p.tiles_n, (unsigned)(p.batch > 1 ? p.batch : p.ksplit));
    const bool pw = dma_pointwise(fast, p.R * p.S, p.Cs, p.sub);
    if (pw)        hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, MODE, true, true>), grid, dim3(256), lds, st, p, src_bytes, wgt_bytes, dst_bytes);
    else if (fast) hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, MODE, true>), grid, dim3(256), lds, st, p, src_bytes, wgt_bytes, dst_bytes);
    else           hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, MODE, false>), grid, dim3(256), lds, st, p, src_bytes, wgt_bytes, dst_bytes);
    if (p.ksplit > 1) {
        const long n4 = (long)p.M * p.ldd / 4;
        int rg = (int)((n4 + 255) / 256);
        if (rg > SEGMI_MAX_GRID) rg = SEGMI_MAX_GRID;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, (const float*)p.ws, p.dst, n4, p.ksplit, n4);
    }
    return segmi_launch_status();
}

int g_dma = -1;  // SEGMI_CONV_DMA=0 forces the register-staged kernels (A/B testing)
bool conv_dma() {
    if (g_dma < 0) {
        const char* e = getenv("SEGMI_CONV_DMA");
        g_dma = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_dma == 1;
}
// bytes spanned by an operand if it fits 32-bit buffer offsets (with room for the OOB marker), else 0
unsigned span32(long elems) {
    const long b = elems * (long)sizeof(float);
    return (b > 0 && b < 0xFFFFFF00L) ? (unsigned)b : 0u;
}

// 64-row tiles (twice the workgroups, three resident per CU at 48 KB of LDS each) instead of 128-row tiles:
//   * small problems — fewer 128-row tiles than the 512 workgroups the chip holds (Xception's 728->728 1x1 on 32x32 maps:
//     384 -> 768 tiles);
//   * round 5, wave quantisation: when the LAST per-CU round of the 128-row tiling is less than half full (tiles mod 256 in
//     (0, 128]).  A CU works its resident tiles off at a roughly constant aggregate rate, so a launch costs ceil-ish(tiles / 256)
//     tile times; with 1096 tiles (DeepLab-R101's 256->1024 1x1 on 16 x 33x33 maps) the fifth round runs 72 tiles on 256 CUs.
//     Halving the tile halves the cost of that round.  Measured per layer in one call (profiles/r05_half_m_layers.txt):
//     33x33 C256->K1024 89.8 -> 95.1 TF/s, C512->K2048 dgrad 91.1 -> 107.6, 65x65 C128->K512 dgrad 77.0 -> 90.9; where the last
//     round is full or more than half full the 64-row kernel is 1.5-3 % slower (less filter reuse per LDS byte) and 128 rows stay.
// SEGMI_CONV_HALF_M: 0 never, 1 always (where the kernel exists), 3 the round-4 rule (small problems only); unset = both rules.
int g_half_m = -2;
bool dma_half_m(int M, int Cd, int batch = 1) {      // batch: the 16 contractions of a Winograd pass share one launch
    const int bn = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
    if (bn == 32 || M <= 64) return false;
    if (g_half_m == -2) { const char* e = getenv("SEGMI_CONV_HALF_M"); g_half_m = (e && *e) ? atoi(e) : -1; }
    if (g_half_m == 0) return false;
    if (g_half_m == 1) return true;
    const long tiles = (long)segmi_cdiv(M, 128) * segmi_cdiv(Cd, bn) * (batch > 1 ? batch : 1);
    if (g_half_m == 3) return batch > 1 ? false : tiles < 2L * SEGMI_NUM_CU;       // round 4: batched launches always took 128-row tiles
    if (tiles < 2L * SEGMI_NUM_CU) return true;
    const long rem = tiles % SEGMI_NUM_CU;
    return rem > 0 && rem <= SEGMI_NUM_CU / 2;
}

// 64 x 64 tiles (five resident per CU at 32 KB of LDS each) for launches whose 64 x 128 tiling leaves the chip's 768 slots under-filled
// (round 6; DeepLab-R101's 1024 -> 256 1x1 on 16 x 33x33 maps: 546 tiles of 64 x 128 work at 2.1 tiles per CU with a maximum of 3 —
// 1092 tiles of 64 x 64 at 4.3 of 5: 85.3 -> 94.2 TF/s forward, profiles/r06_quarter_tiles_ab.txt).  SEGMI_CONV_QUARTER: fill threshold in
// percent of the 768 slots (default 75, 0 = never).
int g_quarter = -1;
bool dma_quarter(int M, int Cd, int batch) {
    if (g_quarter < 0) { const char* e = getenv("SEGMI_CONV_QUARTER"); g_quarter = (e && *e) ? atoi(e) : 75; }
    if (g_quarter <= 0 || batch > 1) return false;
    const long tiles = (long)segmi_cdiv(M, 64) * segmi_cdiv(Cd, 128);
    return tiles * 100 <= (long)g_quarter * 3 * SEGMI_NUM_CU;
}

template <int MODE>
int dispatch_gather(GatherParams& p, hipStream_t st) {
    const unsigned sb = span32((long)p.N * p.Hs * p.Ws * p.lds);
    const unsigned wb = span32((long)p.Cd * (p.sub ? p.wRS : p.R * p.S) * p.Cs);   // parity-class launches index the whole filter
    const unsigned db = span32((long)p.N * p.Hd * p.Wd * p.ldd);                  // the tile leaves through buffer stores
    if (conv_dma() && sb && wb && db) {
        const bool half_m = dma_half_m(p.M, p.Cd, p.batch);              // (a batched launch has batch x the tiles)
        if (p.Cd > 64 && half_m && dma_quarter(p.M, p.Cd, p.batch)) return launch_dma<64, 64, 2, 2, MODE>(p, sb, wb, db, st);
        if (p.Cd > 64) return half_m ? launch_dma<64, 128, 2, 2, MODE>(p, sb, wb, db, st) : launch_dma<128, 128, 2, 2, MODE>(p, sb, wb, db, st);
        if (p.Cd > 32) return half_m ? launch_dma<64, 64, 2, 2, MODE>(p, sb, wb, db, st) : launch_dma<128, 64, 2, 2, MODE>(p, sb, wb, db, st);
        return launch_dma<128, 32, 4, 1, MODE>(p, sb, wb, db, st);
    }
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

struct WgradPlan { int bm, bn, tiles_k, tiles_c, nsplit, chunks_per_split, pack4; };
constexpr int WG_BKP = 32;
bool wgrad_dma_desc(const segmi_conv_desc* d);
WgradPlan plan_wgrad(const segmi_conv_desc* d, int batch = 1) {
    WgradPlan pl;
    pl.pack4 = (d->C == 4 && d->R * d->S > 1 && wgrad_dma_desc(d)) ? 1 : 0;   // RGB stems: fold the taps into the N axis
    const int Cv = pl.pack4 ? d->R * d->S * 4 : d->C;
    pl.bm = d->K > 64 ? 128 : 64;
    pl.bn = Cv > 64 ? 128 : 64;
    pl.tiles_k = segmi_cdiv(d->K, pl.bm);
    pl.tiles_c = segmi_cdiv(Cv, pl.bn);
    const long tiles = (long)pl.tiles_k * pl.tiles_c * (pl.pack4 ? 1 : d->R * d->S) * batch;
    const long M = (long)d->N * d->P * d->Q;
    const long chunks = (M + WG_BKP - 1) / WG_BKP;
    // Split the pixel reduction until the grid is >= 8 "rounds" of the 512 resident workgroups (two 64 KB-LDS workgroups
    // per CU): with that many workgroups the dispatcher's dynamic scheduling hides the round quantisation (the PSP
    // bottleneck's 1152 tiles unsplit are 2.25 rounds: 110 TF/s; x4: 125 TF/s), and the partial-sum traffic
    // (nsplit * |dW| * 8 B) stays ~1 % of the kernel.
    const long slots = 2L * SEGMI_NUM_CU;
    long max_split = chunks / 8 > 0 ? chunks / 8 : 1;            // >= 256 pixels per split
    if (max_split > 512) max_split = 512;
    const long ns_rounds = (8 * slots + tiles - 1) / tiles;       // >= 8 rounds of resident workgroups
    // at least one workgroup per slot — unless ONE FEWER split already fills >= 95 % of the slots: 36 tiles (728 -> 728, Xception's
    // middle flow) x 14 splits = 504 of 512 slots is one full round; insisting on 15 made it 540 = two rounds, and the search below
    // then went on to 26 splits of 10 chunks each (62 TF/s: prologue / epilogue dominated)
    long ns_fill = (slots + tiles - 1) / tiles;
    if (ns_fill > 1 && (ns_fill - 1) * tiles * 100 >= slots * 95) --ns_fill;
    const long ns_long = chunks / 64 > 0 ? chunks / 64 : 1;       // but keep >= 64 chunks (2048 pixels) of K loop per workgroup
    long lo = ns_fill > ns_long ? ns_fill : ns_long;              //   unless filling the chip needs more
    long hi = ns_rounds < max_split ? ns_rounds : max_split;
    if (lo > hi) lo = hi;
    if (lo < 1) lo = 1;
    // below 8 rounds the round quantisation is real (576 workgroups on 512 slots run as two rounds): take the first split in
    // [lo, 2*lo] whose last round is >= 97 % full, else the fullest of that window (a larger split only for > 2 % more), and
    // beyond the window the first that is >= 90 % full
    // (the split is realised as an integer number of chunks per workgroup, so the candidates are the distinct ceil(chunks / cps))
    auto splits_of = [&](long cps) { return (chunks + cps - 1) / cps; };
    long cps = (chunks + lo - 1) / lo;
    double best = -1.0;
    for (long c = cps; c >= 1; --c) {
        const long n = splits_of(c);
        if (n > hi && best >= 0.0) break;
        const long wg = tiles * n;
        const long rounds = (wg + slots - 1) / slots;
        const double eff = rounds >= 8 ? 1.0 : (double)wg / (double)(rounds * slots);
        if (eff > best + (best < 0.9 ? 1e-9 : 0.02)) { best = eff; cps = c; }
        if (eff >= 0.97 || (n >= 2 * lo && best >= 0.9)) break;
    }
    if (const char* e = getenv("SEGMI_WGRAD_SPLIT")) {           // tuning hook
        const long f = atol(e);
        if (f >= 1 && f <= max_split) cps = (chunks + f - 1) / f;
    }
    pl.chunks_per_split = (int)cps;
    pl.nsplit = (int)splits_of(cps);
    return pl;
}

// Forward split-K: only for problems whose output has so few tiles that the chip idles while ONE workgroup walks a long
// reduction chunk by chunk (the four pyramid 1x1 convs 2048->512 on 1x1..6x6 maps: M = 8..288 rows, 64 chunks: 140 us each).
struct FwdSplit { int ksplit, its_per_split; };
FwdSplit plan_fwd_split(const segmi_conv_desc* d) {
    FwdSplit f = {1, 0};
    const long M = (long)d->N * d->P * d->Q;
    const int bn = d->K > 64 ? 128 : (d->K > 32 ? 64 : 32);
    const long tiles = (long)segmi_cdiv(M, dma_half_m((int)M, d->K) ? 64 : 128) * segmi_cdiv(d->K, bn);
    const int T = segmi_cdiv(d->C, 32) * d->R * d->S;
    if (tiles > 32 || T < 16 || (d->ldy & 3)) return f;
    long ks = (2L * SEGMI_NUM_CU) / tiles;                     // fill the 512 resident slots
    if (ks > T / 4) ks = T / 4;                                // >= 4 iterations per split
    if (ks > 64) ks = 64;
    if (ks < 2) return f;
    f.its_per_split = (int)((T + ks - 1) / ks);
    f.ksplit = (T + f.its_per_split - 1) / f.its_per_split;
    return f;
}
bool dma_eligible_fwd(const segmi_conv_desc* d) {
    return conv_dma() && span32((long)d->N * d->H * d->W * d->ldx) && span32((long)d->K * d->R * d->S * d->C) &&
           span32((long)d->N * d->P * d->Q * d->ldy);
}

bool wgrad_dma_desc(const segmi_conv_desc* d) {
    return conv_dma() && span32((long)d->N * d->H * d->W * d->ldx) && span32((long)d->N * d->P * d->Q * d->ldy);
}
bool wgrad_dma(const WgradParams& p, unsigned* xb, unsigned* dyb) {
    *xb = span32((long)p.N * p.H * p.W * p.ldx);
    *dyb = span32((long)p.N * p.P * p.Q * p.ldy);
    return conv_dma() && *xb && *dyb;
}

// the filter-gradient kernel's pointwise form (conv_wgrad_dma_kernel<..., PW>)
static bool wgrad_pointwise(int R, int S, int stride, int pad, int M, int pack4) {
    (void)M;
    return R == 1 && S == 1 && stride == 1 && pad == 0 && !pack4;
}

int g_wgrad_flat = -1;  // SEGMI_WGRAD_FLAT=0: the round-3 workgroup order (A/B hook)
int wgrad_flat() {
    if (g_wgrad_flat < 0) {
        const char* e = getenv("SEGMI_WGRAD_FLAT");
        g_wgrad_flat = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_wgrad_flat;
}

int g_wgrad_skiprows = -1;  // SEGMI_WGRAD_SKIPROWS=0: every chunk is issued (A/B hook)
int wgrad_skiprows() {
    if (g_wgrad_skiprows < 0) {
        const char* e = getenv("SEGMI_WGRAD_SKIPROWS");
        g_wgrad_skiprows = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_wgrad_skiprows;
}

template <int BM, int BN>
int launch_wgrad(WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    p.flat = wgrad_flat();
    p.skiprows = 0;
    const size_t lds = (size_t)2 * WG_BKP * (BM + BN) * sizeof(float);
    dim3 grid((unsigned)(pl.tiles_k * pl.tiles_c * (pl.pack4 ? 1 : p.R * p.S)), (unsigned)pl.nsplit, (unsigned)(p.batch > 1 ? p.batch : 1));
    unsigned xb, dyb;
    if (p.batch > 1 && !wgrad_dma(p, &xb, &dyb)) return SEGMI_ERR_BADARG;     // batch exists in the LDS-DMA kernel only
    if (wgrad_dma(p, &xb, &dyb)) {
        // ROWQ needs whole 32-pixel chunks inside one output row and splits that start on a chunk boundary (they do)
        const bool rowq = p.Q % WG_BKP == 0;
        if (wgrad_pointwise(p.R, p.S, p.stride, p.pad, p.M, p.pack4)) {
            if (rowq) hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, true, true>), grid, dim3(256), lds, st, p, xb, dyb);
            else      hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, false, true>), grid, dim3(256), lds, st, p, xb, dyb);
        }
        else if (rowq) hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, true>), grid, dim3(256), lds, st, p, xb, dyb);
        else {
            p.skiprows = wgrad_skiprows() && p.Q >= WG_BKP && p.R * p.S > 1 && !p.pack4 && p.pad > 0;
            hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, false>), grid, dim3(256), lds, st, p, xb, dyb);
        }
    }
    else hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WG_BKP, 2, 2>), grid, dim3(256), lds, st, p);
    return segmi_launch_status();
}

}  // namespace

extern "C" {

size_t segmi_conv2d_fwd_workspace(const segmi_conv_desc* d) {
    if (!desc_ok(d) || !dma_eligible_fwd(d)) return 0;
    const FwdSplit fs = plan_fwd_split(d);
    return fs.ksplit > 1 ? (size_t)fs.ksplit * d->N * d->P * d->Q * d->ldy * sizeof(float) : 0;
}

// Number of {count, mean, M2} partials the forward kernel's BN-statistics epilogue emits for this problem (one per row tile), or 0
// when the launch that would serve it has no such epilogue (register-staged fallback, split reduction of tiny outputs, K % 4)
static int fwd_stats_parts(const segmi_conv_desc* d) {
    if (!desc_ok(d) || !dma_eligible_fwd(d) || (d->K & 3)) return 0;
    if (plan_fwd_split(d).ksplit > 1) return 0;
    const int M = d->N * d->P * d->Q;
    const int bm = (d->K > 32 && dma_half_m(M, d->K)) ? 64 : 128;
    return segmi_cdiv(M, bm);
}

static int conv_fwd_impl(const segmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                         int accumulate, void* workspace, size_t workspace_bytes, segmi_stream_t stream, float* stats = nullptr) {
    if (!desc_ok(d) || !x || !w || !y) return SEGMI_ERR_BADARG;
    if ((d->C & 3) || (d->ldx & 3) || d->ldx < d->C || (d->ldy & 3) || d->ldy < ((d->K + 3) & ~3) || !aligned16(x) ||
        !aligned16(w) || !aligned16(y))
        return SEGMI_ERR_ALIGN;
    GatherParams p;
    p.src = x; p.wgt = w; p.bias = bias; p.dst = y;
    p.N = d->N; p.Hs = d->H; p.Ws = d->W; p.Cs = d->C; p.lds = d->ldx;
    p.Hd = d->P; p.Wd = d->Q; p.Cd = d->K; p.ldd = d->ldy;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.M = d->N * d->P * d->Q; p.accumulate = accumulate;
    p.ksplit = 1; p.its_per_split = 0; p.ws = nullptr; p.sub = 0;
    p.batch = 1; p.bs_src = p.bs_wgt = p.bs_dst = 0;
    p.stats = nullptr; p.stats_cp = 0;
    if (stats) {
        if (accumulate || fwd_stats_parts(d) == 0) return SEGMI_ERR_BADARG;
        p.stats = stats; p.stats_cp = d->K;
    }
    const FwdSplit fs = plan_fwd_split(d);
    if (!stats && fs.ksplit > 1 && !accumulate && !bias && dma_eligible_fwd(d) && workspace) {      // workspace == NULL: caller opts out of the split
        if (workspace_bytes < segmi_conv2d_fwd_workspace(d) || !aligned16(workspace)) return SEGMI_ERR_WORKSPACE;
        p.ksplit = fs.ksplit; p.its_per_split = fs.its_per_split; p.ws = (float*)workspace;
    }
    return dispatch_gather<MODE_FPROP>(p, (hipStream_t)stream);
}

int segmi_conv2d_fwd(const segmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                     int accumulate, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    return conv_fwd_impl(d, x, w, bias, y, accumulate, workspace, workspace_bytes, stream);
}

int segmi_conv2d_fwd_stats_parts(const segmi_conv_desc* d) { return fwd_stats_parts(d); }

int segmi_conv2d_fwd_stats(const segmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* stats_partials,
                           segmi_stream_t stream) {
    if (!stats_partials) return SEGMI_ERR_BADARG;
    return conv_fwd_impl(d, x, w, bias, y, 0, nullptr, 0, stream, stats_partials);
}

int segmi_conv2d_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                       segmi_stream_t stream) {
    if (!desc_ok(d) || !dy || !w_crsk || !dx) return SEGMI_ERR_BADARG;
    // the reduction axis is K here: the caller pads it to a multiple of 4 (Kpad = round_up(K,4) <= ldy)
    const int Kpad = (d->K + 3) & ~3;
    if ((d->ldy & 3) || d->ldy < Kpad || (d->ldx & 3) || d->ldx < ((d->C + 3) & ~3) || !aligned16(dy) ||
        !aligned16(w_crsk) || !aligned16(dx))
        return SEGMI_ERR_ALIGN;
    GatherParams p;
    p.src = dy; p.wgt = w_crsk; p.bias = nullptr; p.dst = dx;
    p.N = d->N; p.Hs = d->P; p.Ws = d->Q; p.Cs = Kpad; p.lds = d->ldy;
    p.Hd = d->H; p.Wd = d->W; p.Cd = d->C; p.ldd = d->ldx;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.M = d->N * d->H * d->W; p.accumulate = accumulate;
    p.ksplit = 1; p.its_per_split = 0; p.ws = nullptr; p.sub = 0;
    p.batch = 1; p.bs_src = p.bs_wgt = p.bs_dst = 0;
    p.stats = nullptr; p.stats_cp = 0;
    const int st = d->stride;
    const bool dma_ok = conv_dma() && span32((long)d->N * d->P * d->Q * d->ldy) && span32((long)d->C * d->R * d->S * Kpad);
    if (st == 1 || !dma_ok || d->R * d->S > 16) return dispatch_gather<MODE_DGRAD>(p, (hipStream_t)stream);
    // stride > 1: one launch per output parity class, visiting only the taps that exist for the class
    for (int ph = 0; ph < st && ph < d->H; ++ph)
        for (int pw = 0; pw < st && pw < d->W; ++pw) {
            GatherParams q = p;
            q.sub = 1; q.ph = ph; q.pw = pw; q.os = st; q.wRS = d->R * d->S;
            q.Hc = (d->H - ph + st - 1) / st; q.Wc = (d->W - pw + st - 1) / st;
            q.M = d->N * q.Hc * q.Wc;
            int nt = 0;
            for (int r = 0; r < d->R; ++r) {
                const int th = ph + d->pad - r * d->dil;
                if (((th % st) + st) % st) continue;
                for (int s2 = 0; s2 < d->S; ++s2) {
                    const int tw = pw + d->pad - s2 * d->dil;
                    if (((tw % st) + st) % st) continue;
                    // floor division (th may be negative: such sources are out of range and masked per row)
                    q.tab_r[nt] = (th >= 0 ? th / st : -((-th + st - 1) / st));
                    q.tab_s[nt] = (tw >= 0 ? tw / st : -((-tw + st - 1) / st));
                    q.tab_w[nt] = r * d->S + s2;
                    ++nt;
                }
            }
            if (nt == 0 && accumulate) continue;    // nothing to add to this class's pixels
            q.R = 1; q.S = nt; q.stride = 1;        // loop logic of the kernel: nt taps, unit stride in the class's row space
            const int rc = dispatch_gather<MODE_DGRAD>(q, (hipStream_t)stream);
            if (rc != SEGMI_OK) return rc;
        }
    return SEGMI_OK;
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
    p.tiles_k = pl.tiles_k; p.tiles_c = pl.tiles_c; p.chunks_per_split = pl.chunks_per_split; p.pack4 = pl.pack4;
    p.split_stride = (long)d->K * d->R * d->S * d->C;
    p.batch = 1; p.bs_x = p.bs_dy = p.bs_out = 0;
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
        const bool dma = conv_dma() && span32((long)d->N * d->H * d->W * d->ldx) && span32((long)d->N * d->P * d->Q * d->ldy);
        if (dma) snprintf(buf, len, "conv_wgrad_dma_kernel<%d, %d, %s, %s> splitk=%d", pl.bm, pl.bn, d->Q % WG_BKP == 0 ? "true" : "false",
                          wgrad_pointwise(d->R, d->S, d->stride, d->pad, 0, pl.pack4) ? "true" : "false", pl.nsplit);
        else snprintf(buf, len, "conv_wgrad_kernel<%d, %d, %d, 2, 2> splitk=%d", pl.bm, pl.bn, WG_BKP, pl.nsplit);
        return SEGMI_OK;
    }
    const int Cs = op == 0 ? d->C : ((d->K + 3) & ~3), Cd = op == 0 ? d->K : d->C;
    const int bn = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
    const long src_elems = op == 0 ? (long)d->N * d->H * d->W * d->ldx : (long)d->N * d->P * d->Q * d->ldy;
    const long dst_elems = op == 0 ? (long)d->N * d->P * d->Q * d->ldy : (long)d->N * d->H * d->W * d->ldx;
    if (conv_dma() && span32(src_elems) && span32((long)Cd * d->R * d->S * Cs) && span32(dst_elems)) {
        const int M = op == 0 ? d->N * d->P * d->Q : d->N * d->H * d->W;
        const bool fast = !(op == 0 && Cs == 4 && d->R * d->S > 1) && d->R * d->S <= 32 && (op == 0 || d->stride == 1 || d->R * d->S <= 16);
        const bool pw = dma_pointwise(fast, d->R * d->S, Cs, op == 1 && d->stride > 1);
        const bool half = dma_half_m(M, Cd);
        const int bnq = (Cd > 64 && half && dma_quarter(M, Cd, 1)) ? 64 : bn;
        snprintf(buf, len, "conv_dma_kernel<%d, %d, %s, %d, %s, %s>", half ? 64 : 128, bnq, bn == 32 ? "4, 1" : "2, 2", op, fast ? "true" : "false", pw ? "true" : "false");
        return SEGMI_OK;
    }
    const int bk = (conv_bk() == 32 && Cs >= 32) ? 32 : 16;
    snprintf(buf, len, "conv_gather_kernel<128, %d, %d, %s, %d>", bn, bk, bn == 32 ? "4, 1" : "2, 2", op);
    return SEGMI_OK;
}

int segmi_filter_krsc_to_crsk(const float* w, float* wt, int K, int R, int S, int C, int Kpad, segmi_stream_t stream) {
    if (!w || !wt || K <= 0 || R <= 0 || S <= 0 || C <= 0 || Kpad < K) return SEGMI_ERR_BADARG;
    dim3 grid(segmi_cdiv(C, 32), segmi_cdiv(Kpad, 32), R * S);
    hipLaunchKernelGGL(krsc_to_crsk_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, K, R * S, C, Kpad);
    return segmi_launch_status();
}

long segmi_filter_tx_tiles(int K, int R, int S, int C, int Kpad) {
    if (K <= 0 || R <= 0 || S <= 0 || C <= 0 || Kpad < K) return 0;
    return (long)segmi_cdiv(C, 32) * segmi_cdiv(Kpad, 32) * R * S;
}

int segmi_filter_krsc_to_crsk_multi(const segmi_filter_tx* table_dev, int n, long total_tiles, segmi_stream_t stream) {
    if (!table_dev || n <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffL) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(krsc_to_crsk_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table_dev, n);
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
    hipLaunchKernelGGL(colsum_final_kernel, dim3(segmi_cdiv(C, 32)), dim3(32, 8), 0, st, (const float*)workspace, parts, C, out);
    return segmi_launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------- internal (C++ linkage, conv_internal.h)
int segmi_internal_gemm_batched(const float* a, int lda, const float* w, float* d, int ldd, int M, int Cs, int Cd, int batch,
                                long bs_a, long bs_w, long bs_d, hipStream_t st) {
    if (!a || !w || !d || M <= 0 || Cs <= 0 || Cd <= 0 || batch < 1 || batch > 65535) return SEGMI_ERR_BADARG;
    if ((Cs & 3) || (lda & 3) || lda < Cs || (ldd & 3) || ldd < ((Cd + 3) & ~3) || (bs_a & 3) || (bs_w & 3) || (bs_d & 3) ||
        !aligned16(a) || !aligned16(w) || !aligned16(d))
        return SEGMI_ERR_ALIGN;
    if (!conv_dma() || !span32((long)M * lda) || !span32((long)Cd * Cs)) return SEGMI_ERR_BADARG;   // batch exists in the LDS-DMA kernel only
    GatherParams p;
    p.src = a; p.wgt = w; p.bias = nullptr; p.dst = d;
    p.N = 1; p.Hs = 1; p.Ws = M; p.Cs = Cs; p.lds = lda;
    p.Hd = 1; p.Wd = M; p.Cd = Cd; p.ldd = ldd;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.dil = 1;
    p.M = M; p.accumulate = 0;
    p.ksplit = 1; p.its_per_split = 0; p.ws = nullptr; p.sub = 0;
    p.batch = batch; p.bs_src = bs_a; p.bs_wgt = bs_w; p.bs_dst = bs_d;
    p.stats = nullptr; p.stats_cp = 0;
    return dispatch_gather<MODE_FPROP>(p, st);
}

int segmi_internal_gemm_variant(int M, int Cs, int Cd, char* buf, size_t len) {
    if (!buf || len < 64) return SEGMI_ERR_BADARG;
    const int bn = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
    const int bm = dma_half_m(M, Cd, 16) ? 64 : 128;                         // the 16 batched contractions of a Winograd pass
    snprintf(buf, len, "conv_dma_kernel<%d, %d, %s, 0, true, %s>", bm, bn, bn == 32 ? "4, 1" : "2, 2", dma_pointwise(true, 1, Cs, 0) ? "true" : "false");
    return SEGMI_OK;
}

// dW_b[K, C] = sum_m DY_b[m, K]^T X_b[m, C] for b < batch (1x1 filter gradients over M rows, M % 32 == 0), ONE launch of the LDS-DMA
// filter-gradient kernel (blockIdx.z = b) with ONE pixel split planned for batch x the tiles; partial sums go to
// ws[nsplit][batch][K][C] (the caller reduces them: conv_winograd.hip folds that into its filter-gradient transform).
static segmi_conv_desc wgrad_batched_desc(int M, int C, int K) {
    segmi_conv_desc q;
    q.N = 1; q.H = 1; q.W = M; q.C = C; q.K = K; q.R = 1; q.S = 1; q.P = 1; q.Q = M;
    q.stride = 1; q.pad = 0; q.dil = 1; q.ldx = C; q.ldy = (K + 3) & ~3;
    return q;
}
int segmi_internal_wgrad_batched_splits(int M, int C, int K, int batch) {
    if (M <= 0 || (M % WG_BKP) || C <= 0 || (C & 3) || K <= 0 || batch < 1 || batch > 65535) return 0;
    const segmi_conv_desc q = wgrad_batched_desc(M, C, K);
    if (!wgrad_dma_desc(&q)) return 0;
    return plan_wgrad(&q, batch).nsplit;
}
int segmi_internal_wgrad_batched(const float* x, const float* dy, float* ws, int M, int C, int K, int batch, hipStream_t st) {
    const int nsplit = segmi_internal_wgrad_batched_splits(M, C, K, batch);
    if (!x || !dy || !ws || nsplit < 1) return SEGMI_ERR_BADARG;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(ws)) return SEGMI_ERR_ALIGN;
    const segmi_conv_desc q = wgrad_batched_desc(M, C, K);
    const WgradPlan pl = plan_wgrad(&q, batch);
    WgradParams p;
    p.x = x; p.dy = dy; p.out = ws;
    p.N = 1; p.H = 1; p.W = M; p.C = C; p.ldx = q.ldx;
    p.P = 1; p.Q = M; p.K = K; p.ldy = q.ldy;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.dil = 1;
    p.M = M;
    p.tiles_k = pl.tiles_k; p.tiles_c = pl.tiles_c; p.chunks_per_split = pl.chunks_per_split; p.pack4 = 0;
    p.batch = batch; p.bs_x = (long)M * q.ldx; p.bs_dy = (long)M * q.ldy; p.bs_out = (long)K * C;
    p.split_stride = (long)batch * K * C;
    if (pl.bm == 128 && pl.bn == 128) return launch_wgrad<128, 128>(p, pl, st);
    if (pl.bm == 128) return launch_wgrad<128, 64>(p, pl, st);
    if (pl.bn == 128) return launch_wgrad<64, 128>(p, pl, st);
    return launch_wgrad<64, 64>(p, pl, st);
}
int segmi_internal_wgrad_batched_variant(int M, int C, int K, int batch, char* buf, size_t len) {
    const int nsplit = segmi_internal_wgrad_batched_splits(M, C, K, batch);
    if (!buf || len < 64 || nsplit < 1) return SEGMI_ERR_BADARG;
    snprintf(buf, len, "conv_wgrad_dma_kernel<%d, %d, true, true> splitk=%d", K > 64 ? 128 : 64, C > 64 ? 128 : 64, nsplit);
    return SEGMI_OK;
}

bool segmi_internal_gemm_ok(long M, int lda, int Cs, int Cd) {
    return M > 0 && M < (1L << 31) && Cs > 0 && Cd > 0 && !(Cs & 3) && !(lda & 3) && lda >= Cs && conv_dma() && span32(M * lda) &&
           span32((long)Cd * Cs) && span32(M * (long)((Cd + 3) & ~3));      // (the product's rows are Cd rounded up to 4 wide)
}
