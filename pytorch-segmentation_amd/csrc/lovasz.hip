// Lovasz-Softmax loss (classes='present', per_image=False), forward and backward.
// Replaces utils/losses.py:79-89 -> utils/lovasz_losses.py:153-218 (flatten_probas, lovasz_softmax_flat) and
// lovasz_grad :19-31 of the reference, which run softmax + C x (torch.sort over all valid pixels, 2 cumsums, dot)
// and whose backward scatters through C `probas[:, c]` selects (O(C^2 * P) traffic on the reference).
//
// TAIL PRUNING (round 5).  In lovasz_grad (:19-31) every element ranked after the class's LAST foreground element has
// jaccard[i] - jaccard[i-1] = 1 - 1 = 0 EXACTLY (intersection = gts - cumsum(fg) = 0 from there on): it adds 0 to the dot at
// :198 and receives gradient 0.  With thr[c] = min over the foreground pixels of class c of their error, only elements with
// err >= thr[c] (ties kept) can matter, and they are a PREFIX of the class's descending order — ranks, chunk boundaries and
// the summation order of the survivors are those of the full sort, so loss and gradient are bit-identical to it.  On
// random-init and on 80 %-accurate logits the survivors are ~0.7 % of the C x P elements (C = 150); a foreground pixel
// with error 0 (p rounds to 1) keeps its whole class, i.e. the formulation degrades to the full sort.
//
// MI355X formulation — all HBM-bound streaming, no host synchronisation (grids are sized for the worst case; workgroups
// beyond a class's survivor count leave at once):
//   1. lovasz_prepare     per pixel: log-sum-exp; histogram of labels (class presence, |fg_c|), number of valid pixels,
//                         thr[c] = atomicMin over fg pixels of the error bits (LDS first, then global)
//   2. lovasz_keep_count  per (class, 256-pixel unit): number of survivors (keep test in the logit domain, see lov_xthr: no exp);
//                         lovasz_keep_scan: exclusive scan per class
//   3. lovasz_emit        one 64-bit word per SURVIVOR, compacted class-major in pixel order (deterministic: unit offsets +
//                         ballot ranks):  [class | invalid(0) | ~bits30(|fg - p_c|) | fg | pixel index]
//                         (errors lie in [0, 2): their float bits fit 30 bits).
//   4. a SEGMENTED least-significant-digit radix sort, hand-written (segsort_* below) over each class's n_kept[c] keys: the
//      class is implicit in the segment, so only the 31-bit field [invalid | ~error] is sorted — four stable 8-bit passes
//      (histogram per 4096-key tile -> per-class scan -> scatter through an LDS-sorted tile so that every digit's run leaves as
//      one contiguous write), payload (fg bit, pixel index) riding inside the 64-bit key.
//   5. lovasz_chunk_scan / lovasz_grad_dot: two-level scan of the sorted fg bits -> Jaccard index at every rank in the same
//      float32 arithmetic as lovasz_grad (integers are exact in fp32 below 2^24 pixels), first difference, dot with the
//      sorted errors, and scatter of d loss / d p into the pixel-major G[pixel][class] — SURVIVOR entries only.
//   6. lovasz_finalize    loss = mean over present classes; survivor statistics; effective thresholds for the backward.
//   backward: dz_c = g * p_c * (G_c - sum_j G_j p_j) / n_present   (softmax Jacobian).  G is read ONLY where the forward's
//   keep test (recomputed from logits, lse, target and the saved logit-domain thresholds: same subtraction, same compare) says an
//   entry was written; everything else is 0 by the identity above — G is never cleared and never read densely.
//
// Ties: elements of one class with bit-equal errors are ranked in pixel order (stable sort of a pixel-ordered emission;
// torch.sort is unstable); the loss value does not depend on that order, the per-pixel gradients inside a tie group do.
#include "segmi_common.h"
#include "bilinear.h"
#include <cstdlib>
#include <cstring>

// pool_resize.hip: height pass of the separable bilinear backward on a width-reduced buffer (internal, C++ linkage)
int segmi_internal_bilinear_bwd_height(const float* tmp, int ldt, float* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                                       int ac, hipStream_t st);

namespace {

constexpr int CHUNK = 2048;   // ranks per scan block (256 threads x 8)

constexpr int UPX = 256;     // pixels per compaction unit = one wave's contiguous share of a 1024-pixel block
constexpr unsigned NO_THR = 0xFFFFFFFFu;   // threshold of a class without foreground: nothing survives (error bits are < 2^30)

// bits of |fg - p_c| with p_c = exp(z - lse): the ONE expression prepare / count / emit / backward all evaluate (no contraction
// across the subtractions), so that every pass takes the same keep decision for an element
__device__ __forceinline__ float lov_p(float z, float l) { return expf(__fsub_rn(z, l)); }
__device__ __forceinline__ unsigned lov_p_err_bits(float p, bool fg) { return __float_as_uint(fabsf(__fsub_rn(fg ? 1.f : 0.f, p))); }
__device__ __forceinline__ unsigned lov_err_bits(float z, float l, bool fg) { return lov_p_err_bits(lov_p(z, l), fg); }
// THE KEEP TEST, IN THE LOGIT DOMAIN (round 6).  A foreground element always survives (thr[c] is the minimum over the class's
// foreground errors).  A background element's error is p itself, so "p >= thr" is "z - lse >= log thr": ONE subtraction and a
// compare instead of an exp per element in the count, emit and dot passes (they were as ALU-bound — 315 M expf at cfg5 — as they
// are HBM-bound).  xthr = log(thr) lowered by more than the rounding of expf and logf, so that every element with
// expf(z - lse) >= thr passes: the survivors are a SUPERSET of the exact prefix, and the extra elements (p within ~1e-6 relative
// below thr) rank after the class's last foreground element, where the Jaccard difference is exactly 0 — loss and gradient stay
// bit-identical to the full sort.  All passes compare the SAME float d = __fsub_rn(z, lse) with the SAME xthr[c] (computed once, by
// the count kernel, and handed on through the workspace and loss_out), so they decide alike.
__device__ __forceinline__ float lov_xthr(unsigned thr_eff) {
    if (thr_eff == NO_THR) return INFINITY;              // absent class: nothing survives
    if (thr_eff == 0u) return -INFINITY;                 // pruning off, or a foreground probability that rounds to 1: every valid pixel
    const float lg = logf(__uint_as_float(thr_eff));     // thr in (0, 1]: lg <= 0
    return lg - (fabsf(lg) * 1e-6f + 4e-6f);
}
__device__ __forceinline__ bool lov_keep(float d, bool fg, float xthr) { return fg || d >= xthr; }
// effective threshold of class c: absent classes keep nothing; prune == 0 keeps every valid pixel of a present class (the
// round-4 full sort, kept for A/B and as the bit-identity reference of the tests)
__device__ __forceinline__ unsigned lov_thr_eff(const unsigned* __restrict__ thr, const unsigned* __restrict__ counts, int c, int prune) {
    return counts[c] == 0 ? NO_THR : (prune ? thr[c] : 0u);
}

// WHERE A PIXEL'S LOGITS COME FROM (round 6).  UP == false: row r of a [rows, ld] matrix.  UP == true: the loss is evaluated on the
// model's final bilinear upsample of low-resolution logits [N, H, W, C] to [N, OH, OW] (models/deeplabv3_plus.py:361,
// models/pspnet.py:85-91 feed F.interpolate's result to the loss, trainer.py:56-66) WITHOUT that tensor ever existing: every pass
// interpolates the four low-resolution neighbours of its pixel on the fly — at cfg5 they are 79 MB and stay in the L2 / MALL,
// where the upsampled tensor is 1.26 GB that the forward read three times and the backward once more.  The interpolation is ONE
// expression with explicit roundings (lov_up4: the operation order bilinear_fwd_kernel of pool_resize.hip compiles to), so every
// pass — and the unfused path — sees the same bits, which the keep test relies on.
struct LovSrc {
    const float* base; int ld;
    int H, W, OH, OW, ac;        // UP only
    float sh, sw;                // bl_scale of the two axes (UP only)
};
__device__ __forceinline__ float lov_lerp(float w0, float v0, float w1, float v1) { return fmaf(w0, v0, __fmul_rn(w1, v1)); }
template <bool UP>
struct LovPix {
    const float *p00, *p01, *p10, *p11;
    float a0, a1, b0, b1;
    __device__ __forceinline__ void init(const LovSrc& s, long r) {
        if (!UP) { p00 = s.base + r * s.ld; return; }
        const unsigned ur = (unsigned)r;                 // rows < 2^24 (lovasz_layout)
        const unsigned tq = ur / (unsigned)s.OW, ow = ur - tq * (unsigned)s.OW;
        const unsigned n = tq / (unsigned)s.OH, oh = tq - n * (unsigned)s.OH;
        const Lerp a = bl_src((int)oh, s.sh, s.H, s.ac), b = bl_src((int)ow, s.sw, s.W, s.ac);
        const float* img = s.base + (long)n * s.H * s.W * s.ld;
        p00 = img + ((long)a.i0 * s.W + b.i0) * s.ld; p01 = img + ((long)a.i0 * s.W + b.i1) * s.ld;
        p10 = img + ((long)a.i1 * s.W + b.i0) * s.ld; p11 = img + ((long)a.i1 * s.W + b.i1) * s.ld;
        a0 = a.l0; a1 = a.l1; b0 = b.l0; b1 = b.l1;
    }
    __device__ __forceinline__ float4 get(int q) const {
        if (!UP) return ld4(p00 + q * 4);
        const float4 v00 = ld4(p00 + q * 4), v01 = ld4(p01 + q * 4), v10 = ld4(p10 + q * 4), v11 = ld4(p11 + q * 4);
        float4 o;
        o.x = lov_lerp(a0, lov_lerp(b0, v00.x, b1, v01.x), a1, lov_lerp(b0, v10.x, b1, v11.x));
        o.y = lov_lerp(a0, lov_lerp(b0, v00.y, b1, v01.y), a1, lov_lerp(b0, v10.y, b1, v11.y));
        o.z = lov_lerp(a0, lov_lerp(b0, v00.z, b1, v01.z), a1, lov_lerp(b0, v10.z, b1, v11.z));
        o.w = lov_lerp(a0, lov_lerp(b0, v00.w, b1, v01.w), a1, lov_lerp(b0, v10.w, b1, v11.w));
        return o;
    }
    __device__ __forceinline__ float at(int c) const {
        if (!UP) return p00[c];
        const float4 v = get(c >> 2);
        const int o = c & 3;
        return o == 0 ? v.x : o == 1 ? v.y : o == 2 ? v.z : v.w;
    }
};

// A pixel's logits row held in registers by the 8 lanes that share the pixel: lane g owns the float4 groups g, g + 8, ...  KQ > 0:
// ceil(C/32) <= KQ <= 8 groups per lane, ALL loaded up front (KQ 16-byte loads in flight per lane instead of one; the second
// sweep of a kernel re-uses the registers instead of re-reading L1/L2) — the streaming passes went from 2.5-3.3 to 4-5 TB/s at
// C = 150.  KQ == 0: more than 256 classes, the groups are re-read from memory in every sweep (the round-4 form).
template <int KQ>
struct LovRow {
    float4 v[KQ > 0 ? KQ : 1];
    template <class P>
    __device__ __forceinline__ void load(const P& row, int g, int c4n) {
        if (KQ > 0) {
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int q = g + 8 * k;
                v[k] = q < c4n ? row.get(q) : zero4();
            }
        }
    }
    template <class P>
    __device__ __forceinline__ float4 get(const P& row, int k, int q) const { return KQ > 0 ? v[k] : row.get(q); }
};
// trips of lane g over its groups (KQ > 0: compile-time, groups past the row are zeros and every use is guarded by c < C)
#define LOV_TRIPS(KQ, g, c4n) (KQ > 0 ? KQ : ((c4n) - (g) + 7) / 8)

// 8 lanes share a pixel (float4 channel groups, xor-shuffle reductions): coalesced 128-byte row segments
template <int KQ, bool UP>
__global__ __launch_bounds__(256) void lovasz_prepare_kernel(const LovSrc src, const int64_t* __restrict__ target,
                                                             long rows, int C, long ignore, float* __restrict__ lse,
                                                             unsigned* __restrict__ counts /* [C] fg counts, [C] n_valid */,
                                                             unsigned* __restrict__ thr /* [C], pre-set to NO_THR */) {
    extern __shared__ unsigned hist[];   // C + 1 counters, C thresholds
    unsigned* thr_s = hist + C + 1;
    for (int i = threadIdx.x; i <= C; i += 256) hist[i] = 0;
    for (int i = threadIdx.x; i < C; i += 256) thr_s[i] = NO_THR;
    __syncthreads();
    const int g = threadIdx.x & 7;
    const int c4n = (C + 3) >> 2;
    const int trips = LOV_TRIPS(KQ, g, c4n);
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += (long)gridDim.x * 32) {
        LovPix<UP> row;
        row.init(src, r);
        LovRow<KQ> R;
        R.load(row, g, c4n);
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < trips; ++k) {
            const int q = g + 8 * k, c = q * 4;
            const float4 v = R.get(row, k, q);
            if (c < C) m = fmaxf(m, v.x);
            if (c + 1 < C) m = fmaxf(m, v.y);
            if (c + 2 < C) m = fmaxf(m, v.z);
            if (c + 3 < C) m = fmaxf(m, v.w);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < trips; ++k) {
            const int q = g + 8 * k, c = q * 4;
            const float4 v = R.get(row, k, q);
            if (c < C) s += expf(v.x - m);
            if (c + 1 < C) s += expf(v.y - m);
            if (c + 2 < C) s += expf(v.z - m);
            if (c + 3 < C) s += expf(v.w - m);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (g == 0) {
            const float l = m + logf(s);
            lse[r] = l;
            const long t = target[r];
            if (t != ignore) {
                atomicAdd(&hist[C], 1u);
                if (t >= 0 && t < C) {
                    atomicAdd(&hist[(int)t], 1u);
                    atomicMin(&thr_s[(int)t], lov_err_bits(row.at((int)t), l, true));   // unsigned order == float order for errors >= 0
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= C; i += 256)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
    for (int i = threadIdx.x; i < C; i += 256)
        if (thr_s[i] != NO_THR) atomicMin(&thr[i], thr_s[i]);
}

// cnt[c][unit] = number of survivors of class c among the unit's UPX pixels.  Same thread layout as prepare (a wave reads 8
// pixel rows per step as 128-byte segments); a survivor costs one LDS atomic on its wave's counter row.
template <int KQ, bool UP>
__global__ __launch_bounds__(256) void lovasz_keep_count_kernel(const LovSrc src, const int64_t* __restrict__ target,
                                                                const float* __restrict__ lse, long rows, int C, long ignore,
                                                                const unsigned* __restrict__ thr, const unsigned* __restrict__ counts,
                                                                int prune, long nunits, unsigned* __restrict__ cnt, float* __restrict__ xthr_out) {
    extern __shared__ unsigned sh[];     // xthr_s[C], run[4][C]
    float* xthr_s = reinterpret_cast<float*>(sh);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane & 7, pj = lane >> 3;
    unsigned* run = sh + C + w * C;
    for (int i = threadIdx.x; i < C; i += 256) {
        const float xt = lov_xthr(lov_thr_eff(thr, counts, i, prune));
        xthr_s[i] = xt;
        if (blockIdx.x == 0) xthr_out[i] = xt;           // for the emit pass and (through loss_out) the backward
    }
    for (int i = threadIdx.x; i < 4 * C; i += 256) sh[C + i] = 0u;
    __syncthreads();
    const long unit = (long)blockIdx.x * 4 + w;
    const int c4n = (C + 3) >> 2;
    const int trips = LOV_TRIPS(KQ, g, c4n);
    if (unit < nunits) {
        const long r0 = unit * UPX;
#pragma unroll 2
        for (int it = 0; it < UPX / 8; ++it) {
            const long r = r0 + it * 8 + pj;
            if (r >= rows) continue;
            const long t = target[r];
            if (t == ignore) continue;
            LovPix<UP> row;
            row.init(src, r);
            LovRow<KQ> R;
            R.load(row, g, c4n);
            const float l = lse[r];
#pragma unroll
            for (int k = 0; k < trips; ++k) {
                const int q = g + 8 * k, c = q * 4;
                const float4 v = R.get(row, k, q);
                if (c < C && lov_keep(__fsub_rn(v.x, l), t == c, xthr_s[c])) atomicAdd(&run[c], 1u);
                if (c + 1 < C && lov_keep(__fsub_rn(v.y, l), t == c + 1, xthr_s[c + 1])) atomicAdd(&run[c + 1], 1u);
                if (c + 2 < C && lov_keep(__fsub_rn(v.z, l), t == c + 2, xthr_s[c + 2])) atomicAdd(&run[c + 2], 1u);
                if (c + 3 < C && lov_keep(__fsub_rn(v.w, l), t == c + 3, xthr_s[c + 3])) atomicAdd(&run[c + 3], 1u);
            }
        }
    }
    __syncthreads();
    if (unit < nunits)
        for (int c = lane; c < C; c += 64) cnt[(long)c * nunits + unit] = run[c];
}

// in place: cnt[c][:] -> exclusive prefix over the units (= first slot of the unit's survivors in the class segment);
// nkept[c] = the class's survivor count.  One block per class; a thread owns a contiguous range of units.
constexpr int KS_T = 1024;
__global__ __launch_bounds__(KS_T) void lovasz_keep_scan_kernel(unsigned* __restrict__ cnt, long nunits, unsigned* __restrict__ nkept) {
    const int c = blockIdx.x;
    unsigned* a = cnt + (long)c * nunits;
    const long per = (nunits + KS_T - 1) / KS_T;
    const long b = min(nunits, (long)threadIdx.x * per), e = min(nunits, b + per);
    unsigned s = 0;
    for (long i = b; i < e; ++i) s += a[i];
    __shared__ unsigned sm[KS_T];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < KS_T; o <<= 1) {
        const unsigned y = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0u;
        __syncthreads();
        sm[threadIdx.x] += y;
        __syncthreads();
    }
    unsigned run = sm[threadIdx.x] - s;
    for (long i = b; i < e; ++i) {
        const unsigned v = a[i];
        a[i] = run;
        run += v;
    }
    if (threadIdx.x == KS_T - 1) nkept[c] = sm[KS_T - 1];
}

// Survivors -> keys[c * rows + slot], slot = unit offset + rank in pixel order inside the unit.  A wave walks its unit 8 pixels
// at a time; the 8 lanes holding the same class (same lane & 7, same float4 component) rank themselves with one ballot, the
// wave's running slot of the class lives in LDS (a wave's LDS operations execute in order: every lane reads the slot before
// the group's first lane advances it).
template <int KQ, bool UP>
__global__ __launch_bounds__(256) void lovasz_emit_kernel(const LovSrc src, const int64_t* __restrict__ target,
                                                          const float* __restrict__ lse, long rows, int C, long ignore, int PB,
                                                          const float* __restrict__ xthr,
                                                          long nunits, const unsigned* __restrict__ cnt, unsigned long long* __restrict__ keys) {
    extern __shared__ unsigned sh[];     // xthr_s[C], slot[4][C]
    float* xthr_s = reinterpret_cast<float*>(sh);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane & 7, pj = lane >> 3;
    volatile unsigned* slot = sh + C + w * C;
    const long unit = (long)blockIdx.x * 4 + w;
    for (int i = threadIdx.x; i < C; i += 256) xthr_s[i] = xthr[i];
    if (unit < nunits)
        for (int c = lane; c < C; c += 64) slot[c] = cnt[(long)c * nunits + unit];
    __syncthreads();
    if (unit >= nunits) return;
    const int c4n = (C + 3) >> 2;
    const unsigned long long same = 0x0101010101010101ull << g;       // the lanes of my class group
    const unsigned long long below = (1ull << lane) - 1ull;
    const long r0 = unit * UPX;
#pragma unroll 1
    for (int it = 0; it < UPX / 8; ++it) {
        const long r = r0 + it * 8 + pj;
        const bool rok = r < rows;
        const long t = rok ? target[r] : ignore;
        const bool valid = rok && t != ignore;
        if (__ballot(valid) == 0ull) continue;
        const float l = valid ? lse[r] : 0.f;
        LovPix<UP> row;
        row.init(src, valid ? r : 0);
        LovRow<KQ> R;
        if (valid) R.load(row, g, c4n);
        const int trips = KQ > 0 ? KQ : (c4n + 7) / 8;                 // uniform trip count: the ballots below see the whole wave
#pragma unroll
        for (int k = 0; k < trips; ++k) {
            const int q = g + 8 * k, c = q * 4;
            const bool qok = valid && q < c4n;
            float4 v = zero4();
            if (qok) v = R.get(row, k, q);
            const float zs[4] = {v.x, v.y, v.z, v.w};
            bool kp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kp[j] = qok && c + j < C && lov_keep(__fsub_rn(zs[j], l), t == c + j, xthr_s[c + j]);
            if (__ballot(kp[0] || kp[1] || kp[2] || kp[3]) == 0ull) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long m = __ballot(kp[j]) & same;
                if (kp[j]) {
                    const int cc = c + j;
                    const unsigned rank = (unsigned)__popcll(m & below), n = (unsigned)__popcll(m);
                    const unsigned before = slot[cc];
                    if (rank == 0) slot[cc] = before + n;
                    const bool fg = t == cc;
                    const unsigned eb = lov_err_bits(zs[j], l, fg);              // the exp: survivors only
                    const unsigned long long inv30 = (unsigned long long)((~eb) & 0x3FFFFFFFu);
                    keys[(long)cc * rows + before + rank] = ((((unsigned long long)cc << 1) << 30 | inv30) << 1 | (fg ? 1ull : 0ull)) << PB |
                                                            (unsigned long long)r;
                }
            }
        }
    }
}

// ---- segmented LSD radix sort of the class-major key array: keys[c * rows + i], i < rows, sorted per class by the digit field
constexpr int ST = 4096;      // keys per sort tile: 256 threads x 16 rounds (8192-key tiles, 73 KB of LDS, two blocks per CU: 3 % slower)
constexpr int SEG_GX = 64;    // tiles of a class walked concurrently (grid x extent of the sort passes and of the Jaccard pass)
constexpr int SEG_MLP = 8;    // key loads a wave keeps in flight (16 rounds in groups of SEG_MLP; the ballots of a group stay in SGPRs)

__device__ __forceinline__ unsigned seg_digit(unsigned long long k, int shift, unsigned mask) { return (unsigned)(k >> shift) & mask; }

// lanes of this wave holding the same digit: rank of this lane among them (in lane order = key order) and their number
__device__ __forceinline__ void wave_peers(unsigned d, bool valid, int lane, unsigned& rank, unsigned& count) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
    count = (unsigned)__popcll(peers);
}

// hist[(c * ntiles + tile) * 256 + d] = number of keys of tile `tile` of class c whose digit is d (tiles below the class's
// survivor count nkept[c]; the grid is sized for the worst case, the other workgroups leave at once).
// A histogram does not care about order: one LDS atomic per key into a per-WAVE histogram (four 1 KB rows, summed at the end)
// instead of the ballot ranking the stable scatter needs — the ballots (9 per 64 keys) made this pass as compute-bound as it is
// memory-bound (0.75 ms per pass for cfg5's 2.5 GB of keys = 3.3 TB/s); same-address atomics of a digit most keys share (the
// exponent pass) serialise inside one ds instruction, which is still ~4x cheaper than the ballots.
__global__ __launch_bounds__(256) void segsort_hist_kernel(const unsigned long long* __restrict__ keys, long rows, int ntiles, int shift,
                                                           unsigned mask, const unsigned* __restrict__ nkept, unsigned* __restrict__ hist) {
    const int c = blockIdx.y;
    const long n = nkept[c];                             // survivors of the class (0 for an absent class)
    __shared__ unsigned h[4][256];
    const unsigned long long* k = keys + (long)c * rows;
    unsigned* hw = h[threadIdx.x >> 6];
    // the grid's x extent is capped (SEG_GX): a workgroup walks tiles blockIdx.x, + gridDim.x, ... below the survivor count — a
    // worst-case (ntiles x C) grid cost ~0.5 ns per empty workgroup, 37 us per launch at cfg5 for ~500 tiles of work
    for (int tile = blockIdx.x; (long)tile * ST < n; tile += gridDim.x) {
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0; h[2][threadIdx.x] = 0; h[3][threadIdx.x] = 0;
    __syncthreads();
    const long i0 = (long)tile * ST;
#pragma unroll 1
    for (int r0 = 0; r0 < ST / 256; r0 += SEG_MLP) {     // SEG_MLP 512-byte loads per wave in flight (one at a time is latency-bound: 1 TB/s)
        unsigned long long kq[SEG_MLP];
        bool vq[SEG_MLP];
#pragma unroll
        for (int u = 0; u < SEG_MLP; ++u) {
            const long i = i0 + (r0 + u) * 256 + threadIdx.x;
            vq[u] = i < n;
            kq[u] = vq[u] ? k[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < SEG_MLP; ++u)
            if (vq[u]) atomicAdd(&hw[seg_digit(kq[u], shift, mask)], 1u);
    }
    __syncthreads();
    hist[((long)c * ntiles + tile) * 256 + threadIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    }
}

// in place: hist[c][tile][d] -> first output rank (inside the class segment) of that tile's keys with digit d:
// exclusive scan in (digit, tile) order.  One block per class, 256 digits x SCAN_Q tile ranges: thread (d, q) walks its quarter
// of digit d's column (coalesced across d), the quarters and the digits are combined through LDS.  (One thread per digit walking
// all 512 tiles twice: 183 us per pass on 150 of the chip's 256 CUs.)
constexpr int SCAN_Q = 4;
__global__ __launch_bounds__(256 * SCAN_Q) void segsort_scan_kernel(unsigned* __restrict__ hist, int ntiles_max, const unsigned* __restrict__ nkept) {
    const int c = blockIdx.x, d = threadIdx.x, q = threadIdx.y;
    const long n = nkept[c];
    if (n == 0) return;
    unsigned* col = hist + (long)c * ntiles_max * 256 + d;
    const int ntiles = (int)((n + ST - 1) / ST);         // tiles in use
    const int per = (ntiles + SCAN_Q - 1) / SCAN_Q;
    const int tb = min(ntiles, q * per), te = min(ntiles, tb + per);
    unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int t = tb;
    for (; t + 3 < te; t += 4) {
        t0 += col[(long)t * 256]; t1 += col[(long)(t + 1) * 256]; t2 += col[(long)(t + 2) * 256]; t3 += col[(long)(t + 3) * 256];
    }
    for (; t < te; ++t) t0 += col[(long)t * 256];
    __shared__ unsigned part[SCAN_Q][256], sm[256];
    part[q][d] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    unsigned tot = 0;
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < SCAN_Q; ++k) tot += part[k][d];
        sm[d] = tot;
    }
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned y = 0;
        if (q == 0 && d >= o) y = sm[d - o];
        __syncthreads();
        if (q == 0) sm[d] += y;
        __syncthreads();
    }
    if (q == 0) sm[d] -= tot;                            // keys of this class with a smaller digit
    __syncthreads();
    unsigned run = sm[d];
    for (int k = 0; k < q; ++k) run += part[k][d];       // ... plus this digit's keys in the earlier tile ranges
    for (t = tb; t < te; ++t) {
        const unsigned v = col[(long)t * 256];
        col[(long)t * 256] = run;
        run += v;
    }
}

// stable scatter of one tile.  Each WAVE owns a contiguous quarter of the tile (1024 keys = 16 rounds of 64).  ONE read of the
// tile: a lane keeps its 16 keys in registers and, per key, its position among the wave's keys of the same digit (keys of that
// digit in the wave's earlier rounds + rank among the lanes of this round holding it: ballots) — 16 bits.  After the block has
// turned the four per-wave histograms into first slots (exclusive scan over digits, then over waves), placement is one LDS
// read + one LDS write per key with no ballots and no second pass over the tile (round 3 recomputed the ballots while re-reading
// the tile from L2: 2.09 ms per pass for cfg5's 314 M keys).  The tile is assembled digit by digit in LDS, then every digit's
// run is written to its global position as consecutive elements.
template <int R0>
__device__ __forceinline__ void seg_rank_rounds(const unsigned long long* __restrict__ k, long i0, int nk, int w, int lane, int shift,
                                                unsigned mask, unsigned* wrow, unsigned long long (&kq)[ST / 256], unsigned (&pre)[ST / 256 / 2]) {
    constexpr int WQ = ST / 4;
    bool vq[SEG_MLP];
#pragma unroll
    for (int u = 0; u < SEG_MLP; ++u) {                  // SEG_MLP 512-byte loads per wave in flight
        const int i = w * WQ + (R0 + u) * 64 + lane;
        vq[u] = i < nk;
        kq[R0 + u] = vq[u] ? k[i0 + i] : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < SEG_MLP; ++u) {
        const unsigned d = seg_digit(kq[R0 + u], shift, mask);
        unsigned rank, n;
        wave_peers(d, vq[u], lane, rank, n);
        const unsigned before = wrow[d];                 // every lane reads before the digit's first lane adds this round's count
        if (vq[u] && rank == 0) wrow[d] = before + n;    // (a wave's LDS operations execute in order)
        const unsigned pos = before + rank;              // < 1024
        if (((R0 + u) & 1) == 0) pre[(R0 + u) >> 1] = pos; else pre[(R0 + u) >> 1] |= pos << 16;
        __builtin_amdgcn_sched_barrier(0);               // one round's ballot masks at a time (SGPRs)
    }
}

__global__ __launch_bounds__(256) void segsort_scatter_kernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                              long rows, int ntiles, int shift, unsigned mask,
                                                              const unsigned* __restrict__ nkept, const unsigned* __restrict__ hist,
                                                              unsigned* __restrict__ chunk_fg, int nchunks, int PB) {
    const int c = blockIdx.y;
    const long n = nkept[c];
    constexpr int WQ = ST / 4;                         // keys per wave
    constexpr int RW = WQ / 64;                        // rounds per wave
    static_assert(RW == 2 * SEG_MLP, "two load groups per wave");
    __shared__ unsigned long long sorted[ST];
    __shared__ unsigned lds_u[6][256];                 // rows 0-3: wbase[w][d], 4: scan_s, 5: gbase — rows 1-4 are dead after the
    unsigned (*wbase)[256] = lds_u;                    // placement and hold the last pass's SEG_FG_LDS chunk counters
    unsigned* scan_s = lds_u[4];
    unsigned* gbase = lds_u[5];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long* k = in + (long)c * rows;
    for (int tile = blockIdx.x; (long)tile * ST < n; tile += gridDim.x) {     // capped grid, see segsort_hist_kernel
    __syncthreads();                                   // the previous tile's LDS readers are done
    const long i0 = (long)tile * ST;
    const int nk = (int)(n - i0 < (long)ST ? n - i0 : (long)ST);
    wbase[0][tid] = 0; wbase[1][tid] = 0; wbase[2][tid] = 0; wbase[3][tid] = 0;
    gbase[tid] = hist[((long)c * ntiles + tile) * 256 + tid];
    __syncthreads();
    // phase 1: keys -> registers, per-wave digit histograms, position of every key among its wave's keys of the same digit
    unsigned long long kq[RW];
    unsigned pre[RW / 2];
    seg_rank_rounds<0>(k, i0, nk, w, lane, shift, mask, wbase[w], kq, pre);
    seg_rank_rounds<SEG_MLP>(k, i0, nk, w, lane, shift, mask, wbase[w], kq, pre);
    __syncthreads();
    {   // wbase[w][d] <- first slot of (wave w, digit d) in the tile: exclusive scan of the tile histogram over digits, then over waves
        const unsigned h0 = wbase[0][tid], h1 = wbase[1][tid], h2 = wbase[2][tid], h3 = wbase[3][tid];
        const unsigned v = h0 + h1 + h2 + h3;
        scan_s[tid] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned y = tid >= o ? scan_s[tid - o] : 0u;
            __syncthreads();
            scan_s[tid] += y;
            __syncthreads();
        }
        const unsigned start = scan_s[tid] - v;
        wbase[0][tid] = start; wbase[1][tid] = start + h0; wbase[2][tid] = start + h0 + h1; wbase[3][tid] = start + h0 + h1 + h2;
    }
    __syncthreads();
    // phase 2: placement from registers
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = w * WQ + r * 64 + lane;
        const unsigned d = seg_digit(kq[r], shift, mask);
        const unsigned pos = (r & 1) ? pre[r >> 1] >> 16 : pre[r >> 1] & 0xFFFFu;
        if (i < nk) sorted[wbase[w][d] + pos] = kq[r];
    }
    __syncthreads();
    unsigned long long* o = out + (long)c * rows;
    // last pass (chunk_fg != null): pos is the key's final rank — count the foreground keys per scan chunk here instead of
    // re-reading all keys in lovasz_chunk_count_kernel.  Integer counts: exact and order-independent.  The fg keys of a tile
    // share their top digits (their errors are the large ones), i.e. a handful of chunks: they are counted in LDS first and the
    // block adds its non-zero counters to the global ones (one global atomic per fg key serialised on a few addresses:
    // the pass went 1.0 -> 2.0 ms).
    unsigned* cf = lds_u[1];
    const int used = (int)((n + CHUNK - 1) / CHUNK);     // scan chunks of this class in use (<= nchunks <= SEG_FG_LDS)
    if (chunk_fg) {
        for (int i = tid; i < used; i += 256) cf[i] = 0u;
        __syncthreads();
    }
    for (int j = tid; j < nk; j += 256) {
        const unsigned long long kk = sorted[j];
        const unsigned d = seg_digit(kk, shift, mask);
        const long pos = (long)gbase[d] + (unsigned)(j - (int)wbase[0][d]);
        o[pos] = kk;
        if (chunk_fg && ((kk >> PB) & 1ull)) atomicAdd(&cf[pos / CHUNK], 1u);
    }
    if (chunk_fg) {
        __syncthreads();
        for (int i = tid; i < used; i += 256) {
            const unsigned nf = cf[i];
            if (nf) atomicAdd(&chunk_fg[(long)c * nchunks + i], nf);
        }
    }
    }
}
constexpr int SEG_FG_LDS = 4 * 256;                     // chunk counters that fit the dead LDS rows; more chunks: separate count kernel

// chunk_fg[c][k] = number of fg elements among ranks [k*CHUNK, (k+1)*CHUNK) of the class's survivors (only when the scan
// chunks do not fit the last scatter pass's LDS counters)
__global__ __launch_bounds__(256) void lovasz_chunk_count_kernel(const unsigned long long* __restrict__ keys, long rows, int nchunks,
                                                                 const unsigned* __restrict__ nkept, int PB,
                                                                 unsigned* __restrict__ chunk_fg) {
    const int c = blockIdx.y, k = blockIdx.x;
    const long nv = nkept[c];
    if ((long)k * CHUNK >= nv) return;
    const unsigned long long* v = keys + (long)c * rows;
    unsigned n = 0;
    for (int j = threadIdx.x; j < CHUNK; j += 256) {
        const long i = (long)k * CHUNK + j;
        if (i < nv) n += (unsigned)((v[i] >> PB) & 1ull);
    }
    n = (unsigned)wave_sum((float)n);                    // < 2048: exact in fp32
    __shared__ unsigned sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_fg[(long)c * nchunks + k] = sm[0] + sm[1] + sm[2] + sm[3];
}

// exclusive scan of the used part of chunk_fg[c][:] in place (one block per class)
__global__ __launch_bounds__(256) void lovasz_chunk_scan_kernel(unsigned* __restrict__ chunk_fg, int nchunks, const unsigned* __restrict__ nkept) {
    const int c = blockIdx.x;
    const int used = (int)(((long)nkept[c] + CHUNK - 1) / CHUNK);
    if (used == 0) return;
    unsigned* a = chunk_fg + (long)c * nchunks;
    __shared__ unsigned sm[256];
    unsigned carry = 0;
    for (int base = 0; base < used; base += 256) {
        const int i = base + threadIdx.x;
        const unsigned x = i < used ? a[i] : 0u;
        sm[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned y = threadIdx.x >= o ? sm[threadIdx.x - o] : 0u;
            __syncthreads();
            sm[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < used) a[i] = carry + sm[threadIdx.x] - x;
        carry += sm[255];
        __syncthreads();
    }
}

// Jaccard gradient at every rank of the class's survivors, dot product with the sorted errors, scatter of d loss_c / d p into
// the pixel-major G (survivor entries only: everything else is 0 by the pruning identity and never read by the backward)
__global__ __launch_bounds__(256) void lovasz_grad_dot_kernel(const unsigned long long* __restrict__ keys,
                                                              long rows, int nchunks, const unsigned* __restrict__ counts,
                                                              const unsigned* __restrict__ nkept, int PB,
                                                              const unsigned* __restrict__ chunk_fg, float* __restrict__ G, int ldg,
                                                              double* __restrict__ part) {
    const int c = blockIdx.y;
    const long nv = nkept[c];
    const float gts = (float)counts[c];
    const unsigned long long* kk = keys + (long)c * rows;
    __shared__ unsigned sm[256];
    __shared__ float sd[4];
    for (int k = blockIdx.x; (long)k * CHUNK < nv; k += gridDim.x) {          // capped grid, see segsort_hist_kernel
    // each thread owns 8 consecutive ranks
    const long i0 = (long)k * CHUNK + threadIdx.x * 8;
    unsigned long long vv[8];
    unsigned local = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        vv[j] = (i0 + j < nv) ? kk[i0 + j] : 0ull;
        local += (unsigned)((vv[j] >> PB) & 1ull);
    }
    // block exclusive scan of `local`
    __syncthreads();
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned y = threadIdx.x >= o ? sm[threadIdx.x - o] : 0u;
        __syncthreads();
        sm[threadIdx.x] += y;
        __syncthreads();
    }
    unsigned cum = chunk_fg[(long)c * nchunks + k] + sm[threadIdx.x] - local;   // fg count strictly before rank i0
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long i = i0 + j;
        if (i < nv) {
            const unsigned long long key = vv[j];
            const unsigned fg = (unsigned)((key >> PB) & 1ull);
            const unsigned pix = (unsigned)(key & ((1ull << PB) - 1ull));
            const float e = __uint_as_float((~(unsigned)(key >> (PB + 1))) & 0x3FFFFFFFu);
            // lovasz_grad (utils/lovasz_losses.py:19-31) in the same fp32 arithmetic
            const float cum_prev = (float)cum, cum_now = (float)(cum + fg);
            const float inter = gts - cum_now, uni = gts + ((float)(i + 1) - cum_now);
            const float jac = 1.f - inter / uni;
            float grad = jac;
            if (i > 0) {
                const float inter_p = gts - cum_prev, uni_p = gts + ((float)i - cum_prev);
                grad = jac - (1.f - inter_p / uni_p);
            }
            dot += e * grad;
            // d|fg - p| / dp = -sign(fg - p);  e == 0 -> 0 (torch's abs backward uses sign)
            const float sgn = e == 0.f ? 0.f : (fg ? -1.f : 1.f);
            G[(long)pix * ldg + c] = sgn * grad;
            cum += fg;
        }
    }
    dot = wave_sum(dot);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)c * nchunks + k] = (double)sd[0] + sd[1] + sd[2] + sd[3];
    }
}

// loss_out = {mean over present classes of loss_c, n_present, survivors, n_present * n_valid (the keys of the unpruned
// formulation), xthr[C] (the keep thresholds in the logit domain)}: the thresholds travel to the backward inside the caller's tensor
constexpr int FIN_T = 1024;
__global__ __launch_bounds__(FIN_T) void lovasz_finalize_kernel(const double* __restrict__ part, int nchunks, const unsigned* __restrict__ counts,
                                                                const unsigned* __restrict__ nkept, const float* __restrict__ xthr,
                                                                int C, float* __restrict__ loss_out) {
    __shared__ double sm[FIN_T], sk[FIN_T];
    __shared__ int np[FIN_T];
    int present = 0;
    double kept = 0.0;
    // the loss is the mean over present classes of their chunk sums = ONE sum over all (present class, used chunk) entries: the
    // block strides over a class's chunks, class after class (no index division; the loads of the C classes are independent)
    for (int c = threadIdx.x; c < C; c += FIN_T) {
        present += counts[c] != 0;
        kept += (double)nkept[c];
        loss_out[4 + c] = xthr[c];
    }
    // wave w sums classes w, w + 16, ...: lane l takes chunks l, l + 64, ... of the class (fixed order -> deterministic)
    double t0 = 0.0;
    for (int c = threadIdx.x >> 6; c < C; c += FIN_T / 64) {
        const int used = (int)(((long)nkept[c] + CHUNK - 1) / CHUNK);
        const double* pc = part + (long)c * nchunks;
        for (int k = threadIdx.x & 63; k < used; k += 64) t0 += pc[k];
    }
    const double t1 = 0.0;
    sm[threadIdx.x] = t0 + t1; np[threadIdx.x] = present; sk[threadIdx.x] = kept;
    __syncthreads();
    for (int o = FIN_T / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sm[threadIdx.x] += sm[threadIdx.x + o]; np[threadIdx.x] += np[threadIdx.x + o]; sk[threadIdx.x] += sk[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] = np[0] > 0 ? (float)(sm[0] / np[0]) : 0.f;
        loss_out[1] = (float)np[0];
        loss_out[2] = (float)sk[0];
        loss_out[3] = (float)((double)np[0] * (double)counts[C]);
    }
}

constexpr int LPP = 8;
__device__ __forceinline__ float grp_sum8(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
// dz_c = g / n_present * p_c * (G_c - sum_j G_j p_j), streaming: 8 lanes share a pixel like the forward's passes.  G_c is
// fetched only where the forward's keep test holds (the same lov_err_bits against the thresholds the forward saved) — a
// sparse gather of the survivor entries; every other G_c is exactly 0.
__device__ __forceinline__ float lov_g(const float* __restrict__ grow, const float* xthr_s, float d, long t, int c) {
    return lov_keep(d, t == c, xthr_s[c]) ? grow[c] : 0.f;             // d = __fsub_rn(z, lse), the forward's test
}
template <int KQ>
__global__ __launch_bounds__(256) void lovasz_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                         long ignore, const float* __restrict__ lse,
                                                         const float* __restrict__ G, int ldg, long rows, int C,
                                                         const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                         float* __restrict__ dl, int lddl) {
    extern __shared__ float thr_s[];                     // C: xthr
    for (int i = threadIdx.x; i < C; i += 256) thr_s[i] = loss_out[4 + i];
    __syncthreads();
    const int g = threadIdx.x & (LPP - 1);
    const int c4n = (C + 3) >> 2;
    const float np = loss_out[1];
    const float gs = np > 0.f ? grad_out[0] / np : 0.f;
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += (long)gridDim.x * 32) {
        const long t = target[r];
        float* drow = dl + r * lddl;
        if (t == ignore) {                               // whole 8-lane group: an ignored pixel has no survivor, dz = 0
            for (int q = g; q < c4n; q += LPP) st4(drow + q * 4, zero4());
            continue;
        }
        const float l = lse[r];
        LovPix<false> row;
        row.p00 = logits + r * ld;
        const float* grow = G + r * ldg;
        LovRow<KQ> R;
        R.load(row, g, c4n);
        const int trips = LOV_TRIPS(KQ, g, c4n);
        // sweep 1: p = softmax probabilities (kept in the row's registers when KQ > 0: the second sweep needs no exp), s = sum_j G_j p_j
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < trips; ++k) {
            const int q = g + 8 * k, c = q * 4;
            const float4 v = R.get(row, k, q);
            // (only survivors contribute: G_j = 0 elsewhere — the exp is evaluated for them alone)
            if (c < C) { const float d = __fsub_rn(v.x, l); const float gj = lov_g(grow, thr_s, d, t, c); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 1 < C) { const float d = __fsub_rn(v.y, l); const float gj = lov_g(grow, thr_s, d, t, c + 1); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 2 < C) { const float d = __fsub_rn(v.z, l); const float gj = lov_g(grow, thr_s, d, t, c + 2); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 3 < C) { const float d = __fsub_rn(v.w, l); const float gj = lov_g(grow, thr_s, d, t, c + 3); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
        }
        s = grp_sum8(s);
        // sweep 2: dz_c = gs * p_c * (G_c - s)
#pragma unroll
        for (int k = 0; k < trips; ++k) {
            const int q = g + 8 * k, c = q * 4;
            if (q >= c4n) continue;
            const float4 v = R.get(row, k, q);
            const float4 e = make_float4(__fsub_rn(v.x, l), __fsub_rn(v.y, l), __fsub_rn(v.z, l), __fsub_rn(v.w, l));
            float4 d = zero4();
            d.x = __fmul_rn(__fmul_rn(gs, expf(e.x)), __fsub_rn(lov_g(grow, thr_s, e.x, t, c), s));
            if (c + 1 < C) d.y = __fmul_rn(__fmul_rn(gs, expf(e.y)), __fsub_rn(lov_g(grow, thr_s, e.y, t, c + 1), s));
            if (c + 2 < C) d.z = __fmul_rn(__fmul_rn(gs, expf(e.z)), __fsub_rn(lov_g(grow, thr_s, e.z, t, c + 2), s));
            if (c + 3 < C) d.w = __fmul_rn(__fmul_rn(gs, expf(e.w)), __fsub_rn(lov_g(grow, thr_s, e.w, t, c + 3), s));
            st4(drow + q * 4, d);
        }
    }
}

// ---- backward on upsampled logits (UP): the gradient never exists at full resolution.
// dot[r] = sum_j G_j p_j of output pixel r — sweep 1 of lovasz_bwd_kernel, same lane layout and summation order, the logits
// interpolated from the low-resolution tensor (L2-resident).
__global__ __launch_bounds__(256) void lovasz_up_dot_kernel(const LovSrc src, const int64_t* __restrict__ target, long ignore,
                                                            const float* __restrict__ lse, const float* __restrict__ G, int ldg, long rows,
                                                            int C, const float* __restrict__ loss_out, float* __restrict__ dot) {
    extern __shared__ float thr_s[];                     // C: xthr
    for (int i = threadIdx.x; i < C; i += 256) thr_s[i] = loss_out[4 + i];
    __syncthreads();
    const int g = threadIdx.x & (LPP - 1);
    const int c4n = (C + 3) >> 2;
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += (long)gridDim.x * 32) {
        const long t = target[r];
        if (t == ignore) { if (g == 0) dot[r] = 0.f; continue; }
        const float l = lse[r];
        LovPix<true> row;
        row.init(src, r);
        const float* grow = G + r * ldg;
        float s = 0.f;
        for (int q = g; q < c4n; q += LPP) {
            const int c = q * 4;
            const float4 v = row.get(q);
            if (c < C) { const float d = __fsub_rn(v.x, l); const float gj = lov_g(grow, thr_s, d, t, c); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 1 < C) { const float d = __fsub_rn(v.y, l); const float gj = lov_g(grow, thr_s, d, t, c + 1); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 2 < C) { const float d = __fsub_rn(v.z, l); const float gj = lov_g(grow, thr_s, d, t, c + 2); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
            if (c + 3 < C) { const float d = __fsub_rn(v.w, l); const float gj = lov_g(grow, thr_s, d, t, c + 3); if (gj != 0.f) s = fmaf(gj, expf(d), s); }
        }
        s = grp_sum8(s);
        if (g == 0) dot[r] = s;
    }
}

// pass W of the bilinear transpose with the Lovasz gradient evaluated on the way:
//   tmp[(n, oh, wl), c] = sum_ow ww(ow -> wl) * dz(n, oh, ow)[c],   dz_c = gs * p_c * (G_c - dot)
// thread = one float4 channel group of one (n, oh, wl) row (flat index: no idle lanes at C = 150), ~2 * OW / W candidate output
// columns each; the two low-resolution rows of `oh` are loaded once for the columns wl-1, wl, wl+1 and every candidate's logits
// are interpolated from them with lov_lerp in the forward's operation order (same bits -> same keep decisions).  The summation
// order over ow and the fused multiply-add are those of bilinear_bwd_axis_kernel<0>, so the result equals the unfused path's.
__global__ __launch_bounds__(256) void lovasz_up_bwd_w_kernel(const LovSrc src, int N, const int64_t* __restrict__ target, long ignore,
                                                              const float* __restrict__ lse, const float* __restrict__ dot,
                                                              const float* __restrict__ G, int ldg, int C,
                                                              const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                              float* __restrict__ tmp, int ldt) {
    extern __shared__ float thr_s[];                     // C: xthr
    for (int i = threadIdx.x; i < C; i += 256) thr_s[i] = loss_out[4 + i];
    __syncthreads();
    const int c4n = (C + 3) >> 2;
    const int H = src.H, W = src.W, OH = src.OH, OW = src.OW, ac = src.ac;
    const long total = (long)N * OH * W * c4n;
    const float np = loss_out[1];
    const float gs = np > 0.f ? grad_out[0] / np : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned r = (unsigned)(i / c4n);                      // (n * OH + oh) * W + wl  < 2^24
        const int c4 = (int)(i - (long)r * c4n), c = c4 * 4;
        const unsigned tq = r / (unsigned)W;
        const int wl = (int)(r - tq * (unsigned)W);
        const unsigned n = tq / (unsigned)OH;
        const int oh = (int)(tq - n * (unsigned)OH);
        const Lerp a = bl_src(oh, src.sh, H, ac);
        int lo_c, hi_c;
        bl_range(wl, src.sw, W, OW, ac, lo_c, hi_c);
        const float* base = src.base + (long)n * H * W * src.ld + c;
        const int wm = max(wl - 1, 0), wp = min(wl + 1, W - 1);
        const int cols[3] = {wm, wl, wp};
        float4 T0[3], T1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T0[j] = ld4(base + ((long)a.i0 * W + cols[j]) * src.ld);
            T1[j] = ld4(base + ((long)a.i1 * W + cols[j]) * src.ld);
        }
        float4 acc = zero4();
        for (int ow = lo_c; ow <= hi_c; ++ow) {
            const Lerp b = bl_src(ow, src.sw, W, ac);
            const float wt = (b.i0 == wl ? b.l0 : 0.f) + (b.i1 == wl ? b.l1 : 0.f);
            if (wt == 0.f) continue;
            const long hp = (long)tq * OW + ow;
            const long t = target[hp];
            if (t == ignore) continue;                   // dz = 0
            const int j0 = b.i0 == wl ? 1 : (b.i0 < wl ? 0 : 2), j1 = b.i1 == wl ? 1 : (b.i1 < wl ? 0 : 2);
            const float4 v00 = j0 == 1 ? T0[1] : (j0 == 0 ? T0[0] : T0[2]), v01 = j1 == 1 ? T0[1] : (j1 == 0 ? T0[0] : T0[2]);
            const float4 v10 = j0 == 1 ? T1[1] : (j0 == 0 ? T1[0] : T1[2]), v11 = j1 == 1 ? T1[1] : (j1 == 0 ? T1[0] : T1[2]);
            const float l = lse[hp], d = dot[hp];
            const float* grow = G + hp * ldg;
            float zz, p, dz;
#define LOV_UPW(comp, cc)                                                                                                         \
            zz = lov_lerp(a.l0, lov_lerp(b.l0, v00.comp, b.l1, v01.comp), a.l1, lov_lerp(b.l0, v10.comp, b.l1, v11.comp));             \
            p = __fsub_rn(zz, l);                                                                                                 \
            dz = __fmul_rn(__fmul_rn(gs, expf(p)), __fsub_rn(lov_g(grow, thr_s, p, t, cc), d));                                   \
            acc.comp = fmaf(wt, dz, acc.comp);
            LOV_UPW(x, c)
            if (c + 1 < C) { LOV_UPW(y, c + 1) }
            if (c + 2 < C) { LOV_UPW(z, c + 2) }
            if (c + 3 < C) { LOV_UPW(w, c + 3) }
#undef LOV_UPW
        }
        st4(tmp + (long)r * ldt + c, acc);
    }
}

// KQ of a class count: float4 groups per lane when the row fits 8 per lane (C <= 256), else 0 (groups re-read per sweep)
int lov_kq(int C) {
    const int c4n = (C + 3) >> 2;
    return c4n <= 64 ? (c4n + 7) / 8 : 0;
}
#define LOV_DISPATCH(KQV, CALL)                                                                                       \
    switch (KQV) {                                                                                                    \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;               \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;               \
        default: CALL(0); break;                                                                                      \
    }

struct LovaszLayout {
    size_t keys_a, keys_b, chunk_fg, counts, thr, xthr, nkept, part, cnt, temp, total;
    int nchunks, PB, ntiles;
    long nunits;
};
// SEGMI_LOVASZ_PRUNE=0 / segmi_lovasz_set_prune(0): keep every valid pixel of every present class (the full sort) — A/B and tests
int g_lovasz_prune = -1;
bool lovasz_prune() {
    if (g_lovasz_prune < 0) {
        const char* e = getenv("SEGMI_LOVASZ_PRUNE");
        g_lovasz_prune = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    return g_lovasz_prune == 1;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// which of count (1) / emit (2) / backward (4) hold the pixel's row in registers, all loads issued up front (SEGMI_LOVASZ_ROWREGS).
// Re-measured in round 6 with the exp gone from the keep test (profiles/r06_lovasz_fused_upsample.txt, loss forward / backward at the
// cfg5 shard): count 1.566 -> 1.521 ms forward (86 VGPRs) — on by default; emit 1.566 -> 1.781 (208 VGPRs, the unrolled ballot groups);
// backward 2.675 -> 2.900 (135 VGPRs: the unrolled exp chains) — both stay on the re-reading form.
int lovasz_rowregs() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SEGMI_LOVASZ_ROWREGS"); v = e ? atoi(e) : 1; }
    return v;
}

bool lovasz_layout(long rows, int C, LovaszLayout* L) {
    if (rows <= 0 || C <= 0 || C > 1820 || rows >= (1L << 24)) return false;   // fp32-exact cumsums like the reference
    const size_t n = (size_t)rows * C;
    L->nchunks = (int)((rows + CHUNK - 1) / CHUNK);
    int cb = 1;
    while ((1 << cb) < C) ++cb;
    int pb = 1;
    while ((1L << pb) < rows) ++pb;
    if (cb + pb + 32 > 64) return false;                  // class | invalid | 30-bit error | fg | pixel must fit one 64-bit key
    L->PB = pb;
    L->nunits = (rows + UPX - 1) / UPX;
    size_t off = 0;
    // the key buffers are sized for the worst case (every element survives); only the survivors' part of each class segment is touched
    L->keys_a = off; off += align256(n * 8);
    L->keys_b = off; off += align256(n * 8);
    L->chunk_fg = off; off += align256((size_t)C * L->nchunks * 4);
    L->counts = off; off += align256((size_t)(C + 1) * 4);
    L->thr = off; off += align256((size_t)C * 4);
    L->xthr = off; off += align256((size_t)C * 4);
    L->nkept = off; off += align256((size_t)C * 4);
    L->part = off; off += align256((size_t)C * L->nchunks * 8);
    L->cnt = off; off += align256((size_t)C * L->nunits * 4);
    L->ntiles = (int)((rows + ST - 1) / ST);
    L->temp = off; off += align256((size_t)C * L->ntiles * 256 * sizeof(unsigned));   // per-tile digit histograms [C][ntiles][256]
    L->total = off;
    return true;
}

}  // namespace

extern "C" {

int segmi_lovasz_set_prune(int on) {
    if (on != 0 && on != 1) return SEGMI_ERR_BADARG;
    g_lovasz_prune = on;
    return SEGMI_OK;
}

size_t segmi_lovasz_workspace(long rows, int C) {
    LovaszLayout L;
    return lovasz_layout(rows, C, &L) ? L.total : 0;
}

}  // extern "C"

namespace {
// the forward on either source of logits (LovSrc / `up`): segmi_lovasz_fwd and segmi_upsample_lovasz_fwd
int lovasz_fwd_impl(const LovSrc lsrc, bool up, const int64_t* target, long rows, int C, long ignore_index, float* lse,
                    float* G, int ldg, float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!lsrc.base || !target || !lse || !G || !loss_out || C <= 0) return SEGMI_ERR_BADARG;
    LovaszLayout L;
    if (!lovasz_layout(rows, C, &L)) return SEGMI_ERR_BADARG;
    const int ld = lsrc.ld;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (ldg & 3) || ldg < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned long long* ka = (unsigned long long*)(ws + L.keys_a);
    unsigned long long* kb = (unsigned long long*)(ws + L.keys_b);
    unsigned* chunk_fg = (unsigned*)(ws + L.chunk_fg);
    unsigned* counts = (unsigned*)(ws + L.counts);
    unsigned* thr = (unsigned*)(ws + L.thr);
    float* xthr = (float*)(ws + L.xthr);
    unsigned* nkept = (unsigned*)(ws + L.nkept);
    unsigned* cnt = (unsigned*)(ws + L.cnt);
    double* part = (double*)(ws + L.part);
    const int prune = lovasz_prune() ? 1 : 0;

    hipMemsetAsync(counts, 0, (size_t)(C + 1) * 4, st);
    hipMemsetAsync(thr, 0xFF, (size_t)C * 4, st);
    long pb = (rows + 31) / 32;
    if (pb > SEGMI_MAX_GRID) pb = SEGMI_MAX_GRID;
    const int kq = lov_kq(C);
    const unsigned ublocks = (unsigned)((L.nunits + 3) / 4);
#define LOV_PREPARE_(K, U) hipLaunchKernelGGL((lovasz_prepare_kernel<K, U>), dim3((unsigned)pb), dim3(256), (size_t)(2 * C + 1) * 4, st, lsrc, target, rows, C, \
                                              ignore_index, lse, counts, thr)
#define LOV_PREPARE(K) do { if (up) LOV_PREPARE_(K, true); else LOV_PREPARE_(K, false); } while (0)
#define LOV_COUNT_(K, U) hipLaunchKernelGGL((lovasz_keep_count_kernel<K, U>), dim3(ublocks), dim3(256), (size_t)(5 * C) * 4, st, lsrc, target, (const float*)lse, \
                                            rows, C, ignore_index, (const unsigned*)thr, (const unsigned*)counts, prune, L.nunits, cnt, xthr)
#define LOV_COUNT(K) do { if (up) LOV_COUNT_(K, true); else LOV_COUNT_(K, false); } while (0)
#define LOV_EMIT_(K, U) hipLaunchKernelGGL((lovasz_emit_kernel<K, U>), dim3(ublocks), dim3(256), (size_t)(5 * C) * 4, st, lsrc, target, (const float*)lse, rows, C, \
                                           ignore_index, L.PB, (const float*)xthr, L.nunits, (const unsigned*)cnt, ka)
#define LOV_EMIT(K) do { if (up) LOV_EMIT_(K, true); else LOV_EMIT_(K, false); } while (0)
    // measured at C = 150 (profiles/r05_lovasz_alone_kernel_stats*.csv): prepare<5> 267 us against 384 us for the re-reading form;
    // count<5> 255 against 258 (no gain in round 5; -45 us in round 6 without the exp: lovasz_rowregs); emit<5> 667 against 410 (the
    // unrolled ballot groups need 212 VGPRs: 2 waves per SIMD) — emit stays on the re-reading form
    const int rowregs = lovasz_rowregs();
    LOV_DISPATCH(kq, LOV_PREPARE)
    if (rowregs & 1) { LOV_DISPATCH(kq, LOV_COUNT) } else LOV_COUNT(0);
    hipLaunchKernelGGL(lovasz_keep_scan_kernel, dim3((unsigned)C), dim3(KS_T), 0, st, cnt, L.nunits, nkept);
    if (rowregs & 2) { LOV_DISPATCH(kq, LOV_EMIT) } else LOV_EMIT(0);
#undef LOV_PREPARE
#undef LOV_COUNT
#undef LOV_EMIT
#undef LOV_PREPARE_
#undef LOV_COUNT_
#undef LOV_EMIT_
    const bool fused_fg = L.nchunks <= SEG_FG_LDS;         // chunk fg counts come out of the last scatter pass
    // four stable 8-bit passes over the 31-bit field [invalid | ~error] above the fg bit: ka -> kb -> ka -> kb -> ka
    unsigned* hist = (unsigned*)(ws + L.temp);
    unsigned long long *src = ka, *dst = kb;
    const dim3 tgrid((unsigned)(L.ntiles < SEG_GX ? L.ntiles : SEG_GX), (unsigned)C);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = L.PB + 1 + 8 * pass;
        const unsigned mask = pass < 3 ? 0xFFu : 0x7Fu;
        hipLaunchKernelGGL(segsort_hist_kernel, tgrid, dim3(256), 0, st, (const unsigned long long*)src, rows, L.ntiles, shift, mask, (const unsigned*)nkept, hist);
        hipLaunchKernelGGL(segsort_scan_kernel, dim3((unsigned)C), dim3(256, SCAN_Q), 0, st, hist, L.ntiles, (const unsigned*)nkept);
        const bool count_fg = pass == 3 && fused_fg;
        if (count_fg) hipMemsetAsync(chunk_fg, 0, (size_t)C * L.nchunks * sizeof(unsigned), st);
        hipLaunchKernelGGL(segsort_scatter_kernel, tgrid, dim3(256), 0, st, (const unsigned long long*)src, dst, rows, L.ntiles, shift, mask,
                           (const unsigned*)nkept, (const unsigned*)hist, count_fg ? chunk_fg : (unsigned*)nullptr, L.nchunks, L.PB);
        unsigned long long* t = src; src = dst; dst = t;
    }
    const unsigned long long* ks = src;                    // == ka after an even number of passes
    const dim3 grid((unsigned)L.nchunks, (unsigned)C), jgrid((unsigned)(L.nchunks < 2 * SEG_GX ? L.nchunks : 2 * SEG_GX), (unsigned)C);
    if (!fused_fg)
        hipLaunchKernelGGL(lovasz_chunk_count_kernel, grid, dim3(256), 0, st, ks, rows, L.nchunks, (const unsigned*)nkept, L.PB, chunk_fg);
    hipLaunchKernelGGL(lovasz_chunk_scan_kernel, dim3((unsigned)C), dim3(256), 0, st, chunk_fg, L.nchunks, (const unsigned*)nkept);
    hipLaunchKernelGGL(lovasz_grad_dot_kernel, jgrid, dim3(256), 0, st, ks, rows, L.nchunks, (const unsigned*)counts, (const unsigned*)nkept, L.PB,
                       (const unsigned*)chunk_fg, G, ldg, part);
    hipLaunchKernelGGL(lovasz_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, (const double*)part, L.nchunks, (const unsigned*)counts,
                       (const unsigned*)nkept, (const float*)xthr, C, loss_out);
    return segmi_launch_status();
}

LovSrc lov_up_src(const float* lo, int ld, int H, int W, int OH, int OW, int ac) {
    LovSrc s;
    s.base = lo; s.ld = ld; s.H = H; s.W = W; s.OH = OH; s.OW = OW; s.ac = ac ? 1 : 0;
    // the host evaluates bl_scale's expression in the same fp32 arithmetic as the device does in bilinear_fwd_kernel
    s.sh = ac ? (OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f) : (float)H / (float)OH;
    s.sw = ac ? (OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f) : (float)W / (float)OW;
    return s;
}
}  // namespace

extern "C" {

int segmi_lovasz_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float* lse,
                     float* G, int ldg, float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    LovSrc s;
    memset(&s, 0, sizeof(s));
    s.base = logits; s.ld = ld;
    return lovasz_fwd_impl(s, false, target, rows, C, ignore_index, lse, G, ldg, loss_out, workspace, workspace_bytes, stream);
}

// Lovasz-Softmax on bilinearly upsampled logits without materialising them (see LovSrc).  logits_lo [N, H, W, C] (row stride ld),
// target / lse [N, OH, OW], G [N*OH*OW, ldg] (survivor entries only, like segmi_lovasz_fwd).  Workspace: the forward needs
// segmi_lovasz_workspace(N*OH*OW, C); the backward the pass-W buffer [N, OH, W, C4] + one float per output pixel —
// segmi_upsample_lovasz_workspace covers both.
size_t segmi_upsample_lovasz_workspace(int N, int H, int W, int C, int OH, int OW) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return 0;
    const size_t a = segmi_lovasz_workspace((long)N * OH * OW, C);
    if (!a) return 0;
    const size_t b = align256((size_t)N * OH * W * ((C + 3) & ~3) * sizeof(float)) + align256((size_t)N * OH * OW * sizeof(float));
    return a > b ? a : b;
}

int segmi_upsample_lovasz_fwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                              const int64_t* target, long ignore_index, float* lse, float* G, int ldg, float* loss_out,
                              void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return SEGMI_ERR_BADARG;
    return lovasz_fwd_impl(lov_up_src(logits_lo, ld, H, W, OH, OW, align_corners), true, target, (long)N * OH * OW, C, ignore_index, lse, G,
                           ldg, loss_out, workspace, workspace_bytes, stream);
}

int segmi_upsample_lovasz_bwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                              const int64_t* target, long ignore_index, const float* lse, const float* G, int ldg,
                              const float* loss_out, const float* grad_out, float* dlogits_lo, int lddl, void* workspace,
                              size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits_lo || !target || !lse || !G || !loss_out || !grad_out || !dlogits_lo || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 1820 ||
        OH <= 0 || OW <= 0)
        return SEGMI_ERR_BADARG;
    const long rows = (long)N * OH * OW;
    if (rows >= (1L << 24) || (long)N * OH * W >= (1L << 24)) return SEGMI_ERR_BADARG;
    const int ldt = (C + 3) & ~3;
    if ((ld & 3) || ld < ldt || (ldg & 3) || ldg < ldt || (lddl & 3) || lddl < ldt) return SEGMI_ERR_ALIGN;
    const size_t tmp_bytes = align256((size_t)N * OH * W * ldt * sizeof(float));
    if (!workspace || workspace_bytes < tmp_bytes + align256((size_t)rows * sizeof(float)) || ((uintptr_t)workspace & 255)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* tmp = (float*)workspace;
    float* dot = (float*)((char*)workspace + tmp_bytes);
    const LovSrc src = lov_up_src(logits_lo, ld, H, W, OH, OW, align_corners);
    long b = (rows + 31) / 32;
    if (b > 4 * SEGMI_MAX_GRID) b = 4 * SEGMI_MAX_GRID;
    hipLaunchKernelGGL(lovasz_up_dot_kernel, dim3((unsigned)b), dim3(256), (size_t)C * 4, st, src, target, ignore_index, lse, G, ldg, rows, C, loss_out, dot);
    const long total = (long)N * OH * W * (ldt / 4);
    long wb = (total + 255) / 256;
    if (wb > 4 * SEGMI_MAX_GRID) wb = 4 * SEGMI_MAX_GRID;
    hipLaunchKernelGGL(lovasz_up_bwd_w_kernel, dim3((unsigned)wb), dim3(256), (size_t)C * 4, st, src, N, target, ignore_index, lse, (const float*)dot, G, ldg,
                       C, loss_out, grad_out, tmp, ldt);
    return segmi_internal_bilinear_bwd_height(tmp, ldt, dlogits_lo, lddl, N, H, W, C, OH, OW, src.ac, st);
}

int segmi_lovasz_bwd(const float* logits, int ld, const int64_t* target, long ignore_index, const float* lse, const float* G, int ldg,
                     long rows, int C, const float* loss_out, const float* grad_out, float* dlogits, int lddl, segmi_stream_t stream) {
    if (!logits || !target || !lse || !G || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0 || C > 1820) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (ldg & 3) || ldg < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    long b = (rows + 31) / 32;
    if (b > 4 * SEGMI_MAX_GRID) b = 4 * SEGMI_MAX_GRID;
    // (The kernel sweeps a pixel's row twice and the second sweep misses the L2 — PMC: 2.5x the logits fetched,
    //  profiles/r05_lovasz_traffic.txt — but capping the residency with dummy LDS so that it hits made the kernel SLOWER: 806 us at 8
    //  workgroups per CU, 852 / 1031 / 1155 / 1864 us with 20 / 30 / 40 / 60 KB of padding: it is latency-bound, not bandwidth-bound.)
#define LOV_BWD(K) hipLaunchKernelGGL(lovasz_bwd_kernel<K>, dim3((unsigned)b), dim3(256), (size_t)C * 4, (hipStream_t)stream, logits, ld, target, ignore_index, \
                                      lse, G, ldg, rows, C, loss_out, grad_out, dlogits, lddl)
    // the re-reading form: with the row (and its probabilities) held in registers the kernel needs 131 VGPRs at C = 150 and ran
    // 1018 us against 794 us (profiles/r05_lovasz_alone_kernel_stats_rowregs{1,0}.csv)
    if (lovasz_rowregs() & 4) { LOV_DISPATCH(lov_kq(C), LOV_BWD) } else LOV_BWD(0);
#undef LOV_BWD
    return segmi_launch_status();
}

}  // extern "C"
