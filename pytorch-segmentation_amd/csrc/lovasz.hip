// Lovasz-Softmax loss (classes='present', per_image=False), forward and backward.
// Replaces utils/losses.py:79-89 -> utils/lovasz_losses.py:153-218 (flatten_probas, lovasz_softmax_flat) and
// lovasz_grad :19-31 of the reference, which run softmax + C x (torch.sort over all valid pixels, 2 cumsums, dot)
// and whose backward scatters through C `probas[:, c]` selects (O(C^2 * P) traffic on the reference).
//
// MI355X formulation — all HBM-bound streaming, no host synchronisation:
//   1. lovasz_prepare   per pixel: log-sum-exp; histogram of labels (class presence, |fg_c|), number of valid pixels
//   2. lovasz_emit      one 64-bit word per (class, pixel):  [class | invalid | ~bits30(|fg - p_c|) | fg | pixel index]
//                       (errors lie in [0, 2): their float bits fit 30 bits).  Ascending order of the upper field == classes
//                       ascending, errors DESCENDING, ignored pixels last inside their class; the low PB + 1 bits are payload.
//   3. a SEGMENTED least-significant-digit radix sort, hand-written (segsort_* below): the class is implicit in the segment
//      (keys are emitted class-major), so only the 31-bit field [invalid | ~error] is sorted — four stable 8-bit passes
//      (histogram per 4096-key tile -> per-class scan -> scatter through an LDS-sorted tile so that every digit's run leaves as
//      one contiguous write), payload (fg bit, pixel index) riding inside the 64-bit key; absent classes are skipped on the
//      device.  The per-class torch.sort calls of the reference become 12 bandwidth-bound launches.
//      (SEGMI_LOVASZ_SORT=rocprim selects the former rocprim::radix_sort_keys over the 31 + log2(C) upper bits for A/B.)
//   4. lovasz_chunk_count / lovasz_chunk_scan / lovasz_grad_dot: two-level scan of the sorted fg bits -> Jaccard index
//      at every rank in the same float32 arithmetic as lovasz_grad (integers are exact in fp32 below 2^24 pixels),
//      first difference, dot with the sorted errors, and scatter of d loss / d p into the class-major G[class][pixel].
//   5. lovasz_finalize  loss = mean over present classes.
//   backward: dz_c = g * p_c * (G_c - sum_j G_j p_j) / n_present   (softmax Jacobian; G = 0 for absent classes / ignored pixels)
//
// Ties: elements of one class with bit-equal errors may be ranked in any order (torch.sort is unstable too); the loss value
// does not depend on that order, the per-pixel gradients inside a tie group do.
#include "segmi_common.h"
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int CHUNK = 2048;   // ranks per scan block (256 threads x 8)

// 8 lanes share a pixel (float4 channel groups, xor-shuffle reductions): coalesced 128-byte row segments
__global__ __launch_bounds__(256) void lovasz_prepare_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                             long rows, int C, long ignore, float* __restrict__ lse,
                                                             unsigned* __restrict__ counts /* [C] fg counts, [C] n_valid */) {
    extern __shared__ unsigned hist[];   // C + 1
    for (int i = threadIdx.x; i <= C; i += 256) hist[i] = 0;
    __syncthreads();
    const int g = threadIdx.x & 7;
    const int c4n = (C + 3) >> 2;
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += (long)gridDim.x * 32) {
        const float* row = logits + r * ld;
        float m = -INFINITY;
        for (int q = g; q < c4n; q += 8) {
            const float4 v = ld4(row + q * 4);
            const int c = q * 4;
            m = fmaxf(m, v.x);
            if (c + 1 < C) m = fmaxf(m, v.y);
            if (c + 2 < C) m = fmaxf(m, v.z);
            if (c + 3 < C) m = fmaxf(m, v.w);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
        for (int q = g; q < c4n; q += 8) {
            const float4 v = ld4(row + q * 4);
            const int c = q * 4;
            s += expf(v.x - m);
            if (c + 1 < C) s += expf(v.y - m);
            if (c + 2 < C) s += expf(v.z - m);
            if (c + 3 < C) s += expf(v.w - m);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (g == 0) {
            lse[r] = m + logf(s);
            const long t = target[r];
            if (t != ignore) {
                atomicAdd(&hist[C], 1u);
                if (t >= 0 && t < C) atomicAdd(&hist[(int)t], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= C; i += 256)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// Block = 256 pixels.  The logits are pixel-major (a pixel's C classes are contiguous) and the keys class-major (a class's
// pixels are contiguous): 32 classes at a time go through an LDS tile — read as 128-byte row segments (8 lanes x 16 B per
// pixel), written as 2 KB runs of one class.  (Thread-per-pixel reads straight from HBM were 600-byte-strided dwords: 3.5 ms
// for cfg5's 1.26 GB of logits, 4x the stream time.)
__global__ __launch_bounds__(256) void lovasz_emit_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                          const float* __restrict__ lse, long rows, int C, long ignore, int PB,
                                                          unsigned long long* __restrict__ keys) {
    __shared__ float tile[256][33];
    const long r0 = (long)blockIdx.x * 256;
    const long r = r0 + threadIdx.x;
    const bool rok = r < rows;
    const long t = rok ? target[r] : ignore;
    const bool valid = rok && t != ignore;
    const float l = rok ? lse[r] : 0.f;
    const int lr = threadIdx.x >> 3, lq = (threadIdx.x & 7) * 4;     // load role: pixel lr (+32 per pass), classes lq..lq+3 of the group
    for (int cb = 0; cb < C; cb += 32) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int pl = pass * 32 + lr;
            const long rr = r0 + pl;
            float4 v = zero4();
            if (rr < rows && cb + lq < ld) v = ld4(logits + rr * ld + cb + lq);      // ld is a multiple of 4: whole float4 or none
            tile[pl][lq] = v.x; tile[pl][lq + 1] = v.y; tile[pl][lq + 2] = v.z; tile[pl][lq + 3] = v.w;
        }
        __syncthreads();
        const int cn = min(32, C - cb);
        if (rok)
            for (int j = 0; j < cn; ++j) {
                const int c = cb + j;
                const float p = expf(tile[threadIdx.x][j] - l);
                const bool fg = valid && t == c;
                const float e = fabsf((fg ? 1.f : 0.f) - p);
                const unsigned long long inv30 = (unsigned long long)((~__float_as_uint(e)) & 0x3FFFFFFFu);
                keys[(long)c * rows + r] = ((((unsigned long long)c << 1 | (valid ? 0ull : 1ull)) << 30 | inv30) << 1 | (fg ? 1ull : 0ull)) << PB |
                                           (unsigned long long)r;
            }
        __syncthreads();
    }
}

// ---- segmented LSD radix sort of the class-major key array: keys[c * rows + i], i < rows, sorted per class by the digit field
constexpr int ST = 4096;      // keys per sort tile: 256 threads x 16 rounds (8192-key tiles, 73 KB of LDS, two blocks per CU: 3 % slower)
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

// hist[(c * ntiles + tile) * 256 + d] = number of keys of tile `tile` of class c whose digit is d.
// A histogram does not care about order: one LDS atomic per key into a per-WAVE histogram (four 1 KB rows, summed at the end)
// instead of the ballot ranking the stable scatter needs — the ballots (9 per 64 keys) made this pass as compute-bound as it is
// memory-bound (0.75 ms per pass for cfg5's 2.5 GB of keys = 3.3 TB/s); same-address atomics of a digit most keys share (the
// exponent pass) serialise inside one ds instruction, which is still ~4x cheaper than the ballots.
__global__ __launch_bounds__(256) void segsort_hist_kernel(const unsigned long long* __restrict__ keys, long rows, int ntiles, int shift,
                                                           unsigned mask, const unsigned* __restrict__ counts, unsigned* __restrict__ hist) {
    const int c = blockIdx.y, tile = blockIdx.x;
    if (counts[c] == 0) return;                          // absent class: never read downstream (classes='present')
    __shared__ unsigned h[4][256];
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0; h[2][threadIdx.x] = 0; h[3][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long* k = keys + (long)c * rows;
    const long i0 = (long)tile * ST;
    unsigned* hw = h[threadIdx.x >> 6];
#pragma unroll 1
    for (int r0 = 0; r0 < ST / 256; r0 += SEG_MLP) {     // SEG_MLP 512-byte loads per wave in flight (one at a time is latency-bound: 1 TB/s)
        unsigned long long kq[SEG_MLP];
        bool vq[SEG_MLP];
#pragma unroll
        for (int u = 0; u < SEG_MLP; ++u) {
            const long i = i0 + (r0 + u) * 256 + threadIdx.x;
            vq[u] = i < rows;
            kq[u] = vq[u] ? k[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < SEG_MLP; ++u)
            if (vq[u]) atomicAdd(&hw[seg_digit(kq[u], shift, mask)], 1u);
    }
    __syncthreads();
    hist[((long)c * ntiles + tile) * 256 + threadIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

// in place: hist[c][tile][d] -> first output rank (inside the class segment) of that tile's keys with digit d:
// exclusive scan in (digit, tile) order.  One block per class, 256 digits x SCAN_Q tile ranges: thread (d, q) walks its quarter
// of digit d's column (coalesced across d), the quarters and the digits are combined through LDS.  (One thread per digit walking
// all 512 tiles twice: 183 us per pass on 150 of the chip's 256 CUs.)
constexpr int SCAN_Q = 4;
__global__ __launch_bounds__(256 * SCAN_Q) void segsort_scan_kernel(unsigned* __restrict__ hist, int ntiles, const unsigned* __restrict__ counts) {
    const int c = blockIdx.x, d = threadIdx.x, q = threadIdx.y;
    if (counts[c] == 0) return;
    unsigned* col = hist + (long)c * ntiles * 256 + d;
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
                                                              const unsigned* __restrict__ counts, const unsigned* __restrict__ hist,
                                                              unsigned* __restrict__ chunk_fg, int nchunks, int PB) {
    const int c = blockIdx.y, tile = blockIdx.x;
    if (counts[c] == 0) return;
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
    const long i0 = (long)tile * ST;
    const int nk = (int)(rows - i0 < (long)ST ? rows - i0 : (long)ST);
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
    if (chunk_fg) {
        for (int i = tid; i < nchunks; i += 256) cf[i] = 0u;
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
        for (int i = tid; i < nchunks; i += 256) {
            const unsigned n = cf[i];
            if (n) atomicAdd(&chunk_fg[(long)c * nchunks + i], n);
        }
    }
}
constexpr int SEG_FG_LDS = 4 * 256;                     // chunk counters that fit the dead LDS rows; more chunks: separate count kernel

// chunk_fg[c][k] = number of fg elements among ranks [k*CHUNK, (k+1)*CHUNK) (ranks < n_valid only)
__global__ __launch_bounds__(256) void lovasz_chunk_count_kernel(const unsigned long long* __restrict__ keys, long rows, int nchunks,
                                                                 const unsigned* __restrict__ counts, int C, int PB,
                                                                 unsigned* __restrict__ chunk_fg) {
    const int c = blockIdx.y, k = blockIdx.x;
    if (counts[c] == 0) return;                          // absent class: skipped by classes='present'
    const long nv = counts[C];
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

// exclusive scan of chunk_fg[c][:] in place (one block per class)
__global__ __launch_bounds__(256) void lovasz_chunk_scan_kernel(unsigned* __restrict__ chunk_fg, int nchunks, const unsigned* __restrict__ counts) {
    const int c = blockIdx.x;
    if (counts[c] == 0) return;
    unsigned* a = chunk_fg + (long)c * nchunks;
    __shared__ unsigned sm[256];
    unsigned carry = 0;
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const unsigned x = i < nchunks ? a[i] : 0u;
        sm[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned y = threadIdx.x >= o ? sm[threadIdx.x - o] : 0u;
            __syncthreads();
            sm[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < nchunks) a[i] = carry + sm[threadIdx.x] - x;
        carry += sm[255];
        __syncthreads();
    }
}

// Jaccard gradient at every rank, dot product with the sorted errors, scatter of d loss_c / d p into G
__global__ __launch_bounds__(256) void lovasz_grad_dot_kernel(const unsigned long long* __restrict__ keys,
                                                              long rows, int nchunks, const unsigned* __restrict__ counts, int C, int PB,
                                                              const unsigned* __restrict__ chunk_fg, float* __restrict__ G,
                                                              double* __restrict__ part, unsigned pix_lo, unsigned pix_hi) {
    // 1-D grid; block b runs on XCD b % 8 and takes class (b/8 / nchunks)*8 + b%8: all chunks of a class scatter into that
    // class's plane of G from ONE XCD, whose L2 (with the Infinity Cache behind it) merges the 4-byte writes into full lines.
    // (A pixel-major G[pixel][class] received its 32 dwords per line from 32 classes at 32 different times: 4.9 ms for cfg5.)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int c = (jb / nchunks) * 8 + xcd, k = jb % nchunks;
    if (c >= C || counts[c] == 0) return;
    const long nv = counts[C];
    const float gts = (float)counts[c];
    const unsigned long long* kk = keys + (long)c * rows;
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
    __shared__ unsigned sm[256];
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
            const bool invalid = (key >> (PB + 31)) & 1ull;       // cannot occur below n_valid (ignored pixels sort last); kept as a guard
            const float e = invalid ? 0.f : __uint_as_float((~(unsigned)(key >> (PB + 1))) & 0x3FFFFFFFu);
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
            const float sgn = invalid || e == 0.f ? 0.f : (fg ? -1.f : 1.f);
            if (pix >= pix_lo && pix < pix_hi) G[(long)c * rows + pix] = sgn * grad;       // this launch's pixel window (see the caller)
            cum += fg;
        }
    }
    dot = wave_sum(dot);
    __shared__ float sd[4];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0 && pix_lo == 0) part[(long)c * nchunks + k] = (double)sd[0] + sd[1] + sd[2] + sd[3];
}

// loss_out = {mean over present classes of loss_c, n_present}
constexpr int FIN_T = 1024;
__global__ __launch_bounds__(FIN_T) void lovasz_finalize_kernel(const double* __restrict__ part, int nchunks, const unsigned* __restrict__ counts,
                                                                int C, long rows, float* __restrict__ loss_out) {
    __shared__ double sm[FIN_T];
    __shared__ int np[FIN_T];
    double total = 0.0;
    int present = 0;
    const long nv = counts[C];
    const int used = (int)((nv + CHUNK - 1) / CHUNK);
    // the loss is the mean over present classes of their chunk sums = ONE sum over all (present class, chunk) entries: the block
    // strides over a class's chunks, class after class (no index division; the loads of the C classes are independent)
    for (int c = threadIdx.x; c < C; c += FIN_T) present += counts[c] != 0;
    double t0 = 0.0, t1 = 0.0;
    for (int c = 0; c < C; ++c) {
        if (counts[c] == 0) continue;
        const double* pc = part + (long)c * nchunks;
        int k = threadIdx.x;
        for (; k + FIN_T < used; k += 2 * FIN_T) { t0 += pc[k]; t1 += pc[k + FIN_T]; }
        if (k < used) t0 += pc[k];
    }
    total = t0 + t1;
    sm[threadIdx.x] = total; np[threadIdx.x] = present;
    __syncthreads();
    for (int o = FIN_T / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sm[threadIdx.x] += sm[threadIdx.x + o]; np[threadIdx.x] += np[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] = np[0] > 0 ? (float)(sm[0] / np[0]) : 0.f;
        loss_out[1] = (float)np[0];
    }
}

constexpr int LPP = 8;
__device__ __forceinline__ float grp_sum8(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
// dz_c = g / n_present * p_c * (G_c - sum_j G_j p_j).  G is class-major (G[c][pixel]); a block stages the [C][TP] slab of
// its TP pixels in LDS (coalesced TP*4-byte runs per class), then 8 lanes share a pixel like the forward's prepare kernel.
__global__ __launch_bounds__(256) void lovasz_bwd_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ lse,
                                                         const float* __restrict__ G, long rows, int C, int TP,
                                                         const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                         float* __restrict__ dl, int lddl) {
    extern __shared__ float gs_tile[];                   // [C][TP + 1]
    const int pitch = TP + 1;
    const int g = threadIdx.x & (LPP - 1);
    const int c4n = (C + 3) >> 2;
    const float np = loss_out[1];
    const float gs = np > 0.f ? grad_out[0] / np : 0.f;
    const long ntiles = (rows + TP - 1) / TP;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * TP;
        __syncthreads();                                 // previous tile's readers are done
        for (int idx = threadIdx.x; idx < C * TP; idx += 256) {
            const int c = idx / TP, pl = idx - c * TP;
            gs_tile[c * pitch + pl] = r0 + pl < rows ? G[(long)c * rows + r0 + pl] : 0.f;
        }
        __syncthreads();
        for (int pl = threadIdx.x / LPP; pl < TP; pl += 256 / LPP) {
            const long r = r0 + pl;
            if (r >= rows) continue;                     // whole 8-lane group leaves together: the shuffles below stay matched
            const float l = lse[r];
            const float* row = logits + r * ld;
            const float* gr = gs_tile + pl;
            float s = 0.f;
            for (int q = g; q < c4n; q += LPP) {
                const float4 v = ld4(row + q * 4);
                const int c = q * 4;
                s += gr[c * pitch] * expf(v.x - l);
                if (c + 1 < C) s += gr[(c + 1) * pitch] * expf(v.y - l);
                if (c + 2 < C) s += gr[(c + 2) * pitch] * expf(v.z - l);
                if (c + 3 < C) s += gr[(c + 3) * pitch] * expf(v.w - l);
            }
            s = grp_sum8(s);
            for (int q = g; q < c4n; q += LPP) {
                const float4 v = ld4(row + q * 4);
                const int c = q * 4;
                float4 d;
                d.x = gs * expf(v.x - l) * (gr[c * pitch] - s);
                d.y = c + 1 < C ? gs * expf(v.y - l) * (gr[(c + 1) * pitch] - s) : 0.f;
                d.z = c + 2 < C ? gs * expf(v.z - l) * (gr[(c + 2) * pitch] - s) : 0.f;
                d.w = c + 3 < C ? gs * expf(v.w - l) * (gr[(c + 3) * pitch] - s) : 0.f;
                st4(dl + r * lddl + q * 4, d);
            }
        }
    }
}

// pixels per backward tile: the [C][TP+1] fp32 slab must fit 64 KB of LDS
int lovasz_bwd_tp(int C) {
    for (int tp = 64; tp >= 8; tp >>= 1)
        if ((size_t)C * (tp + 1) * sizeof(float) <= 64u * 1024u) return tp;
    return 0;
}

struct LovaszLayout {
    size_t keys_a, keys_b, chunk_fg, counts, part, temp, total;
    int nchunks, begin_bit, end_bit, PB, ntiles;
    size_t temp_bytes;
};
// SEGMI_LOVASZ_SORT=rocprim: the device-wide rocPRIM sort over (class, error) instead of the hand-written segmented sort (A/B)
int g_lovasz_sort = -1;       // 0 segmented (hand-written, default), 1 rocPRIM; segmi_lovasz_set_sort / SEGMI_LOVASZ_SORT
bool lovasz_use_rocprim() {
    if (g_lovasz_sort < 0) {
        const char* e = getenv("SEGMI_LOVASZ_SORT");
        g_lovasz_sort = (e && !strcmp(e, "rocprim")) ? 1 : 0;
    }
    return g_lovasz_sort == 1;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool lovasz_layout(long rows, int C, LovaszLayout* L) {
    if (rows <= 0 || C <= 0 || rows >= (1L << 31) || rows >= (1L << 24)) return false;   // fp32-exact cumsums like the reference
    const size_t n = (size_t)rows * C;
    L->nchunks = (int)((rows + CHUNK - 1) / CHUNK);
    int cb = 1;
    while ((1 << cb) < C) ++cb;
    int pb = 1;
    while ((1L << pb) < rows) ++pb;
    if (cb + pb + 32 > 64) return false;                  // class | invalid | 30-bit error | fg | pixel must fit one 64-bit key
    L->PB = pb;
    L->begin_bit = pb + 1;
    L->end_bit = pb + 32 + cb;
    size_t off = 0;
    L->keys_a = off; off += align256(n * 8);
    L->keys_b = off; off += align256(n * 8);
    L->chunk_fg = off; off += align256((size_t)C * L->nchunks * 4);
    L->counts = off; off += align256((size_t)(C + 1) * 4);
    L->part = off; off += align256((size_t)C * L->nchunks * 8);
    // scratch of the sort: per-tile digit histograms [C][ntiles][256] of the segmented sort, or rocPRIM's temporary storage
    // (histograms / lookback state: a host-side size query, nothing is launched)
    L->ntiles = (int)((rows + ST - 1) / ST);
    size_t tb = (size_t)C * L->ntiles * 256 * sizeof(unsigned);
    if (lovasz_use_rocprim()) {
        rocprim::double_buffer<unsigned long long> dk(nullptr, nullptr);
        if (rocprim::radix_sort_keys(nullptr, tb, dk, n, (unsigned)L->begin_bit, (unsigned)L->end_bit, (hipStream_t)0) != hipSuccess) tb = 64u << 20;
    }
    L->temp_bytes = tb;
    L->temp = off; off += align256(tb);
    L->total = off;
    return true;
}

}  // namespace

extern "C" {

int segmi_lovasz_set_sort(int algorithm) {
    if (algorithm != 0 && algorithm != 1) return SEGMI_ERR_BADARG;
    g_lovasz_sort = algorithm;
    return SEGMI_OK;
}

size_t segmi_lovasz_workspace(long rows, int C) {
    LovaszLayout L;
    return lovasz_layout(rows, C, &L) ? L.total : 0;
}

int segmi_lovasz_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float* lse,
                     float* G, int ldg, float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!logits || !target || !lse || !G || !loss_out || C <= 0) return SEGMI_ERR_BADARG;
    LovaszLayout L;
    if (!lovasz_layout(rows, C, &L)) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (ldg & 3) || ldg < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned long long* ka = (unsigned long long*)(ws + L.keys_a);
    unsigned long long* kb = (unsigned long long*)(ws + L.keys_b);
    unsigned* chunk_fg = (unsigned*)(ws + L.chunk_fg);
    unsigned* counts = (unsigned*)(ws + L.counts);
    double* part = (double*)(ws + L.part);

    hipMemsetAsync(counts, 0, (size_t)(C + 1) * 4, st);
    hipMemsetAsync(G, 0, (size_t)rows * ldg * sizeof(float), st);
    long pb = (rows + 31) / 32;
    if (pb > SEGMI_MAX_GRID) pb = SEGMI_MAX_GRID;
    hipLaunchKernelGGL(lovasz_prepare_kernel, dim3((unsigned)pb), dim3(256), (size_t)(C + 1) * 4, st, logits, ld, target, rows, C,
                       ignore_index, lse, counts);
    hipLaunchKernelGGL(lovasz_emit_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logits, ld, target, (const float*)lse,
                       rows, C, ignore_index, L.PB, ka);
    const unsigned long long* ks = ka;
    const bool fused_fg = !lovasz_use_rocprim() && L.nchunks <= SEG_FG_LDS;   // chunk fg counts come out of the last scatter pass
    if (lovasz_use_rocprim()) {
        rocprim::double_buffer<unsigned long long> dk(ka, kb);
        size_t tb = L.temp_bytes;
        if (rocprim::radix_sort_keys(ws + L.temp, tb, dk, (size_t)rows * C, (unsigned)L.begin_bit, (unsigned)L.end_bit, st) != hipSuccess) return SEGMI_ERR_LAUNCH;
        ks = dk.current();
    } else {
        // four stable 8-bit passes over the 31-bit field [invalid | ~error] above the fg bit: ka -> kb -> ka -> kb -> ka
        unsigned* hist = (unsigned*)(ws + L.temp);
        unsigned long long *src = ka, *dst = kb;
        const dim3 tgrid((unsigned)L.ntiles, (unsigned)C);
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = L.PB + 1 + 8 * pass;
            const unsigned mask = pass < 3 ? 0xFFu : 0x7Fu;
            hipLaunchKernelGGL(segsort_hist_kernel, tgrid, dim3(256), 0, st, (const unsigned long long*)src, rows, L.ntiles, shift, mask, (const unsigned*)counts, hist);
            hipLaunchKernelGGL(segsort_scan_kernel, dim3((unsigned)C), dim3(256, SCAN_Q), 0, st, hist, L.ntiles, (const unsigned*)counts);
            const bool count_fg = pass == 3 && fused_fg;
            if (count_fg) hipMemsetAsync(chunk_fg, 0, (size_t)C * L.nchunks * sizeof(unsigned), st);
            hipLaunchKernelGGL(segsort_scatter_kernel, tgrid, dim3(256), 0, st, (const unsigned long long*)src, dst, rows, L.ntiles, shift, mask,
                               (const unsigned*)counts, (const unsigned*)hist, count_fg ? chunk_fg : (unsigned*)nullptr, L.nchunks, L.PB);
            unsigned long long* t = src; src = dst; dst = t;
        }
        ks = src;                                          // == ka after an even number of passes
    }
    dim3 grid((unsigned)L.nchunks, (unsigned)C);
    if (!fused_fg)
        hipLaunchKernelGGL(lovasz_chunk_count_kernel, grid, dim3(256), 0, st, ks, rows, L.nchunks, (const unsigned*)counts, C, L.PB, chunk_fg);
    hipLaunchKernelGGL(lovasz_chunk_scan_kernel, dim3((unsigned)C), dim3(256), 0, st, chunk_fg, L.nchunks, (const unsigned*)counts);
    // The scatter of d loss / d p into a class plane of G is 4-byte writes at random pixels: with the whole plane (rows * 4 B =
    // 8.4 MB at cfg5) in flight an XCD's 4 MB L2 evicts partially written lines.  The pass is therefore run once per PIXEL WINDOW of
    // <= 4.5 MB (the sorted keys are re-read, the writes of a window merge into full lines in L2).  Measured at cfg5 in one call
    // (SEGMI_LOVASZ_WINDOWS overrides): 1 window 69.2 ms/step, 2 windows 68.45, 3 windows 68.67, 4 windows 69.71.
    const dim3 grid1((unsigned)(8 * ((C + 7) / 8)) * (unsigned)L.nchunks);
    int nwin = (int)(((size_t)rows * 4 + (9u << 19) - 1) / (9u << 19));
    if (const char* e = getenv("SEGMI_LOVASZ_WINDOWS")) { const int v = atoi(e); if (v >= 1 && v <= 16) nwin = v; }
    if (nwin < 1) nwin = 1;
    const unsigned per = (unsigned)((rows + nwin - 1) / nwin);
    for (int wdw = 0; wdw < nwin; ++wdw)
        hipLaunchKernelGGL(lovasz_grad_dot_kernel, grid1, dim3(256), 0, st, ks, rows, L.nchunks, (const unsigned*)counts, C, L.PB,
                           (const unsigned*)chunk_fg, G, part, (unsigned)wdw * per, wdw + 1 == nwin ? 0xFFFFFFFFu : (unsigned)(wdw + 1) * per);
    hipLaunchKernelGGL(lovasz_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, (const double*)part, L.nchunks, (const unsigned*)counts, C, rows, loss_out);
    return segmi_launch_status();
}

int segmi_lovasz_bwd(const float* logits, int ld, const float* lse, const float* G, int ldg, long rows, int C,
                     const float* loss_out, const float* grad_out, float* dlogits, int lddl, segmi_stream_t stream) {
    if (!logits || !lse || !G || !loss_out || !grad_out || !dlogits || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((ld & 3) || ld < ((C + 3) & ~3) || (ldg & 3) || ldg < ((C + 3) & ~3) || (lddl & 3) || lddl < ((C + 3) & ~3)) return SEGMI_ERR_ALIGN;
    const int tp = lovasz_bwd_tp(C);
    if (tp == 0) return SEGMI_ERR_BADARG;               // > 1820 classes
    long b = (rows + tp - 1) / tp;
    if (b > 4 * SEGMI_MAX_GRID) b = 4 * SEGMI_MAX_GRID;
    hipLaunchKernelGGL(lovasz_bwd_kernel, dim3((unsigned)b), dim3(256), (size_t)C * (tp + 1) * sizeof(float), (hipStream_t)stream, logits, ld,
                       lse, G, rows, C, tp, loss_out, grad_out, dlogits, lddl);
    return segmi_launch_status();
}

}  // extern "C"
