// Batch normalisation (training and frozen), fused ReLU / residual add, and their backward.
// Replaces aten::native_batch_norm(+_backward), relu_/threshold_backward and add_ at every
// nn.BatchNorm2d / nn.ReLU(inplace=True) / `out += residual` site of the reference
// (models/resnet.py:101-121, models/pspnet.py:17-30,64-70, models/unet.py:12-21,
//  models/deeplabv3_plus.py:70-132,260-330; F.batch_norm fallback utils/sync_batchnorm/batchnorm.py:65-68).
//
// All kernels are HBM-bound streams over an NHWC [rows, C] matrix (rows = N*H*W):
//   bn_stats      1 read          -> packed partial {count, mean, M2} per channel (Welford / Chan)
//   bn_apply      1 read (+1) 1 write: y = x*scale + shift (+ residual), optional ReLU
//   bn_bwd_reduce 2-3 reads       -> {sum dy', sum dy'*xhat}
//   bn_bwd_apply  2-3 reads, 1-2 writes
// The packed partials are what SyncBN all-reduces over RCCL between stats and finalize
// (reference semantics: utils/sync_batchnorm/batchnorm.py:105-145, re-expressed with Chan's merge so
// the multi-GPU result matches single-device global-batch F.batch_norm).
#include "rowgeom.h"
#include "welford.h"
#include <cstdlib>

namespace {

// part: [gridDim.y][3][C4*4] = {n, mean, m2}
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, int ld, long rows, int c4n,
                                                               float* __restrict__ part) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 < c4n;
    Wf4 w;
    wf_init(w);
    // contiguous row range per block: keeps the per-thread Welford count small and exact
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    if (cok)
#pragma unroll 4
        for (long r = r0 + threadIdx.y; r < r1; r += blockDim.y) wf_push(w, ld4(x + r * ld + c4 * 4));
    __shared__ Wf4 sm[256];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    sm[t] = w;
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
        const int C = c4n * 4;
        float* o = part + (long)blockIdx.y * 3 * C;
        st4(o + c4 * 4, make_float4(a.n, a.n, a.n, a.n));
        st4(o + C + c4 * 4, a.mean);
        st4(o + 2 * C + c4 * 4, a.m2);
    }
}

struct FinalizeArgs {               // per-channel epilogue of the statistics (bn_finalize_kernel's arithmetic)
    const float* gamma; const float* beta; float eps, momentum; int clamp_mode;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    float* mean; float* invstd; float* scale; float* shift;
};
// the per-channel operands of the finalisation that do not depend on the statistics: requested at the START of a merge kernel, so
// their HBM round trip overlaps the partials' instead of following the merge (these kernels are 4-7 us of pure latency, 61-141 of
// them per step)
struct FinalizeOperands { float g, b, rm, rv; };
__device__ __forceinline__ FinalizeOperands finalize_operands(const FinalizeArgs& f, int c) {
    FinalizeOperands o;
    o.g = f.gamma ? f.gamma[c] : 1.f; o.b = f.beta ? f.beta[c] : 0.f;
    o.rm = f.running_mean ? f.running_mean[c] : 0.f; o.rv = f.running_mean ? f.running_var[c] : 0.f;
    return o;
}
__device__ __forceinline__ void finalize_channel(const FinalizeArgs& f, int c, float n, float m, float q, const FinalizeOperands& o) {
    // no contraction in here: the fused and the two-call paths inline this function into different kernels and must round alike
    // (HIP's __fmul_rn / __fadd_rn are plain operators, which the compiler contracts as it sees fit; the one fma below is explicit)
#pragma clang fp contract(off)
    const float var = q / n;  // biased
    const float is = f.clamp_mode ? 1.f / sqrtf(fmaxf(var, f.eps)) : 1.f / sqrtf(var + f.eps);
    f.mean[c] = m;
    f.invstd[c] = is;
    const float sc = o.g * is;
    f.scale[c] = sc;
    f.shift[c] = __builtin_fmaf(-m, sc, o.b);
    if (f.running_mean) {
        const float unbiased = q / fmaxf(n - 1.f, 1.f);
        f.running_mean[c] = (1.f - f.momentum) * o.rm + f.momentum * m;          // aten's CPU update: no fma
        f.running_var[c] = (1.f - f.momentum) * o.rv + f.momentum * unbiased;
    }
}
__device__ __forceinline__ void finalize_channel(const FinalizeArgs& f, int c, float n, float m, float q) {
    finalize_channel(f, c, n, m, q, finalize_operands(f, c));
}

// merge nparts packed partials (stride 3*Cp each) into out[3*Cp].  Block = (16 channels, 16 part lanes):
// each lane folds parts y, y+16, ... serially (<= 32 steps at the 512-part cap), then a Chan tree over
// the 16 lanes in LDS — the serial one-thread-per-channel form cost 160 us per BN layer.
// FINAL: single-device BN — the merged statistics go straight to mean/invstd/scale/shift (no packed partial, no second launch)
// blockIdx.y = slice of `slice` consecutive partials (first level of a two-level merge: gridDim.y partials come out, FINAL false);
// gridDim.y == 1 merges everything
template <bool FINAL>
__global__ __launch_bounds__(256) void bn_stats_merge_kernel(const float* __restrict__ part, int nparts, int Cp,
                                                             float* __restrict__ out, FinalizeArgs fin, int slice) {
    const int c = blockIdx.x * 16 + threadIdx.x;
    const bool cok = c < Cp;
    if (gridDim.y > 1) {
        part += (long)blockIdx.y * slice * 3 * Cp;
        nparts = min(slice, nparts - (int)blockIdx.y * slice);
        out += (long)blockIdx.y * 3 * Cp;
    }
    FinalizeOperands fo = {1.f, 0.f, 0.f, 0.f};
    if (FINAL && threadIdx.y == 0 && cok) fo = finalize_operands(fin, c);
    float n = 0.f, m = 0.f, q = 0.f;
    if (cok) {
        // four partials per step are loaded unconditionally BEFORE the (serial) Chan merges, so 12 loads are in flight at once
        for (int i0 = threadIdx.y; i0 < nparts; i0 += 64) {
            float nb[4], mb[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 16 * u;
                const bool ok = i < nparts;
                const float* p = part + (long)(ok ? i : 0) * 3 * Cp;
                nb[u] = ok ? p[c] : 0.f; mb[u] = p[Cp + c]; qb[u] = p[2 * Cp + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (nb[u] > 0.f) {
                    const float nn = n + nb[u];
                    chan1(n, m, q, nb[u], mb[u], qb[u], nn);
                    n = nn;
                }
        }
    }
    __shared__ float sn[16][17], sm[16][17], sq[16][17];
    sn[threadIdx.y][threadIdx.x] = n; sm[threadIdx.y][threadIdx.x] = m; sq[threadIdx.y][threadIdx.x] = q;
    __syncthreads();
    for (int s = 8; s > 0; s >>= 1) {
        if ((int)threadIdx.y < s) {
            float na = sn[threadIdx.y][threadIdx.x], ma = sm[threadIdx.y][threadIdx.x], qa = sq[threadIdx.y][threadIdx.x];
            const float nb = sn[threadIdx.y + s][threadIdx.x];
            if (nb > 0.f) {
                const float nn = na + nb;
                chan1(na, ma, qa, nb, sm[threadIdx.y + s][threadIdx.x], sq[threadIdx.y + s][threadIdx.x], nn);
                sn[threadIdx.y][threadIdx.x] = nn; sm[threadIdx.y][threadIdx.x] = ma; sq[threadIdx.y][threadIdx.x] = qa;
            }
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        if (FINAL) {
            if (c == 0 && fin.num_batches_tracked) *fin.num_batches_tracked += 1;
            finalize_channel(fin, c, sn[0][threadIdx.x], sm[0][threadIdx.x], sq[0][threadIdx.x], fo);
        } else {
            out[c] = sn[0][threadIdx.x]; out[Cp + c] = sm[0][threadIdx.x]; out[2 * Cp + c] = sq[0][threadIdx.x];
        }
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C, int Cp, const float* gamma,
                                   const float* beta, float eps, float momentum, int clamp_mode, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                                   float* scale, float* shift, float* count_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c == 0 && count_out) {       // elements per channel over all partials (the global batch of a SyncBN layer)
        float tot = 0.f;
        for (int i = 0; i < nparts; ++i) tot += part[(long)i * 3 * Cp];
        *count_out = tot;
    }
    if (c >= C) return;
    float n = 0.f, m = 0.f, q = 0.f;
    for (int i = 0; i < nparts; ++i) {
        const float* p = part + (long)i * 3 * Cp;
        const float nb = p[c];
        if (nb > 0.f) {
            const float nn = n + nb;
            chan1(n, m, q, nb, p[Cp + c], p[2 * Cp + c], nn);
            n = nn;
        }
    }
    const FinalizeArgs f = {gamma, beta, eps, momentum, clamp_mode, running_mean, running_var, nullptr, mean, invstd, scale, shift};
    finalize_channel(f, c, n, m, q);
}

__global__ void bn_eval_coeffs_kernel(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                                      int C, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = rm[c];
    invstd[c] = is;
    scale[c] = g * is;
    shift[c] = b - rm[c] * g * is;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ res,
                                                       int ldr, float* __restrict__ y, int ldy, long rows, int c4n,
                                                       const float* __restrict__ scale, const float* __restrict__ shift) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    const float4 sc = ld4(scale + c4 * 4), sh = ld4(shift + c4 * 4);
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        float4 v = ld4(x + r * ldx + c4 * 4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        if (RES) {
            const float4 q = ld4(res + r * ldr + c4 * 4);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        st4(y + r * ldy + c4 * 4, v);
    }
}

// part: [gridDim.y][2][Cp] DOUBLES.  The two sums cancel heavily (sum dy*xhat is orders of magnitude below sum |dy*xhat|), and
// aten's CPU kernel — the reference — accumulates them in double (at::acc_type<float, false>): so does this one.  The kernel is
// HBM-bound (2-3 fp32 tensor reads per element against 2 fp64 adds per element on a 64-lane fp64 pipe): the wider accumulators
// are free, and d gamma / d beta / the dx correction terms carry no accumulation error beyond their final rounding to fp32.
struct D4 { double x, y, z, w; };
__device__ __forceinline__ D4 dzero4() { return D4{0.0, 0.0, 0.0, 0.0}; }
template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                            int ldx, const float* __restrict__ y, int ldy, long rows, int c4n,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            double* __restrict__ part) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 < c4n;
    D4 s0 = dzero4(), s1 = dzero4();
    if (cok) {
        const float4 mu = ld4(mean + c4 * 4);
        float4 sc = zero4(), sh = zero4();
        if (RELU && !y) { sc = ld4(scale + c4 * 4); sh = ld4(shift + c4 * 4); }
        // four rows per trip: all 8 (12 with a saved output) loads are issued before the first use
        const long stride = (long)gridDim.y * blockDim.y;
        auto fold = [&](float4 g, const float4& v, const float4& o) {
            if (RELU) {
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            s0.x += (double)g.x; s0.y += (double)g.y; s0.z += (double)g.z; s0.w += (double)g.w;
            // sum dy * (x - mean) in double like aten (`dotp += (x - mean) * dy` with a double mean); invstd multiplies the block's sum
            s1.x = fma((double)g.x, (double)v.x - (double)mu.x, s1.x); s1.y = fma((double)g.y, (double)v.y - (double)mu.y, s1.y);
            s1.z = fma((double)g.z, (double)v.z - (double)mu.z, s1.z); s1.w = fma((double)g.w, (double)v.w - (double)mu.w, s1.w);
        };
        // ReLU mask: from the saved output, or (no residual) recomputed with the forward's own fmaf — bit-identical
        // to what bn_apply_kernel evaluated, and one full read of y less
        auto outv = [&](long r, const float4& v) {
            return y ? ld4(y + r * ldy + c4 * 4)
                     : make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
        };
        long r = (long)blockIdx.y * blockDim.y + threadIdx.y;
        for (; r + 3 * stride < rows; r += 4 * stride) {
            float4 g[4], v[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { g[u] = ld4(dy + (r + u * stride) * lddy + c4 * 4); v[u] = ld4(x + (r + u * stride) * ldx + c4 * 4); }
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = RELU ? outv(r + u * stride, v[u]) : zero4();
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(g[u], v[u], o[u]);
        }
        for (; r < rows; r += stride) {
            const float4 g = ld4(dy + r * lddy + c4 * 4), v = ld4(x + r * ldx + c4 * 4);
            fold(g, v, RELU ? outv(r, v) : zero4());
        }
    }
    __shared__ D4 sm0[256], sm1[256];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    sm0[t] = s0; sm1[t] = s1;
    __syncthreads();
    for (int s = blockDim.y >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.y < s) {
            D4 a = sm0[t], b = sm0[t + s * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; sm0[t] = a;
            a = sm1[t]; b = sm1[t + s * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; sm1[t] = a;
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        const int Cp = c4n * 4;
        double* o = part + (long)blockIdx.y * 2 * Cp + c4 * 4;
        const D4 a = sm0[t], b = sm1[t];
        const float4 is = ld4(invstd + c4 * 4);
        const double v[8] = {a.x, a.y, a.z, a.w, b.x * (double)is.x, b.y * (double)is.y, b.z * (double)is.z, b.w * (double)is.w};
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        o[Cp] = v[4]; o[Cp + 1] = v[5]; o[Cp + 2] = v[6]; o[Cp + 3] = v[7];
    }
    // (sum_parts_kernel adds the row partials.  Folding that second launch into this kernel — the last workgroup of a channel column
    //  sums the column's partials, found by a ticket counter — was built and measured slower twice in round 5, with agent-scope
    //  fences and with device-scope atomics, profiles/r05_bn_tickets_ab.txt; removed in round 6.)
}
// out[i] = (float) sum_p part[p][i] in double; block = (32 elements, 32 part lanes): every lane issues ALL its (<= 16 at the 512-part
// cap) loads before the first add — the 8-lane form of rounds 4-5 walked 64 partials per lane in 16 dependent round trips: 6.5 us
// per call, 61-141 calls per step — then a fixed-order sum over the 32 lanes through LDS (deterministic).
__global__ __launch_bounds__(1024) void sum_parts_kernel(const double* __restrict__ part, int nparts, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 32 + threadIdx.x;
    double a = 0.0;
    if (i < n) {
        for (int p0 = threadIdx.y; p0 < nparts; p0 += 32 * 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + 32 * u;
                v[u] = p < nparts ? part[(long)p * n + i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
    }
    __shared__ double sm[32][33];
    sm[threadIdx.y][threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.y == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sm[k][threadIdx.x];
        out[i] = (float)t;
    }
}

template <bool RELU, bool TRAIN, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                           int ldx, const float* __restrict__ y, int ldy, long rows, int c4n,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ sums,
                                                           float inv_count, const float* __restrict__ count_dev,
                                                           float* __restrict__ dx, int lddx,
                                                           float* __restrict__ dres, int lddres) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    if (TRAIN && count_dev) inv_count = 1.f / count_dev[0];
    const int Cp = c4n * 4;
    const float4 sc = ld4(scale + c4 * 4);
    float4 sh = zero4();
    if (RELU && !y) sh = ld4(shift + c4 * 4);
    float4 mu = zero4(), is = zero4(), k0 = zero4(), k1 = zero4();
    if (TRAIN) {
        mu = ld4(mean + c4 * 4); is = ld4(invstd + c4 * 4);
        const float4 a = ld4(sums + c4 * 4), b = ld4(sums + Cp + c4 * 4);
        k0 = make_float4(a.x * inv_count, a.y * inv_count, a.z * inv_count, a.w * inv_count);
        k1 = make_float4(b.x * inv_count, b.y * inv_count, b.z * inv_count, b.w * inv_count);
    }
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        float4 g = ld4(dy + r * lddy + c4 * 4);
        float4 v = zero4();
        if (TRAIN || (RELU && !y)) v = ld4(x + r * ldx + c4 * 4);
        if (RELU) {
            const float4 o = y ? ld4(y + r * ldy + c4 * 4)
                               : make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (DRES) st4(dres + r * lddres + c4 * 4, g);
        float4 d;
        if (TRAIN) {
            d.x = sc.x * (g.x - k0.x - (v.x - mu.x) * is.x * k1.x);
            d.y = sc.y * (g.y - k0.y - (v.y - mu.y) * is.y * k1.y);
            d.z = sc.z * (g.z - k0.z - (v.z - mu.z) * is.z * k1.z);
            d.w = sc.w * (g.w - k0.w - (v.w - mu.w) * is.w * k1.w);
        } else {
            d = make_float4(sc.x * g.x, sc.y * g.y, sc.z * g.z, sc.w * g.w);
        }
        st4(dx + r * lddx + c4 * 4, d);
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       long rows, int c4n) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        float4 v = ld4(x + r * ldx + c4 * 4);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        st4(y + r * ldy + c4 * 4, v);
    }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y,
                                                       int ldy, float* __restrict__ dx, int lddx, long rows, int c4n) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        float4 g = ld4(dy + r * lddy + c4 * 4);
        const float4 o = ld4(y + r * ldy + c4 * 4);
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        st4(dx + r * lddx + c4 * 4, g);
    }
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                  float* __restrict__ o, int ldo, long rows, int c4n) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= c4n) return;
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < rows; r += (long)gridDim.y * blockDim.y) {
        float4 u = ld4(a + r * lda + c4 * 4);
        const float4 v = ld4(b + r * ldb + c4 * 4);
        u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
        st4(o + r * ldo + c4 * 4, u);
    }
}


// ---- double-precision statistics (SEGMI_BN_STATS_F64=1, experiment / A-B): per-thread sums of x and x*x in double (aten's CPU
// batch_norm accumulates in double), block tree in LDS, partial {n, sum, sumsq} doubles; the merge adds partials and finalizes with
// var = sumsq / n - mean^2 in double.  part: [gridDim.y][3][C4*4] doubles.
__global__ __launch_bounds__(256) void bn_stats_partial_f64_kernel(const float* __restrict__ x, int ld, long rows, int c4n, double* __restrict__ part) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool cok = c4 < c4n;
    D4 s = dzero4(), q = dzero4();
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    double cnt = 0.0;
    if (cok)
#pragma unroll 4
        for (long r = r0 + threadIdx.y; r < r1; r += blockDim.y) {
            const float4 v = ld4(x + r * ld + c4 * 4);
            const double a = v.x, b = v.y, c = v.z, d = v.w;
            s.x += a; s.y += b; s.z += c; s.w += d;
            q.x = fma(a, a, q.x); q.y = fma(b, b, q.y); q.z = fma(c, c, q.z); q.w = fma(d, d, q.w);
            cnt += 1.0;
        }
    __shared__ D4 ss[256], sq[256];
    __shared__ double sn[256];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    ss[t] = s; sq[t] = q; sn[t] = cnt;
    __syncthreads();
    for (int k = blockDim.y >> 1; k > 0; k >>= 1) {
        if ((int)threadIdx.y < k) {
            D4 a = ss[t], b = ss[t + k * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; ss[t] = a;
            a = sq[t]; b = sq[t + k * blockDim.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; sq[t] = a;
            sn[t] += sn[t + k * blockDim.x];
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        const int C = c4n * 4;
        double* o = part + (long)blockIdx.y * 3 * C + c4 * 4;
        const D4 a = ss[t], b = sq[t];
        o[0] = o[1] = o[2] = o[3] = sn[t];
        o[C] = a.x; o[C + 1] = a.y; o[C + 2] = a.z; o[C + 3] = a.w;
        o[2 * C] = b.x; o[2 * C + 1] = b.y; o[2 * C + 2] = b.z; o[2 * C + 3] = b.w;
    }
}
template <bool FINAL>
__global__ __launch_bounds__(256) void bn_stats_merge_f64_kernel(const double* __restrict__ part, int nparts, int Cp, float* __restrict__ out, FinalizeArgs fin) {
    const int c = blockIdx.x * 16 + threadIdx.x;
    const bool cok = c < Cp;
    double n = 0.0, s = 0.0, q = 0.0;
    if (cok)
        for (int i = threadIdx.y; i < nparts; i += 16) {
            const double* p = part + (long)i * 3 * Cp;
            n += p[c]; s += p[Cp + c]; q += p[2 * Cp + c];
        }
    __shared__ double sn[16][17], ss[16][17], sq[16][17];
    sn[threadIdx.y][threadIdx.x] = n; ss[threadIdx.y][threadIdx.x] = s; sq[threadIdx.y][threadIdx.x] = q;
    __syncthreads();
    for (int k = 8; k > 0; k >>= 1) {
        if ((int)threadIdx.y < k) {
            sn[threadIdx.y][threadIdx.x] += sn[threadIdx.y + k][threadIdx.x];
            ss[threadIdx.y][threadIdx.x] += ss[threadIdx.y + k][threadIdx.x];
            sq[threadIdx.y][threadIdx.x] += sq[threadIdx.y + k][threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.y == 0 && cok) {
        const double nn = sn[0][threadIdx.x], mean = ss[0][threadIdx.x] / nn;
        const double m2 = fmax(sq[0][threadIdx.x] - mean * ss[0][threadIdx.x], 0.0);
        if (FINAL) {
            if (c == 0 && fin.num_batches_tracked) *fin.num_batches_tracked += 1;
            finalize_channel(fin, c, (float)nn, (float)mean, (float)m2);
        } else {
            out[c] = (float)nn; out[Cp + c] = (float)mean; out[2 * Cp + c] = (float)m2;
        }
    }
}
int g_stats_f64 = -1;
bool stats_f64() {
    if (g_stats_f64 < 0) {
        const char* e = getenv("SEGMI_BN_STATS_F64");
        g_stats_f64 = (e && atoi(e) == 1) ? 1 : 0;
    }
    return g_stats_f64 == 1;
}

constexpr int STATS_MAX_PARTS = 512;
int stats_parts(long rows) {
    long p = (rows + 127) / 128;  // >= 128 rows per partial block
    if (p < 1) p = 1;
    if (p > STATS_MAX_PARTS) p = STATS_MAX_PARTS;
    return (int)p;
}
bool ld_ok(int ld, int C) { return ld >= ((C + 3) & ~3) && (ld & 3) == 0; }

}  // namespace

extern "C" {

size_t segmi_bn_stats_workspace(long rows, int C) {
    const int Cp = (C + 3) & ~3;
    return (size_t)stats_parts(rows) * 3 * Cp * sizeof(double);      // (fp32 partials, or doubles under SEGMI_BN_STATS_F64)
}

int segmi_bn_stats(const float* x, int ld, long rows, int C, float* partial, void* workspace, size_t workspace_bytes,
                   segmi_stream_t stream) {
    if (!x || !partial || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((C & 3) || !ld_ok(ld, C)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_bn_stats_workspace(rows, C)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int parts = stats_parts(rows);
    RowGeom g = row_geom(rows, C, 1, 1);
    g.grid.y = parts;
    if (stats_f64()) {
        hipLaunchKernelGGL(bn_stats_partial_f64_kernel, g.grid, g.block, 0, st, x, ld, rows, g.c4, (double*)workspace);
        hipLaunchKernelGGL((bn_stats_merge_f64_kernel<false>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, (const double*)workspace, parts, C, partial, FinalizeArgs{});
        return segmi_launch_status();
    }
    hipLaunchKernelGGL(bn_stats_partial_kernel, g.grid, g.block, 0, st, x, ld, rows, g.c4, (float*)workspace);
    hipLaunchKernelGGL((bn_stats_merge_kernel<false>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, (const float*)workspace, parts, C, partial, FinalizeArgs{}, 0);
    return segmi_launch_status();
}

int segmi_bn_stats_finalize(const float* x, int ld, long rows, int C, const float* gamma, const float* beta, float eps,
                            float momentum, int clamp_mode, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                            void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!x || rows <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return SEGMI_ERR_BADARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SEGMI_ERR_BADARG;
    if ((C & 3) || !ld_ok(ld, C)) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_bn_stats_workspace(rows, C)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int parts = stats_parts(rows);
    RowGeom g = row_geom(rows, C, 1, 1);
    g.grid.y = parts;
    const FinalizeArgs f = {gamma, beta, eps, momentum, clamp_mode, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift};
    if (stats_f64()) {
        hipLaunchKernelGGL(bn_stats_partial_f64_kernel, g.grid, g.block, 0, st, x, ld, rows, g.c4, (double*)workspace);
        hipLaunchKernelGGL((bn_stats_merge_f64_kernel<true>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, (const double*)workspace, parts, C, (float*)nullptr, f);
        return segmi_launch_status();
    }
    hipLaunchKernelGGL(bn_stats_partial_kernel, g.grid, g.block, 0, st, x, ld, rows, g.c4, (float*)workspace);
    hipLaunchKernelGGL((bn_stats_merge_kernel<true>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, (const float*)workspace, parts, C, (float*)nullptr, f, 0);
    return segmi_launch_status();
}

// ---- statistics from partials a PRODUCER already wrote (the convolution's BN-statistics epilogue, segmi_conv2d_fwd_stats):
// up to 512 partials merge in one launch; more (row tiles of the 128x128 / 256x256 maps) go through a first level of 64-partial
// slices into the workspace.
static int merge_level1_parts(int nparts) { return nparts > STATS_MAX_PARTS ? segmi_cdiv(nparts, 64) : 0; }
size_t segmi_bn_parts_workspace(int nparts, int C) { return (size_t)merge_level1_parts(nparts) * 3 * C * sizeof(float) + 16; }

static int merge_parts(const float* partials, int nparts, int C, float* out, const FinalizeArgs* fin, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    const float* src = partials;
    int n = nparts;
    const int l1 = merge_level1_parts(nparts);
    if (l1) {
        if (!workspace || workspace_bytes < segmi_bn_parts_workspace(nparts, C)) return SEGMI_ERR_WORKSPACE;
        hipLaunchKernelGGL((bn_stats_merge_kernel<false>), dim3(segmi_cdiv(C, 16), l1), dim3(16, 16), 0, st, partials, nparts, C,
                           (float*)workspace, FinalizeArgs{}, 64);
        src = (const float*)workspace;
        n = l1;
    }
    if (fin) hipLaunchKernelGGL((bn_stats_merge_kernel<true>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, src, n, C, (float*)nullptr, *fin, 0);
    else     hipLaunchKernelGGL((bn_stats_merge_kernel<false>), dim3(segmi_cdiv(C, 16)), dim3(16, 16), 0, st, src, n, C, out, FinalizeArgs{}, 0);
    return segmi_launch_status();
}

int segmi_bn_stats_from_parts(const float* partials, int nparts, int C, float* partial, void* workspace, size_t workspace_bytes,
                              segmi_stream_t stream) {
    if (!partials || !partial || nparts <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if (C & 3) return SEGMI_ERR_ALIGN;
    return merge_parts(partials, nparts, C, partial, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

int segmi_bn_finalize_from_parts(const float* partials, int nparts, int C, const float* gamma, const float* beta, float eps,
                                 float momentum, int clamp_mode, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                                 void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!partials || nparts <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return SEGMI_ERR_BADARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SEGMI_ERR_BADARG;
    if (C & 3) return SEGMI_ERR_ALIGN;
    const FinalizeArgs f = {gamma, beta, eps, momentum, clamp_mode, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift};
    return merge_parts(partials, nparts, C, nullptr, &f, workspace, workspace_bytes, (hipStream_t)stream);
}

int segmi_bn_finalize(const float* partials, int nparts, int C, const float* gamma, const float* beta, float eps,
                      float momentum, int clamp_mode, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                      float* count_out, segmi_stream_t stream) {
    if (!partials || nparts <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return SEGMI_ERR_BADARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(segmi_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, partials, nparts, C, C,
                       gamma, beta, eps, momentum, clamp_mode, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift,
                       count_out);
    return segmi_launch_status();
}

int segmi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                         float eps, int C, float* mean, float* invstd, float* scale, float* shift,
                         segmi_stream_t stream) {
    if (!running_mean || !running_var || C <= 0 || !mean || !invstd || !scale || !shift) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(segmi_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                       running_var, gamma, beta, eps, C, mean, invstd, scale, shift);
    return segmi_launch_status();
}

int segmi_bn_apply(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy, long rows, int C,
                   const float* scale, const float* shift, int relu, segmi_stream_t stream) {
    if (!x || !y || !scale || !shift || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if ((C & 3) || !ld_ok(ldx, C) || !ld_ok(ldy, C) || (residual && !ld_ok(ldr, C))) return SEGMI_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
#define LAUNCH_APPLY(R, S) hipLaunchKernelGGL((bn_apply_kernel<R, S>), g.grid, g.block, 0, st, x, ldx, residual, ldr, y, ldy, rows, g.c4, scale, shift)
    if (relu) { if (residual) LAUNCH_APPLY(true, true); else LAUNCH_APPLY(true, false); }
    else      { if (residual) LAUNCH_APPLY(false, true); else LAUNCH_APPLY(false, false); }
#undef LAUNCH_APPLY
    return segmi_launch_status();
}

// Row partials of the backward reduction: about 2048 blocks in total (8 per CU) whatever the channel count — a 256-channel
// layer is one block column wide and ran on half the chip with the former rows/256 rule — but >= 16 rows per thread.
static int bwd_parts(long rows, int C) {
    const RowGeom g = row_geom(rows, C, 1, 1);
    // (round 6, alternating cfg2 runs: 1024 / 1536 / 3072 workgroups 57.21 / 57.15 / 57.03 ms against 56.4-56.9 for 2048 — the kernel
    //  holds 116 VGPRs, 4 workgroups per CU: 2048 is two full rounds.  Walking the rows of bn_bwd_reduce / bn_bwd_apply / bn_apply from
    //  the last one down, so that a sweep starts with the rows the kernel before it touched last, changed nothing either: 56.2-56.8
    //  against 56.0-56.8 ms for every combination — the 256 MB memory-side cache does not turn that recency into hits.
    //  profiles/r06_bn_order_and_blocks_ab.txt)
    long p = 2048 / (long)g.grid.x;
    const long cap = (rows + (long)g.ry * 16 - 1) / ((long)g.ry * 16);
    if (p > cap) p = cap;
    if (p < 1) p = 1;
    if (p > STATS_MAX_PARTS) p = STATS_MAX_PARTS;          // (1024 / 2048 partials measured at cfg2: 55.76 / 55.92 ms against 55.79: no gain)
    return (int)p;
}
size_t segmi_bn_bwd_reduce_workspace(long rows, int C) {
    const int Cp = (C + 3) & ~3;
    return (size_t)bwd_parts(rows, C) * 2 * Cp * sizeof(double);
}

int segmi_bn_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, long rows, int C,
                        const float* mean, const float* invstd, const float* scale, const float* shift, int relu, float* sums,
                        void* workspace, size_t workspace_bytes, segmi_stream_t stream) {
    if (!dy || !x || !mean || !invstd || !sums || rows <= 0 || C <= 0 || (relu && !y && (!scale || !shift))) return SEGMI_ERR_BADARG;
    if ((C & 3) || !ld_ok(lddy, C) || !ld_ok(ldx, C) || (relu && y && !ld_ok(ldy, C))) return SEGMI_ERR_ALIGN;
    if (!workspace || workspace_bytes < segmi_bn_bwd_reduce_workspace(rows, C)) return SEGMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int parts = bwd_parts(rows, C);
    RowGeom g = row_geom(rows, C, 1, 1);
    g.grid.y = parts;
    if ((uintptr_t)workspace & 7) return SEGMI_ERR_ALIGN;
    if (relu) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true>), g.grid, g.block, 0, st, dy, lddy, x, ldx, y, ldy, rows, g.c4, mean, invstd, scale, shift, (double*)workspace);
    else      hipLaunchKernelGGL((bn_bwd_reduce_kernel<false>), g.grid, g.block, 0, st, dy, lddy, x, ldx, y, ldy, rows, g.c4, mean, invstd, scale, shift, (double*)workspace);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(segmi_cdiv(2 * C, 32)), dim3(32, 32), 0, st, (const double*)workspace, parts, 2 * C, sums);
    return segmi_launch_status();
}

int segmi_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, long rows, int C,
                       const float* mean, const float* invstd, const float* scale, const float* shift, const float* sums,
                       float count, const float* count_dev, int relu, int training, float* dx, int lddx, float* dres,
                       int lddres, segmi_stream_t stream) {
    if (!dy || !scale || !dx || rows <= 0 || C <= 0 || (relu && !y && (!x || !shift))) return SEGMI_ERR_BADARG;
    if (training && (!x || !mean || !invstd || !sums || (!count_dev && count <= 0.f))) return SEGMI_ERR_BADARG;
    const bool need_x = training || (relu && !y);
    if ((C & 3) || !ld_ok(lddy, C) || !ld_ok(lddx, C) || (need_x && !ld_ok(ldx, C)) || (relu && y && !ld_ok(ldy, C)) ||
        (dres && !ld_ok(lddres, C)))
        return SEGMI_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
    const float inv = (training && !count_dev) ? 1.f / count : 0.f;
#define LAUNCH_BA(R, T, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<R, T, D>), g.grid, g.block, 0, st, dy, lddy, x, ldx, y, ldy, rows, g.c4, mean, invstd, scale, shift, sums, inv, count_dev, dx, lddx, dres, lddres)
    const int key = (relu ? 4 : 0) | (training ? 2 : 0) | (dres ? 1 : 0);
    switch (key) {
        case 0: LAUNCH_BA(false, false, false); break;
        case 1: LAUNCH_BA(false, false, true); break;
        case 2: LAUNCH_BA(false, true, false); break;
        case 3: LAUNCH_BA(false, true, true); break;
        case 4: LAUNCH_BA(true, false, false); break;
        case 5: LAUNCH_BA(true, false, true); break;
        case 6: LAUNCH_BA(true, true, false); break;
        default: LAUNCH_BA(true, true, true); break;
    }
#undef LAUNCH_BA
    return segmi_launch_status();
}

int segmi_relu_fwd(const float* x, int ldx, float* y, int ldy, long rows, int C, segmi_stream_t stream) {
    if (!x || !y || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if (!ld_ok(ldx, C) || !ld_ok(ldy, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(relu_fwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, x, ldx, y, ldy, rows, g.c4);
    return segmi_launch_status();
}
int segmi_relu_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, long rows, int C,
                   segmi_stream_t stream) {
    if (!dy || !y || !dx || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if (!ld_ok(lddy, C) || !ld_ok(ldy, C) || !ld_ok(lddx, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(relu_bwd_kernel, g.grid, g.block, 0, (hipStream_t)stream, dy, lddy, y, ldy, dx, lddx, rows, g.c4);
    return segmi_launch_status();
}
int segmi_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long rows, int C,
              segmi_stream_t stream) {
    if (!a || !b || !out || rows <= 0 || C <= 0) return SEGMI_ERR_BADARG;
    if (!ld_ok(lda, C) || !ld_ok(ldb, C) || !ld_ok(ldo, C)) return SEGMI_ERR_ALIGN;
    RowGeom g = row_geom(rows, C, 4, SEGMI_MAX_GRID);
    hipLaunchKernelGGL(add_kernel, g.grid, g.block, 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, g.c4);
    return segmi_launch_status();
}

}  // extern "C"
