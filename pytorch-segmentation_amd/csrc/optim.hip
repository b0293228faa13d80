// Fused multi-tensor SGD step (momentum, weight decay, per-group learning rates).
// Replaces torch.optim.SGD.step as configured by the reference (config.json:48-52 `SGD`, lr 0.01, momentum 0.9,
// weight_decay 1e-4; differential learning rates base/base_trainer.py:46-57: decoder lr, backbone lr/10):
//     g   = grad + weight_decay * p
//     buf = momentum * buf + g            (buf starts at 0, so the first step gives buf = g like torch)
//     p  -= lr * buf
// ONE launch walks a device-resident chunk table (tensor pointers + 64 K-element ranges, built once); the per-group
// hyper-parameters travel as kernel arguments because the schedulers change lr (and OneCycle the momentum) every iteration.
// HBM-bound: 5 float streams per parameter (read p, g, buf; write p, buf).
#include "segmi_common.h"

namespace {

constexpr int MAX_GROUPS = 8;
constexpr int CHUNK_ELEMS = 65536;

struct SgdHyper {
    float lr[MAX_GROUPS], wd[MAX_GROUPS], mom[MAX_GROUPS];
};

// DEVHYPER: the hyper-parameters are read from device memory (layout of SgdHyper) instead of the kernel arguments, so a step
// captured into a hipGraph follows the lr schedule: the host rewrites a pinned staging buffer and the captured H2D copy node
// refreshes `hd` on every replay (kernel arguments would be frozen at capture).
template <bool DEVHYPER>
__global__ __launch_bounds__(256) void sgd_multi_kernel(const segmi_sgd_chunk* __restrict__ table, SgdHyper h, const SgdHyper* __restrict__ hd) {
    const segmi_sgd_chunk c = table[blockIdx.x];
    const float lr = DEVHYPER ? hd->lr[c.group] : h.lr[c.group];
    const float wd = DEVHYPER ? hd->wd[c.group] : h.wd[c.group];
    const float mom = DEVHYPER ? hd->mom[c.group] : h.mom[c.group];
    float* __restrict__ p = c.param;
    const float* __restrict__ g = c.grad;
    float* __restrict__ m = c.momentum;
    const long n = c.count;
    if (c.vec4) {
        for (long i = threadIdx.x * 4; i < n; i += 256 * 4) {
            float4 pv = ld4(p + i), mv = ld4(m + i);
            const float4 gv = ld4(g + i);
            mv.x = fmaf(mom, mv.x, fmaf(wd, pv.x, gv.x)); mv.y = fmaf(mom, mv.y, fmaf(wd, pv.y, gv.y));
            mv.z = fmaf(mom, mv.z, fmaf(wd, pv.z, gv.z)); mv.w = fmaf(mom, mv.w, fmaf(wd, pv.w, gv.w));
            pv.x = fmaf(-lr, mv.x, pv.x); pv.y = fmaf(-lr, mv.y, pv.y); pv.z = fmaf(-lr, mv.z, pv.z); pv.w = fmaf(-lr, mv.w, pv.w);
            st4(m + i, mv);
            st4(p + i, pv);
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 256) {
            const float pv = p[i];
            const float mv = fmaf(mom, m[i], fmaf(wd, pv, g[i]));
            m[i] = mv;
            p[i] = fmaf(-lr, mv, pv);
        }
    }
}

}  // namespace

extern "C" {

int segmi_sgd_chunk_elems(void) { return CHUNK_ELEMS; }

int segmi_sgd_step(const segmi_sgd_chunk* table_dev, int nchunks, const float* lr_host, const float* weight_decay_host,
                   const float* momentum_host, int ngroups, segmi_stream_t stream) {
    if (!table_dev || nchunks <= 0 || !lr_host || !weight_decay_host || !momentum_host || ngroups <= 0 || ngroups > MAX_GROUPS)
        return SEGMI_ERR_BADARG;
    SgdHyper h;
    for (int i = 0; i < MAX_GROUPS; ++i) {
        h.lr[i] = i < ngroups ? lr_host[i] : 0.f;
        h.wd[i] = i < ngroups ? weight_decay_host[i] : 0.f;
        h.mom[i] = i < ngroups ? momentum_host[i] : 0.f;
    }
    hipLaunchKernelGGL((sgd_multi_kernel<false>), dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, table_dev, h,
                       (const SgdHyper*)nullptr);
    return segmi_launch_status();
}

int segmi_sgd_hyper_floats(void) { return 3 * MAX_GROUPS; }

int segmi_sgd_step_dev(const segmi_sgd_chunk* table_dev, int nchunks, const float* hyper_dev, segmi_stream_t stream) {
    if (!table_dev || nchunks <= 0 || !hyper_dev) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL((sgd_multi_kernel<true>), dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, table_dev, SgdHyper{},
                       reinterpret_cast<const SgdHyper*>(hyper_dev));
    return segmi_launch_status();
}

}  // extern "C"
