// Welford / Chan running statistics {count, mean, M2} for a float4 channel group — shared by the BN statistics kernels (bn.hip)
// and by the producers that emit BN partials from their epilogue (conv_winograd.hip's output transform).  Internal, not ABI.
#pragma once
#include "segmi_common.h"

namespace {

struct Wf4 {  // Welford state for 4 channels sharing one count
    float n;
    float4 mean, m2;
};
__device__ __forceinline__ void wf_init(Wf4& w) { w.n = 0.f; w.mean = zero4(); w.m2 = zero4(); }
__device__ __forceinline__ void wf_push(Wf4& w, float4 x) {
    w.n += 1.f;
    const float inv = 1.f / w.n;
    float d;
    d = x.x - w.mean.x; w.mean.x += d * inv; w.m2.x += d * (x.x - w.mean.x);
    d = x.y - w.mean.y; w.mean.y += d * inv; w.m2.y += d * (x.y - w.mean.y);
    d = x.z - w.mean.z; w.mean.z += d * inv; w.m2.z += d * (x.z - w.mean.z);
    d = x.w - w.mean.w; w.mean.w += d * inv; w.m2.w += d * (x.w - w.mean.w);
}
__device__ __forceinline__ void chan1(float na, float& ma, float& qa, float nb, float mb, float qb, float n) {
    const float d = mb - ma;
    const float f = nb / n;
    ma += d * f;
    qa += qb + d * d * na * f;
}
__device__ __forceinline__ void wf_merge(Wf4& a, const Wf4& b) {
    const float n = a.n + b.n;
    if (n == 0.f) return;
    chan1(a.n, a.mean.x, a.m2.x, b.n, b.mean.x, b.m2.x, n);
    chan1(a.n, a.mean.y, a.m2.y, b.n, b.mean.y, b.m2.y, n);
    chan1(a.n, a.mean.z, a.m2.z, b.n, b.mean.z, b.m2.z, n);
    chan1(a.n, a.mean.w, a.m2.w, b.n, b.mean.w, b.m2.w, n);
    a.n = n;
}

}  // namespace
