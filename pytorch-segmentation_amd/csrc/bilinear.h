// Bilinear-resize coordinate arithmetic shared by pool_resize.hip (aten::upsample_bilinear2d and its backward) and loss.hip
// (cross entropy evaluated on bilinearly upsampled logits without materialising them).  Internal, not part of the C ABI.
#pragma once
#include "segmi_common.h"

namespace {

// Source coordinate exactly as aten's area_pixel_compute_source_index (fp32):
//   align_corners: src = dst * (in-1)/(out-1)            (scale 0 when out == 1)
//   otherwise    : src = max(0, fma(in/out, dst+0.5, -0.5))
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ float bl_scale(int in, int out, int ac) {
    if (ac) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}
__device__ __forceinline__ Lerp bl_src(int dst, float scale, int in, int ac) {
    float s = ac ? scale * (float)dst : fmaxf(__fmaf_rn(scale, (float)dst + 0.5f, -0.5f), 0.f);
    Lerp L;
    L.i0 = min((int)s, in - 1);
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = s - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

// candidate output range [lo, hi] whose source coordinate can touch input index i
__device__ __forceinline__ void bl_range(int i, float scale, int in, int out, int ac, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    float a, b;
    if (ac) { a = ((float)i - 1.f) / scale; b = ((float)i + 1.f) / scale; }
    else    { a = ((float)i - 0.5f) / scale - 0.5f; b = ((float)i + 1.5f) / scale - 0.5f; }
    lo = max(0, (int)floorf(a) - 1);
    hi = min(out - 1, (int)ceilf(b) + 1);
    if (i == 0) lo = 0;              // clamped sources (src < 0 -> 0)
    if (i == in - 1) hi = out - 1;   // clamped i1
}

}  // namespace
