// Training-time augmentation on the device: the cv2 / PIL sequence of the reference's
// BaseDataSet._augmentation (base/base_dataset.py:63-120) and __getitem__ (:125-136) as four gather kernels per sample:
//
//   aug_resize   cv2.resize(image, INTER_LINEAR) + cv2.resize(label, INTER_NEAREST)        (:71-72)   uint8 HWC3 / int32 HW
//   aug_rotate   cv2.warpAffine(getRotationMatrix2D(centre, angle, 1.0)), bilinear image / nearest label, constant border 0 (:76-81)
//   aug_blur     cv2.GaussianBlur(ksize, sigma, BORDER_REFLECT_101), separable                (:113-117)
//   aug_finish   copyMakeBorder(bottom/right, 0) + random crop + fliplr + ToTensor + Normalize(mean, std) + label -> int64
//                (:84-110,129-136), written straight into the NHWC-backed fp32 batch (pixel stride 4) the model consumes
//
// The random decisions (long side, angle, crop origin, flip, sigma) are drawn on the host in the reference's order
// (dataloaders/gpu_augment.py), so a seeded run takes the same decisions; the resampling arithmetic is fp32 where cv2 uses
// fixed point (11-bit resize coefficients, 1/32-pixel warp coordinates — the latter IS mirrored), so results can differ from
// cv2 by one uint8 level at isolated pixels.  cv2 is not installed in the build image: parity against it is unpinned; the
// kernels are held bit-exactly (labels) / within one level (images) to oracle/augment_ref.py, the numpy restatement of the
// same formulas.  HBM-bound gathers, one thread per output pixel; off the timed hot path (the metric excludes data loading).
#include "segmi_common.h"

namespace {

__device__ __forceinline__ unsigned char sat_u8(float v) {
    v = rintf(v);
    return (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
}

// cv2.resize INTER_LINEAR (half-pixel centres, edge clamp) / INTER_NEAREST (floor(dst * scale))
__global__ __launch_bounds__(256) void aug_resize_kernel(const unsigned char* __restrict__ img, const int* __restrict__ lab, int sh, int sw,
                                                         unsigned char* __restrict__ oimg, int* __restrict__ olab, int dh, int dw) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)dh * dw) return;
    const int y = (int)(p / dw), x = (int)(p % dw);
    const float fy = (float)sh / (float)dh, fx = (float)sw / (float)dw;
    float sy = ((float)y + 0.5f) * fy - 0.5f, sx = ((float)x + 0.5f) * fx - 0.5f;
    int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    float wy = sy - (float)y0, wx = sx - (float)x0;
    if (y0 < 0) { y0 = 0; wy = 0.f; }
    if (x0 < 0) { x0 = 0; wx = 0.f; }
    if (y0 >= sh - 1) { y0 = sh - 1; wy = 0.f; }
    if (x0 >= sw - 1) { x0 = sw - 1; wx = 0.f; }
    const int y1 = min(y0 + 1, sh - 1), x1 = min(x0 + 1, sw - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = img[((long)y0 * sw + x0) * 3 + c], b = img[((long)y0 * sw + x1) * 3 + c];
        const float d = img[((long)y1 * sw + x0) * 3 + c], e = img[((long)y1 * sw + x1) * 3 + c];
        oimg[p * 3 + c] = sat_u8((1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * d + wx * e));
    }
    const int ny = min((int)floorf((float)y * fy), sh - 1), nx = min((int)floorf((float)x * fx), sw - 1);
    olab[p] = lab[(long)ny * sw + nx];
}

// cv2.warpAffine with the INVERSE of M = getRotationMatrix2D((w/2, h/2), angle, 1): source coordinates quantised to 1/32 pixel
// (INTER_BITS = 5) for the bilinear image, rounded to the nearest pixel for the label; outside pixels are the border value 0
__global__ __launch_bounds__(256) void aug_rotate_kernel(const unsigned char* __restrict__ img, const int* __restrict__ lab, int h, int w,
                                                         float m00, float m01, float m02, float m10, float m11, float m12,
                                                         unsigned char* __restrict__ oimg, int* __restrict__ olab) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)h * w) return;
    const int y = (int)(p / w), x = (int)(p % w);
    const float sxf = m00 * (float)x + m01 * (float)y + m02, syf = m10 * (float)x + m11 * (float)y + m12;
    const float qx = rintf(sxf * 32.f), qy = rintf(syf * 32.f);          // 1/32-pixel grid
    const int X = (int)qx, Y = (int)qy;
    const int x0 = X >> 5, y0 = Y >> 5;                                   // floor (arithmetic shift)
    const float wx = (float)(X & 31) * (1.f / 32.f), wy = (float)(Y & 31) * (1.f / 32.f);
    auto px = [&](int yy, int xx, int c) -> float {
        return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (float)img[((long)yy * w + xx) * 3 + c] : 0.f;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (1.f - wy) * ((1.f - wx) * px(y0, x0, c) + wx * px(y0, x0 + 1, c)) +
                        wy * ((1.f - wx) * px(y0 + 1, x0, c) + wx * px(y0 + 1, x0 + 1, c));
        oimg[p * 3 + c] = sat_u8(v);
    }
    const int nx = (int)rintf(sxf), ny = (int)rintf(syf);
    olab[p] = (ny >= 0 && ny < h && nx >= 0 && nx < w) ? lab[(long)ny * w + nx] : 0;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// one axis of cv2.GaussianBlur (kernel <= 7 taps: ksize = int(3.3 * sigma) made odd, sigma < 1) with BORDER_REFLECT_101;
// AXIS 0 = along x into a float scratch image, AXIS 1 = along y, rounding to uint8
template <int AXIS>
__global__ __launch_bounds__(256) void aug_blur_kernel(const unsigned char* __restrict__ img, const float* __restrict__ tmp_in, int h, int w,
                                                       int ksize, float k0, float k1, float k2, float k3,
                                                       float* __restrict__ tmp_out, unsigned char* __restrict__ oimg) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)h * w) return;
    const int y = (int)(p / w), x = (int)(p % w);
    const float kk[4] = {k0, k1, k2, k3};        // centre, +-1, +-2, +-3
    const int r = ksize >> 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int d = -r; d <= r; ++d) {
            const float wgt = kk[d < 0 ? -d : d];
            if (AXIS == 0) acc += wgt * (float)img[((long)y * w + reflect101(x + d, w)) * 3 + c];
            else           acc += wgt * tmp_in[((long)reflect101(y + d, h) * w + x) * 3 + c];
        }
        if (AXIS == 0) tmp_out[p * 3 + c] = acc;
        else           oimg[p * 3 + c] = sat_u8(acc);
    }
}

// crop window [sy, sy+crop) x [sx, sx+crop) of the image padded with zeros at the bottom / right, optional horizontal flip,
// ToTensor (v / 255) + Normalize, channels padded to the NHWC-backed pixel stride ld (>= 4, extra channels zero); label -> int64
__global__ __launch_bounds__(256) void aug_finish_kernel(const unsigned char* __restrict__ img, const int* __restrict__ lab, int h, int w,
                                                         int crop_h, int crop_w, int sy, int sx, int flip, float m0, float m1, float m2,
                                                         float s0, float s1, float s2, float* __restrict__ out, int ld,
                                                         int64_t* __restrict__ olab) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)crop_h * crop_w) return;
    const int y = (int)(p / crop_w), x = (int)(p % crop_w);
    const int xs = flip ? crop_w - 1 - x : x;
    const int yy = sy + y, xx = sx + xs;
    const bool in = yy < h && xx < w;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    int l = 0;
    if (in) {
        const unsigned char* q = img + ((long)yy * w + xx) * 3;
        v0 = q[0]; v1 = q[1]; v2 = q[2];
        l = lab[(long)yy * w + xx];
    }
    float* o = out + p * ld;
    o[0] = (v0 / 255.f - m0) / s0; o[1] = (v1 / 255.f - m1) / s1; o[2] = (v2 / 255.f - m2) / s2;   // ToTensor's div(255), Normalize's sub / div
    for (int c = 3; c < ld; ++c) o[c] = 0.f;
    olab[p] = (int64_t)l;
}

int blocks(long n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" {

int segmi_aug_resize(const uint8_t* image, const int32_t* label, int src_h, int src_w, uint8_t* out_image, int32_t* out_label,
                     int dst_h, int dst_w, segmi_stream_t stream) {
    if (!image || !label || !out_image || !out_label || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(aug_resize_kernel, dim3(blocks((long)dst_h * dst_w)), dim3(256), 0, (hipStream_t)stream, image, label, src_h, src_w,
                       out_image, out_label, dst_h, dst_w);
    return segmi_launch_status();
}

int segmi_aug_rotate(const uint8_t* image, const int32_t* label, int h, int w, const float* inv_affine6, uint8_t* out_image,
                     int32_t* out_label, segmi_stream_t stream) {
    if (!image || !label || !out_image || !out_label || !inv_affine6 || h <= 0 || w <= 0) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(aug_rotate_kernel, dim3(blocks((long)h * w)), dim3(256), 0, (hipStream_t)stream, image, label, h, w, inv_affine6[0],
                       inv_affine6[1], inv_affine6[2], inv_affine6[3], inv_affine6[4], inv_affine6[5], out_image, out_label);
    return segmi_launch_status();
}

int segmi_aug_blur(const uint8_t* image, int h, int w, int ksize, const float* kernel_half4, float* scratch, uint8_t* out_image,
                   segmi_stream_t stream) {
    if (!image || !out_image || !scratch || !kernel_half4 || h <= 0 || w <= 0 || ksize < 1 || ksize > 7 || !(ksize & 1)) return SEGMI_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nb = blocks((long)h * w);
    hipLaunchKernelGGL((aug_blur_kernel<0>), dim3(nb), dim3(256), 0, st, image, (const float*)nullptr, h, w, ksize, kernel_half4[0],
                       kernel_half4[1], kernel_half4[2], kernel_half4[3], scratch, (unsigned char*)nullptr);
    hipLaunchKernelGGL((aug_blur_kernel<1>), dim3(nb), dim3(256), 0, st, (const unsigned char*)nullptr, (const float*)scratch, h, w, ksize,
                       kernel_half4[0], kernel_half4[1], kernel_half4[2], kernel_half4[3], (float*)nullptr, out_image);
    return segmi_launch_status();
}

int segmi_aug_finish(const uint8_t* image, const int32_t* label, int h, int w, int crop_h, int crop_w, int start_h, int start_w,
                     int flip, const float* mean3, const float* std3, float* out, int ld, int64_t* out_label, segmi_stream_t stream) {
    if (!image || !label || !out || !out_label || !mean3 || !std3 || h <= 0 || w <= 0 || crop_h <= 0 || crop_w <= 0 || start_h < 0 ||
        start_w < 0 || ld < 4)
        return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(aug_finish_kernel, dim3(blocks((long)crop_h * crop_w)), dim3(256), 0, (hipStream_t)stream, image, label, h, w, crop_h,
                       crop_w, start_h, start_w, flip ? 1 : 0, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out, ld, out_label);
    return segmi_launch_status();
}

}  // extern "C"
