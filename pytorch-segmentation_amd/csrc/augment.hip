// Training-time augmentation on the device: the cv2 / PIL sequence of the reference's
// BaseDataSet._augmentation (base/base_dataset.py:63-120), _val_augmentation (:40-61) and __getitem__ (:125-136) as gather
// kernels per sample:
//
//   aug_resize   cv2.resize(image, INTER_LINEAR) + cv2.resize(label, INTER_NEAREST) / PIL NEAREST    (:48-50,71-72)  uint8 HWC3 / int32 HW
//   aug_rotate   cv2.warpAffine(getRotationMatrix2D(centre, angle, 1.0)), bilinear image / nearest label, constant border 0 (:76-81)
//   aug_blur     cv2.GaussianBlur(3x3, sigma, BORDER_REFLECT_101)                                     (:113-117)
//   aug_finish   copyMakeBorder(bottom/right, 0) + crop + fliplr + ToTensor + Normalize(mean, std) + label -> int64
//                (:84-110,129-136), written straight into the NHWC-backed fp32 batch (pixel stride 4) the model consumes
//
// cv2's 8-bit paths are FIXED POINT (11-bit resize coefficients, 1/32-pixel warp grid with 15-bit weights, 8.8 Gaussian taps);
// the kernels evaluate exactly that integer arithmetic.  Everything that OpenCV derives in double / float on the host — source
// offsets and coefficients per output column / row, the inverted affine map scaled by 2^10 — arrives as small int32 TABLES
// computed by the host code in the same way (dataloaders/gpu_augment.py), so there is no floating point in the pixel path at
// all and the results are held BIT-EXACTLY to oracle/augment_ref.py, an independent numpy restatement of OpenCV's published
// algorithms.  cv2 itself is not installed in the build image: parity against the library is unpinned (oracle header).
// HBM-bound gathers, one thread per output pixel; off the timed hot path (the metric excludes data loading).
#include "segmi_common.h"

namespace {

__device__ __forceinline__ unsigned char sat_u8i(int v) { return (unsigned char)min(max(v, 0), 255); }

// tab: xs[dw] | xa[dw] (a0 | a1 << 16) | ys[dh] | yb[dh] (b0 | b1 << 16) | lx[dw] | ly[dh]   (int32 each)
// area2x: both scales are exactly 2 — cv::resize turns INTER_LINEAR into the 2x2 box average
__global__ __launch_bounds__(256) void aug_resize_kernel(const unsigned char* __restrict__ img, const int* __restrict__ lab, int sh, int sw,
                                                         unsigned char* __restrict__ oimg, int* __restrict__ olab, int dh, int dw,
                                                         const int* __restrict__ tab, int area2x) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)dh * dw) return;
    const int y = (int)(p / dw), x = (int)(p % dw);
    const int* xs = tab;
    const int* xa = tab + dw;
    const int* ys = tab + 2 * dw;
    const int* yb = ys + dh;
    const int* lx = yb + dh;
    const int* ly = lx + dw;
    if (area2x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned char* q = img + ((long)(2 * y) * sw + 2 * x) * 3 + c;
            oimg[p * 3 + c] = (unsigned char)(((int)q[0] + q[3] + q[(long)sw * 3] + q[(long)sw * 3 + 3] + 2) >> 2);
        }
    } else {
        const int x0 = xs[x], x1 = min(x0 + 1, sw - 1);
        const int a0 = (short)(xa[x] & 0xFFFF), a1 = xa[x] >> 16;
        const int sy = ys[y];
        const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);       // rows clipped, coefficients kept
        const int b0 = (short)(yb[y] & 0xFFFF), b1 = yb[y] >> 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int d0 = (int)img[((long)y0 * sw + x0) * 3 + c] * a0 + (int)img[((long)y0 * sw + x1) * 3 + c] * a1;
            const int d1 = (int)img[((long)y1 * sw + x0) * 3 + c] * a0 + (int)img[((long)y1 * sw + x1) * 3 + c] * a1;
            oimg[p * 3 + c] = sat_u8i((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    olab[p] = lab[(long)ly[y] * sw + lx[x]];
}

// tab: adelta[w] | bdelta[w] | X0[h] | Y0[h]: cv::warpAffine's fixed-point (2^10) inverse map without the rounding offset
__global__ __launch_bounds__(256) void aug_rotate_kernel(const unsigned char* __restrict__ img, const int* __restrict__ lab, int h, int w,
                                                         const int* __restrict__ tab, unsigned char* __restrict__ oimg, int* __restrict__ olab) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)h * w) return;
    const int y = (int)(p / w), x = (int)(p % w);
    const int ad = tab[x], bd = tab[w + x], X0 = tab[2 * w + y], Y0 = tab[2 * w + h + y];
    const int X = (X0 + 16 + ad) >> 5, Y = (Y0 + 16 + bd) >> 5;           // 1/32-pixel grid (arithmetic shifts = floor)
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
    auto px = [&](int yy, int xx, int c) -> int {
        return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (int)img[((long)yy * w + xx) * 3 + c] : 0;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c)
        oimg[p * 3 + c] = sat_u8i((px(sy, sx, c) * w00 + px(sy, sx + 1, c) * w01 + px(sy + 1, sx, c) * w10 + px(sy + 1, sx + 1, c) * w11 + (1 << 14)) >> 15);
    const int nx = (X0 + 512 + ad) >> 10, ny = (Y0 + 512 + bd) >> 10;
    olab[p] = (ny >= 0 && ny < h && nx >= 0 && nx < w) ? lab[(long)ny * w + nx] : 0;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// cv2.GaussianBlur(3x3) on CV_8U: taps {m0, m1, m0} in 8.8 fixed point; AXIS 0 = rows into a uint16 scratch image
// (ufixedpoint16), AXIS 1 = columns with rounding (+2^15) >> 16; BORDER_REFLECT_101
template <int AXIS>
__global__ __launch_bounds__(256) void aug_blur_kernel(const unsigned char* __restrict__ img, const unsigned short* __restrict__ tmp_in, int h, int w,
                                                       int m0, int m1, unsigned short* __restrict__ tmp_out, unsigned char* __restrict__ oimg) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)h * w) return;
    const int y = (int)(p / w), x = (int)(p % w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (AXIS == 0) {
            const int l = img[((long)y * w + reflect101(x - 1, w)) * 3 + c], r = img[((long)y * w + reflect101(x + 1, w)) * 3 + c];
            tmp_out[p * 3 + c] = (unsigned short)(m0 * (l + r) + m1 * (int)img[p * 3 + c]);
        } else {
            const int u = tmp_in[((long)reflect101(y - 1, h) * w + x) * 3 + c], d = tmp_in[((long)reflect101(y + 1, h) * w + x) * 3 + c];
            oimg[p * 3 + c] = sat_u8i((m0 * (u + d) + m1 * (int)tmp_in[p * 3 + c] + (1 << 15)) >> 16);
        }
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
                     int dst_h, int dst_w, const int32_t* tables_dev, int area2x, segmi_stream_t stream) {
    if (!image || !label || !out_image || !out_label || !tables_dev || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return SEGMI_ERR_BADARG;
    if (area2x && (src_h < 2 * dst_h || src_w < 2 * dst_w)) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(aug_resize_kernel, dim3(blocks((long)dst_h * dst_w)), dim3(256), 0, (hipStream_t)stream, image, label, src_h, src_w,
                       out_image, out_label, dst_h, dst_w, tables_dev, area2x ? 1 : 0);
    return segmi_launch_status();
}

int segmi_aug_rotate(const uint8_t* image, const int32_t* label, int h, int w, const int32_t* tables_dev, uint8_t* out_image,
                     int32_t* out_label, segmi_stream_t stream) {
    if (!image || !label || !out_image || !out_label || !tables_dev || h <= 0 || w <= 0) return SEGMI_ERR_BADARG;
    hipLaunchKernelGGL(aug_rotate_kernel, dim3(blocks((long)h * w)), dim3(256), 0, (hipStream_t)stream, image, label, h, w, tables_dev,
                       out_image, out_label);
    return segmi_launch_status();
}

int segmi_aug_blur(const uint8_t* image, int h, int w, int m0, int m1, uint16_t* scratch, uint8_t* out_image, segmi_stream_t stream) {
    if (!image || !out_image || !scratch || h <= 0 || w <= 0 || m0 < 0 || m1 < 0 || 2 * m0 + m1 != 256) return SEGMI_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nb = blocks((long)h * w);
    hipLaunchKernelGGL((aug_blur_kernel<0>), dim3(nb), dim3(256), 0, st, image, (const unsigned short*)nullptr, h, w, m0, m1, scratch, (unsigned char*)nullptr);
    hipLaunchKernelGGL((aug_blur_kernel<1>), dim3(nb), dim3(256), 0, st, (const unsigned char*)nullptr, (const unsigned short*)scratch, h, w, m0, m1,
                       (unsigned short*)nullptr, out_image);
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
