/* segmi.h — C ABI of libsegmi.so, the MI355X (gfx950) kernel library behind the drop-in
 * `models.*`, `utils.losses.*` and `utils.sync_batchnorm.*` modules.
 *
 * The reference (yassouali/pytorch-segmentation) has no FFI: every FLOP of its hot path is an ATen
 * operator reached through nn.Module.  Each entry point below therefore cites the reference call
 * site(s) whose ATen operator it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch types.  All pointers are DEVICE pointers unless a name ends
 *    in `_host`.  The caller owns every buffer (activations, workspaces); nothing is allocated,
 *    freed or synchronised inside the library.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *  - activations are NHWC fp32: element (n,h,w,c) lives at ((n*H+h)*W+w)*ld + c with ld >= C the
 *    pixel stride in elements ("ld*" arguments), which lets producers write channel slices of a
 *    concat buffer in place.  All ld and all base pointers must be multiples of 4 elements (16 B).
 *  - filters are KRSC fp32 (K output channels, RxS taps, C input channels, C contiguous).
 *  - return value: SEGMI_OK (0) or a negative segmi_status; segmi_strerror() names it.
 */
#ifndef SEGMI_H
#define SEGMI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* segmi_stream_t; /* hipStream_t */

enum segmi_status {
    SEGMI_OK = 0,
    SEGMI_ERR_BADARG = -1,    /* null pointer, non-positive size, unsupported combination */
    SEGMI_ERR_ALIGN = -2,     /* ld / channel count / pointer not 16-byte aligned */
    SEGMI_ERR_WORKSPACE = -3, /* workspace smaller than segmi_*_workspace() reports */
    SEGMI_ERR_LAUNCH = -4     /* hipGetLastError() after the launch */
};
const char* segmi_strerror(int status);
/* library / ABI version, bumped on any signature change */
int segmi_abi_version(void);

/* ------------------------------------------------------------------ layout transforms */
/* NCHW-contiguous -> NHWC(ld) (zero-filling channels C..ld-1) and back.  Used only at the model
 * boundary (reference: data arrives NCHW from base/base_dataset.py:125-136). */
int segmi_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int ld, segmi_stream_t stream);
int segmi_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int ld, segmi_stream_t stream);
/* dst[row, 0:C] = src[row, 0:C] for `rows` rows with independent row strides; dst columns
 * C..Cfill-1 are zero-filled (Cfill >= C).  Replaces torch.cat (models/pspnet.py:37,
 * models/deeplabv3_plus.py:293,329, models/unet.py:56) and pads/crops filter channels. */
int segmi_copy_rows(const float* src, int ld_src, float* dst, int ld_dst, long rows, int C, int Cfill, segmi_stream_t stream);

/* ------------------------------------------------------------------ dense convolution (K1)
 * Replaces aten::conv2d / convolution_backward at every nn.Conv2d call site:
 * models/resnet.py:80-87,137-145,184-185; models/pspnet.py:18,27,61,65,69;
 * models/deeplabv3_plus.py:21,80,94,143,146,256,275,279,306,312-319; models/unet.py:15,18,77.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), LDS-staged NHWC / KRSC tiles. */
typedef struct segmi_conv_desc {
    int N, H, W, C; /* input  x [N,H,W,C], pixel stride ldx; C % 4 == 0 */
    int K, R, S;    /* filter w [K,R,S,C] */
    int P, Q;       /* output y [N,P,Q,K], pixel stride ldy; P = (H+2*pad-dil*(R-1)-1)/stride+1 */
    int stride, pad, dil;
    int ldx, ldy;
} segmi_conv_desc;

/* y = conv(x, w) (+ bias[k]) (+ y if accumulate).  Problems with very few output tiles and a long reduction (the PSP
 * pyramid's 1x1 convolutions on 1x1..6x6 maps) split the reduction across workgroups through `workspace`
 * (segmi_conv2d_fwd_workspace() bytes, 0 for everything else; deterministic fixed-order sum; passing workspace = NULL
 * opts out of the split). */
size_t segmi_conv2d_fwd_workspace(const segmi_conv_desc* d);
int segmi_conv2d_fwd(const segmi_conv_desc* d, const float* x, const float* w_krsc, const float* bias, float* y,
                     int accumulate, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
/* Forward convolution that feeds a BatchNorm (the conv -> bn pairs of models/resnet.py:105-121, models/unet.py:12-21,
 * models/deeplabv3_plus.py:80-86): y = conv(x, w) (+ bias) AND, from the output tile while it is still in registers, the
 * per-channel Welford partials {count, mean, M2} of every row tile: stats_partials [parts][3][K] floats, parts =
 * segmi_conv2d_fwd_stats_parts(d) (0 = this problem has no such epilogue: K % 4 != 0, split-reduction launches of tiny outputs,
 * operands beyond 4 GiB; call segmi_conv2d_fwd then).  The partials are what segmi_bn_stats would compute from y — the BN layer
 * merges them (segmi_bn_finalize_from_parts / segmi_bn_stats_from_parts) and never reads y for its statistics:
 * one HBM pass of nn.BatchNorm2d's aten::native_batch_norm less (14.2 -> 9.5 GB per cfg2 step). */
int segmi_conv2d_fwd_stats_parts(const segmi_conv_desc* d);
int segmi_conv2d_fwd_stats(const segmi_conv_desc* d, const float* x, const float* w_krsc, const float* bias, float* y,
                           float* stats_partials, segmi_stream_t stream);
/* dx = conv_transpose(dy, w) (+ dx if accumulate).  w_crsk is the filter re-laid as [C,R,S,K]
 * (segmi_filter_krsc_to_crsk); requires K % 4 == 0 padding handled by the caller via ldy. */
int segmi_conv2d_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                       segmi_stream_t stream);
/* dw[K,R,S,C] = sum_{n,p,q} dy (x) x ; deterministic split-K through `workspace`. */
size_t segmi_conv2d_wgrad_workspace(const segmi_conv_desc* d);
int segmi_conv2d_wgrad(const segmi_conv_desc* d, const float* x, const float* dy, float* dw_krsc, void* workspace,
                       size_t workspace_bytes, segmi_stream_t stream);
/* Name of the kernel variant the dispatcher picks for this problem (op: 0 fwd, 1 dgrad, 2 wgrad),
 * as it appears in a rocprofv3 kernel trace; used by bench.py to attribute measured time. */
int segmi_conv2d_variant(const segmi_conv_desc* d, int op, char* buf, size_t len);
int segmi_filter_krsc_to_crsk(const float* w_krsc, float* w_crsk, int K, int R, int S, int C, int Kpad,
                              segmi_stream_t stream);
/* The same transposition for n filters in ONE launch (a training step needs it once per convolution and step: the filters
 * change with every optimizer step).  `table_dev` is a DEVICE array of n entries; entry i's tile_begin is the sum of
 * segmi_filter_tx_tiles() of the entries before it, total_tiles the sum over all of them.  Replaces the same
 * aten::cudnn_convolution_backward_input filter handling as segmi_filter_krsc_to_crsk. */
typedef struct segmi_filter_tx {
    const float* w_krsc;
    float* w_crsk;
    int K, R, S, C, Kpad;
    int tile_begin;
} segmi_filter_tx;
long segmi_filter_tx_tiles(int K, int R, int S, int C, int Kpad);
int segmi_filter_krsc_to_crsk_multi(const segmi_filter_tx* table_dev, int n, long total_tiles, segmi_stream_t stream);
/* Winograd F(2x2, 3x3) form of segmi_conv2d_fwd / segmi_conv2d_dgrad for 3x3 filters with stride 1 and pad == dil (the "same"
 * convolutions of models/resnet.py:84-86, models/pspnet.py:27-30, models/deeplabv3_plus.py:264-284,307-318, models/unet.py:15-18):
 * 2.25x fewer multiplications in the same arithmetic (transform constants 0, +-1, +-1/2), the 16 transform-domain contractions
 * as one batched launch of the implicit-GEMM kernel under the process-wide conv arithmetic; dilation d runs as d*d dense
 * sub-grids.  Same operands and layouts as the direct entry points (w_krsc for fwd, w_crsk = segmi_filter_krsc_to_crsk for
 * dgrad); `workspace` holds the transformed filter, input and product (segmi_conv2d_winograd_workspace bytes, 16-byte aligned).
 * op: 0 fwd, 1 dgrad.  _ok() says whether the pass is supported for this problem; the caller decides whether it pays. */
int segmi_conv2d_winograd_ok(const segmi_conv_desc* d, int op);
size_t segmi_conv2d_winograd_workspace(const segmi_conv_desc* d, int op);
int segmi_conv2d_winograd_fwd(const segmi_conv_desc* d, const float* x, const float* w_krsc, const float* bias, float* y,
                              int accumulate, float* v_keep, float* stats_partials, void* workspace, size_t workspace_bytes,
                              segmi_stream_t stream);
/* stats_partials (optional; needs accumulate == 0 and K % 4 == 0): the output transform also emits the BN-statistics partials
 * of y — segmi_conv2d_winograd_fwd_stats_parts(d) blocks of {count, mean, M2}[K] floats, the layout segmi_conv2d_fwd_stats
 * writes and segmi_bn_finalize_from_parts / segmi_bn_stats_from_parts consume — so the following BatchNorm never reads y for
 * its statistics.  _parts() returns 0 when the problem has no such epilogue. */
int segmi_conv2d_winograd_fwd_stats_parts(const segmi_conv_desc* d);
/* v_keep (optional, segmi_conv2d_winograd_v_bytes(d) bytes, 16-byte aligned, caller-owned): the forward pass writes its
 * transformed input V = B^T x B there instead of into the workspace, so that segmi_conv2d_winograd_wgrad(v_kept = that buffer)
 * contracts it again without re-reading x and re-writing 4x its size (memory for traffic: 288 GB of HBM per GPU). */
size_t segmi_conv2d_winograd_v_bytes(const segmi_conv_desc* d);
int segmi_conv2d_winograd_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                                void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_conv2d_winograd_variant(const segmi_conv_desc* d, int op, char* buf, size_t len);
/* Filter gradient of the same layers in the Winograd domain: dw = G^T [ sum_tiles (A dy A^T) (.) (B^T x B) ] G, the 16
 * contractions over the tiles as ONE batched launch of the direct filter-gradient kernel (1x1 problems, one deterministic
 * pixel split planned for all 16); the split reduction is folded into the G^T . G pass.  Replaces the wgrad half of
 * convolution_backward at the sites listed above; same operands as segmi_conv2d_wgrad.  `workspace` holds both transformed
 * operands and the partial sums [nsplit][16][K][C] (nsplit = the "splitk=" of segmi_conv2d_winograd_wgrad_variant). */
int segmi_conv2d_winograd_wgrad_ok(const segmi_conv_desc* d);
size_t segmi_conv2d_winograd_wgrad_workspace(const segmi_conv_desc* d);
int segmi_conv2d_winograd_wgrad(const segmi_conv_desc* d, const float* x, const float* v_kept, const float* dy, float* dw_krsc,
                                void* workspace, size_t workspace_bytes, segmi_stream_t stream);   /* x may be NULL when v_kept is given */
int segmi_conv2d_winograd_wgrad_variant(const segmi_conv_desc* d, char* buf, size_t len);
/* Measurement hooks for the three Winograd passes (bench.py's roofline leg, segmi/profile.py): _tiles() = number of 2x2 output
 * tiles T (the 16 contractions are [T x Cin] x [Cin x Cout], i.e. 32*T*Cin*Cout executed FLOPs); _trace() hands over two
 * caller-owned hipEvent_t that the NEXT Winograd call of this thread records on its stream right before / after its contraction
 * launch (one-shot; NULL, NULL cancels), so the MFMA-bound kernel is timed apart from the HBM-bound transforms around it. */
long segmi_conv2d_winograd_tiles(const segmi_conv_desc* d);
int segmi_conv2d_winograd_trace(void* ev_begin, void* ev_end);
/* The three convolution passes above compute on v_mfma_f32_32x32x2_f32 — fp32 products, fp32 accumulation, the reference's
 * arithmetic (trainer.py:56).  (ABI v9 removed the second arithmetic of v3-v8, fp32 products as three-plane bf16 splits on the bf16
 * matrix pipe — segmi_conv_set_math / segmi_conv_get_math / segmi_*_presplit: DESIGN.md section 4.3 has the measurements and why
 * it was retired.) */
/* db[k] = sum over rows of dy[row,k]  (classifier biases: models/pspnet.py:61,69; models/unet.py:37,77) */
size_t segmi_colsum_workspace(long rows, int C);
int segmi_colsum(const float* dy, int ld, long rows, int C, float* out, void* workspace, size_t workspace_bytes,
                 segmi_stream_t stream);

/* ------------------------------------------------------------------ depthwise convolution (K2)
 * nn.Conv2d(C, C, 3, groups=C) of Xception's SeparableConv2d (models/deeplabv3_plus.py:80).  Uses
 * segmi_conv_desc with K == C; the filter is [R,S,C] (tap-major, channels contiguous), R*S <= 9.
 * HBM-bound streaming kernels (0.75 % of Xception's MACs): deliberately not a GEMM. */
int segmi_dwconv2d_fwd(const segmi_conv_desc* d, const float* x, const float* w_rsc, float* y, segmi_stream_t stream);
/* The same launch with the BN-statistics epilogue (the BatchNorm between the depthwise and the pointwise convolution of
 * SeparableConv2d, models/deeplabv3_plus.py:76-86): segmi_dwconv2d_fwd_stats_parts(d) blocks of {count, mean, M2}[C] floats in the
 * layout of segmi_conv2d_fwd_stats (possibly > 512 partials: segmi_bn_finalize_from_parts merges in two levels); y is
 * bit-identical to segmi_dwconv2d_fwd.  stats_partials 16-byte aligned. */
int segmi_dwconv2d_fwd_stats_parts(const segmi_conv_desc* d);
int segmi_dwconv2d_fwd_stats(const segmi_conv_desc* d, const float* x, const float* w_rsc, float* y, float* stats_partials,
                             segmi_stream_t stream);
/* The same convolution reading the PRE-NORMALISATION tensor z of the BatchNorm(+ReLU) in front of it (3x3, stride 1, pad == dil in
 * {1, 2}: segmi_dwconv2d_pre_ok): every tap is max(fmaf(z, pre_scale[c], pre_shift[c]), 0) (ReLU when pre_relu) evaluated on the loaded
 * value — segmi_bn_apply's own expression, zero padding applied AFTER it — so the result equals segmi_bn_apply followed by
 * segmi_dwconv2d_fwd(_stats) bit for bit, and the normalised tensor between a pointwise convolution's BatchNorm and the next depthwise
 * layer (models/deeplabv3_plus.py:99-119: ReLU -> SeparableConv2d inside Block.rep) is never written or read.  stats_partials may be
 * NULL.  segmi_dwconv2d_wgrad_pre is the filter gradient with the same fused load of its x operand. */
int segmi_dwconv2d_pre_ok(const segmi_conv_desc* d);
int segmi_dwconv2d_fwd_pre(const segmi_conv_desc* d, const float* z, const float* pre_scale, const float* pre_shift, int pre_relu,
                           const float* w_rsc, float* y, float* stats_partials, segmi_stream_t stream);
int segmi_dwconv2d_wgrad_pre(const segmi_conv_desc* d, const float* z, const float* pre_scale, const float* pre_shift, int pre_relu,
                             const float* dy, float* dw_rsc, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_dwconv2d_dgrad(const segmi_conv_desc* d, const float* dy, const float* w_rsc, float* dx, segmi_stream_t stream);
size_t segmi_dwconv2d_wgrad_workspace(const segmi_conv_desc* d);
int segmi_dwconv2d_wgrad(const segmi_conv_desc* d, const float* x, const float* dy, float* dw_rsc, void* workspace,
                         size_t workspace_bytes, segmi_stream_t stream);

/* ------------------------------------------------------------------ 2x2 pixel shuffles (K3)
 * nn.ConvTranspose2d(Cin, K, kernel_size=2, stride=2) (models/unet.py:37) == 1x1 convolution with 4K
 * outputs (column k*4 + r*2 + s, weight W[c,k,r,s]) followed by depth_to_space (+ bias[k]); its backward
 * begins with space_to_depth of dy.  src/dst NHWC; H, W are the LOW-resolution sizes; K % 4 == 0. */
int segmi_depth_to_space2(const float* src, int ld_src, float* dst, int ld_dst, const float* bias, int N, int H, int W,
                          int K, segmi_stream_t stream);
int segmi_space_to_depth2(const float* src, int ld_src, float* dst, int ld_dst, int N, int H, int W, int K,
                          segmi_stream_t stream);

/* ------------------------------------------------------------------ batch norm (K4/K5/K6)
 * Replaces aten::native_batch_norm(+_backward), relu_/threshold_backward and the residual add_:
 * every nn.BatchNorm2d call site (61 in PSPNet-R50), F.batch_norm fallback
 * utils/sync_batchnorm/batchnorm.py:65-68, nn.ReLU(inplace=True), models/resnet.py:118.
 * stats: per-channel (count, mean, M2) by Welford + Chan merge (matches two-pass F.batch_norm);
 * SyncBN all-reduces the packed partial [count, mean, M2] between the two calls. */
size_t segmi_bn_stats_workspace(long rows, int C);
/* partial[3*C] = {count (replicated per channel), mean, M2} of x[rows, C] */
int segmi_bn_stats(const float* x, int ld, long rows, int C, float* partial, void* workspace, size_t workspace_bytes,
                   segmi_stream_t stream);
/* Merge `nparts` packed partials (e.g. one per rank after all-gather; nparts = 1 locally) and
 * produce mean/invstd, scale = gamma*invstd, shift = beta - mean*scale; update running stats
 * in place with `momentum` (unbiased var) when running_mean != NULL, and bump
 * *num_batches_tracked (nn.BatchNorm2d bookkeeping) when that pointer is non-NULL.
 * clamp_mode = 0: invstd = 1/sqrt(var+eps) (nn.BatchNorm2d);  1: clamp(var,eps)^-0.5
 * (utils/sync_batchnorm/batchnorm.py:145). */
int segmi_bn_finalize(const float* partials, int nparts, int C, const float* gamma, const float* beta, float eps,
                      float momentum, int clamp_mode, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                      float* count_out, segmi_stream_t stream);
/* Single-device batch statistics in one call: segmi_bn_stats followed by segmi_bn_finalize(nparts = 1) with the merge
 * and the finalize fused into one launch (bit-identical to the two-call sequence; workspace as segmi_bn_stats). */
int segmi_bn_stats_finalize(const float* x, int ld, long rows, int C, const float* gamma, const float* beta, float eps,
                            float momentum, int clamp_mode, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                            void* workspace, size_t workspace_bytes, segmi_stream_t stream);
/* The same two results from Welford partials a PRODUCER already wrote (segmi_conv2d_fwd_stats: [nparts][3][C] floats):
 * _stats_from_parts = segmi_bn_stats without reading x (one packed partial for the SyncBN all-gather), _finalize_from_parts =
 * segmi_bn_stats_finalize without reading x.  More than 512 partials are merged in two levels through `workspace`
 * (segmi_bn_parts_workspace bytes). */
size_t segmi_bn_parts_workspace(int nparts, int C);
int segmi_bn_stats_from_parts(const float* partials, int nparts, int C, float* partial, void* workspace, size_t workspace_bytes,
                              segmi_stream_t stream);
int segmi_bn_finalize_from_parts(const float* partials, int nparts, int C, const float* gamma, const float* beta, float eps,
                                 float momentum, int clamp_mode, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                                 void* workspace, size_t workspace_bytes, segmi_stream_t stream);
/* eval / frozen BN: scale/shift from running statistics */
int segmi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                         float eps, int C, float* mean, float* invstd, float* scale, float* shift,
                         segmi_stream_t stream);
/* y = x*scale + shift (+ residual) ; relu optional */
int segmi_bn_apply(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy, long rows, int C,
                   const float* scale, const float* shift, int relu, segmi_stream_t stream);
/* sums[2*C] = {sum dy', sum dy'*xhat},  dy' = relu ? dy*[y>0] : dy  */
size_t segmi_bn_bwd_reduce_workspace(long rows, int C);
/* ReLU mask source: y (the saved output) when given; y == NULL (allowed when no residual was added) recomputes
 * fmaf(x, scale, shift) > 0 exactly as bn_apply evaluated it, saving one full read of y per pass.
 * Two launches: row partials in fp64, then their sum (a one-launch form with ticket counters was measured slower and removed in ABI v9). */
int segmi_bn_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, long rows, int C,
                        const float* mean, const float* invstd, const float* scale, const float* shift, int relu,
                        float* sums, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
/* dgamma = sums[C:2C], dbeta = sums[0:C];
 * training: dx = scale*(dy' - sums0/count - xhat*sums1/count); eval (frozen): dx = scale*dy'.
 * d_residual (optional) = dy'.  `count` is the (global) element count per channel; when count_dev != NULL it is read
 * from device memory instead (SyncBN: the count segmi_bn_finalize summed from the all-gathered partials — no host-side
 * exchange, valid for ragged shards). */
int segmi_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, long rows, int C,
                       const float* mean, const float* invstd, const float* scale, const float* shift, const float* sums,
                       float count, const float* count_dev, int relu, int training, float* dx, int lddx, float* dres,
                       int lddres, segmi_stream_t stream);
/* standalone ReLU (models/deeplabv3_plus.py:99-101,210) */
int segmi_relu_fwd(const float* x, int ldx, float* y, int ldy, long rows, int C, segmi_stream_t stream);
int segmi_relu_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, long rows, int C,
                   segmi_stream_t stream);
/* out = a + b (residual sum of Xception blocks models/deeplabv3_plus.py:131; gradient accumulation) */
int segmi_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long rows, int C,
              segmi_stream_t stream);

/* ------------------------------------------------------------------ pooling / resize (K7/K8/K9) */
/* aten::max_pool2d_with_indices (+bwd): models/resnet.py:151, models/deeplabv3_plus.py:24,
 * models/unet.py:27.  idx holds the winning tap (r*k+s) per element, 1 byte each. */
int segmi_maxpool_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int N, int H, int W, int C, int P,
                      int Q, int k, int stride, int pad, segmi_stream_t stream);
int segmi_maxpool_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int N, int H, int W, int C,
                      int P, int Q, int k, int stride, int pad, segmi_stream_t stream);
/* aten::adaptive_avg_pool2d (+bwd): models/pspnet.py:26 (bins 1,2,3,6), models/deeplabv3_plus.py:274 */
int segmi_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH, int OW,
                               segmi_stream_t stream);
int segmi_adaptive_avgpool_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int OH,
                               int OW, int accumulate, segmi_stream_t stream);
/* Fused pyramid pooling: the <= 4 AdaptiveAvgPool2d(b) of the PSP module (models/pspnet.py:25-37, bins 1,2,3,6) over ONE
 * read of x (cell sums over the union of all window boundaries, then assembly), and their backward as ONE write of dx from
 * the small dy tensors (instead of four full-size gradient maps that autograd adds).  `bins`, `y`/`dy`, `ldy` are HOST
 * arrays of nlevels entries (device pointers inside); bins[l] <= 8.  Both directions use a workspace of
 * segmi_pyramid_pool_workspace bytes (per-cell sums forward, per-cell gradients backward), 16-byte aligned. */
size_t segmi_pyramid_pool_workspace(int N, int H, int W, int C, int nlevels, const int* bins);
int segmi_pyramid_pool_fwd(const float* x, int ldx, int N, int H, int W, int C, int nlevels, const int* bins, float* const* y,
                           const int* ldy, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_pyramid_pool_bwd(const float* const* dy, const int* lddy, float* dx, int lddx, int N, int H, int W, int C,
                           int nlevels, const int* bins, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
/* aten::upsample_bilinear2d (+bwd), align_corners in {0,1}: models/pspnet.py:35-36,86,91;
 * models/deeplabv3_plus.py:291,328,361; models/unet.py:46-47.  bwd is the exact transpose in
 * gather form (deterministic, no atomics), evaluated separably (width pass into the workspace, then height pass). */
int segmi_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH, int OW,
                       int align_corners, segmi_stream_t stream);
size_t segmi_bilinear_bwd_workspace(int N, int H, int W, int C, int OH, int OW);
int segmi_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int OH, int OW,
                       int align_corners, void* workspace, size_t workspace_bytes, segmi_stream_t stream);

/* ------------------------------------------------------------------ dropout (K14)
 * nn.Dropout2d(0.1) models/pspnet.py:22,68 (per (n,c) mask) and nn.Dropout models/deeplabv3_plus.py:282,318
 * (per element).  The mask is a counter-based hash of (seed, index): regenerated in backward.
 * seed_epoch_dev (nullable, device memory): a step counter folded into the seed on the device — a training step captured
 * once into a hipGraph then draws fresh masks on every replay (the by-value seed is frozen at capture). */
int segmi_dropout(const float* x, int ldx, float* y, int ldy, int N, long HW, int C, float p, int channelwise,
                  uint64_t seed, const uint64_t* seed_epoch_dev, segmi_stream_t stream);

/* ------------------------------------------------------------------ factored PSP bottleneck (models/pspnet.py:25-38) */
/* conv3x3(cat[features, up(p_1), ..., up(p_L)]) = conv3x3(features; W[:, :Cx]) + sum_l sum_taps B * T_l with
 * T_l = p_l (x) W[:, slice_l] a small GEMM (csrc/pyramid_bottleneck.hip): the upsampled pyramid branches and the concat buffer
 * are never built and half of the bottleneck's MACs disappear.  Pieces:
 *   segmi_filter_slice / _unslice : channel slice [c0, c0+Cs) of a KRSC filter [K, RS, Ctot] as a contiguous [K*RS, Cs] matrix
 *                                   with rows ordered (k, rs) (rs_major = 0: itself a KRSC filter) or (rs, k) (rs_major = 1:
 *                                   the 1x1 filter whose output channel rs*K + k is T's layout); unslice scatters a gradient
 *                                   slice back into the full KRSC gradient.
 *   segmi_pyramid_up_fwd          : y[n,h,w,k] = sum_l sum_{r,s} bilinear(align_corners=True, bh_l x bw_l -> H x W) of
 *                                   T_l[n, :, :, (r*3+s)*K + k] evaluated at (h+r-1, w+s-1), zero outside the map (the 3x3
 *                                   convolution's padding).  T_l: [N, bh_l, bw_l, 9K] dense.  y is OVERWRITTEN (pixel stride ldy).
 *                                   (Also the DeepLab decoder's upsample + concat + 3x3, models/deeplabv3_plus.py:323-330: one
 *                                   "level" of 33 x 33 nodes.)
 *   segmi_pyramid_up_bwd          : the transpose: G_l (same layout as T_l) from dy [N, H, W, K] (pixel stride lddy).
 * Workspace (both directions): segmi_pyramid_up_workspace, 16-byte aligned.  K % 4 == 0, at most 4 levels. */
int segmi_filter_slice(const float* w_krsc, int K, int RS, int Ctot, int c0, int Cs, int rs_major, float* out, segmi_stream_t stream);
int segmi_filter_unslice(const float* grad_slice, int K, int RS, int Ctot, int c0, int Cs, int rs_major, float* dw_krsc,
                         segmi_stream_t stream);
size_t segmi_pyramid_up_workspace(int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w);
int segmi_pyramid_up_fwd(const float* const* T, int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w, float* y,
                         int ldy, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_pyramid_up_bwd(const float* dy, int lddy, int N, int H, int W, int K, int nlevels, const int* bins_h, const int* bins_w,
                         float* const* G, void* workspace, size_t workspace_bytes, segmi_stream_t stream);

/* ------------------------------------------------------------------ RCCL exchange steps (comm_{init,allreduce_async,wait})
 * One process per GPU over xGMI.  Replaces nn.DataParallel's gather / reduce_add onto GPU 0 (base/base_trainer.py:33-38) and the
 * SyncBN master/slave exchange (utils/sync_batchnorm/batchnorm.py:105-126: torch.cuda.comm.reduce_add / broadcast_coalesced at
 * :117,120; comm.py:102-133) for hosts that do not use torch.distributed (the Python drop-in's default transport is
 * torch.distributed "nccl" = the same RCCL; SEGMI_COMM=abi routes its gradient buckets through these entry points instead).
 * RCCL is bound at run time (dlopen; segmi_comm_available() == 0 when no librccl can be found): no link-time dependency.
 *   rank 0: segmi_comm_get_unique_id -> ship the segmi_comm_unique_id_bytes() bytes to every rank by any means -> all ranks:
 *   segmi_comm_init(&c, world, rank, id) on their current HIP device.
 *   segmi_comm_allreduce_async / _allgather_async: the communicator's side stream waits for everything enqueued on `stream` so
 *   far (the buffer is complete), the collective is enqueued on the side stream, an event OF THIS CALL is recorded and its
 *   ticket (>= 1, increasing) returned through ticket_out (may be NULL).  segmi_comm_wait_ticket(c, t, s) makes stream s wait for
 *   exactly that collective (bucket i's optimizer step can start while buckets i+1... are in flight); segmi_comm_wait(c, s)
 *   for the communicator's LAST collective.  No host sync.  A communicator belongs to the HIP device that was current at
 *   segmi_comm_init: calls made with another current device return SEGMI_ERR_BADARG. */
typedef struct segmi_comm segmi_comm;
int segmi_comm_available(void);
int segmi_comm_unique_id_bytes(void);
int segmi_comm_get_unique_id(void* id_out, size_t bytes);
int segmi_comm_init(segmi_comm** comm_out, int world, int rank, const void* unique_id, size_t bytes);
int segmi_comm_world(const segmi_comm* comm);
int segmi_comm_allreduce_async(segmi_comm* comm, const float* send, float* recv, size_t count, int average, long* ticket_out,
                               segmi_stream_t stream);
int segmi_comm_allgather_async(segmi_comm* comm, const float* send, float* recv, size_t count_per_rank, long* ticket_out,
                               segmi_stream_t stream);
int segmi_comm_wait_ticket(segmi_comm* comm, long ticket, segmi_stream_t stream);
int segmi_comm_wait(segmi_comm* comm, segmi_stream_t stream);
int segmi_comm_destroy(segmi_comm* comm);

/* ------------------------------------------------------------------ training-time augmentation */
/* The cv2 / PIL sequence of BaseDataSet._augmentation, _val_augmentation and __getitem__ (base/base_dataset.py:40-136) on the
 * device, one call per stage and sample; images are uint8 HWC (3 channels), labels int32 HW, all device memory; the random
 * decisions are the caller's (dataloaders/gpu_augment.py draws them in the reference's order).  The pixel arithmetic is
 * OpenCV's FIXED-POINT arithmetic for 8-bit images, in integers; what OpenCV derives in double / float per output row and
 * column arrives as int32 device tables the caller computes the same way (layouts below):
 *   segmi_aug_resize : cv2.resize INTER_LINEAR (image; 11-bit coefficients, or the 2x2 box average when area2x: both scales are
 *                      exactly 2) / nearest label to dst_h x dst_w.  tables_dev = xs[dw] | xa[dw] | ys[dh] | yb[dh] | lx[dw] | ly[dh]:
 *                      source column and packed coefficients (c0 | c1 << 16) per output column, the same per output row (rows are
 *                      clipped by the kernel), source column / row of the label (cv2 INTER_NEAREST or PIL NEAREST)   (:48-50,71-72)
 *   segmi_aug_rotate : cv2.warpAffine, bilinear (1/32-pixel grid, 15-bit weights) / nearest, constant border 0.
 *                      tables_dev = adelta[w] | bdelta[w] | X0[h] | Y0[h]: the inverted getRotationMatrix2D((w/2, h/2), angle, 1.0)
 *                      scaled by 2^10 as cv::warpAffine rounds it, without the interpolation's rounding offset         (:76-81)
 *   segmi_aug_blur   : cv2.GaussianBlur(3x3, sigma) on CV_8U, BORDER_REFLECT_101: taps {m0, m1, m0} in 8.8 fixed point
 *                      (2*m0 + m1 == 256); scratch = 3*h*w uint16                                                    (:113-117)
 *   segmi_aug_finish : zero padding at the bottom / right up to the crop, crop at (start_h, start_w), optional fliplr,
 *                      ToTensor + Normalize(mean, std) into the fp32 NHWC batch slot `out` (pixel stride ld >= 4, channels
 *                      3..ld-1 zeroed) and the int64 label slot                                                     (:84-110,129-136) */
int segmi_aug_resize(const uint8_t* image, const int32_t* label, int src_h, int src_w, uint8_t* out_image, int32_t* out_label,
                     int dst_h, int dst_w, const int32_t* tables_dev, int area2x, segmi_stream_t stream);
int segmi_aug_rotate(const uint8_t* image, const int32_t* label, int h, int w, const int32_t* tables_dev, uint8_t* out_image,
                     int32_t* out_label, segmi_stream_t stream);
int segmi_aug_blur(const uint8_t* image, int h, int w, int m0, int m1, uint16_t* scratch, uint8_t* out_image, segmi_stream_t stream);
int segmi_aug_finish(const uint8_t* image, const int32_t* label, int h, int w, int crop_h, int crop_w, int start_h, int start_w,
                     int flip, const float* mean3, const float* std3, float* out, int ld, int64_t* out_label, segmi_stream_t stream);

/* ------------------------------------------------------------------ per-pixel losses (K10/K11/K12) */
/* CrossEntropyLoss2d (utils/losses.py:24-31): nn.CrossEntropyLoss(weight, ignore_index, reduction='mean'):
 * sum_i w[t_i] * (-log_softmax_i[t_i]) / sum_i w[t_i] over pixels with target != ignore_index; class_weight (C floats,
 * device) may be NULL (w = 1: the plain mean over valid pixels).
 * fwd writes lse[rows] and loss_out[3] = {loss, denominator = sum of w (= n_valid without weights), numerator};
 * bwd recomputes softmax from (logits, lse): dlogits = w[t] * (softmax - onehot) * (*grad_out) / loss_out[1], 0 at ignored
 * pixels.  A caller that wants another normalisation (reduction='sum'; the global-batch mean of a data-parallel job, where
 * the denominators of all ranks are all-reduced first) hands bwd its own loss_out[1]. */
size_t segmi_ce_workspace(long rows);
int segmi_ce_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index,
                 const float* class_weight, float* lse, float* loss_out, void* workspace, size_t workspace_bytes,
                 segmi_stream_t stream);
int segmi_ce_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C,
                 long ignore_index, const float* class_weight, const float* loss_out, const float* grad_out, float* dlogits,
                 int lddl, segmi_stream_t stream);

/* CrossEntropyLoss2d applied to bilinearly upsampled logits — F.interpolate(logits_lo, size=(OH, OW), mode='bilinear',
 * align_corners) followed by the loss (models/pspnet.py:85-91 / models/deeplabv3_plus.py:361 + trainer.py:56-66) — WITHOUT
 * materialising the [N, C, OH, OW] tensor: fwd interpolates each output pixel's four low-resolution neighbours on the fly and
 * keeps only lse[N*OH*OW]; bwd recomputes the softmax the same way while reducing along the width, so the gradient is produced
 * directly for logits_lo [N, H, W, C] (pixel stride ld / lddl).  loss_out, class_weight, grad_out as segmi_ce_fwd / _bwd.
 * Workspace: segmi_upsample_ce_workspace (16-byte aligned). */
size_t segmi_upsample_ce_workspace(int N, int H, int W, int C, int OH, int OW);
int segmi_upsample_ce_fwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                          const int64_t* target, long ignore_index, const float* class_weight, float* lse, float* loss_out,
                          void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_upsample_ce_bwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                          const int64_t* target, const float* lse, long ignore_index, const float* class_weight,
                          const float* loss_out, const float* grad_out, float* dlogits_lo, int lddl, void* workspace,
                          size_t workspace_bytes, segmi_stream_t stream);

/* DiceLoss (utils/losses.py:33-50): softmax, one-hot, whole-batch 1 - (2*sum(p*y)+s)/(sum(p)+sum(y)+s).
 * Reproduces the reference's in-place rewrite of ignored pixels to target.min() (target is mutated when
 * ignore_index is not in range(target.min(), target.max()) and an ignored pixel exists); stats[4] =
 * {tmin, tmax, n_ignored, remapped} are computed on the device.  loss_out[4] = {loss, sum p*y, denominator, -}. */
size_t segmi_dice_workspace(long rows);
int segmi_dice_fwd(const float* logits, int ld, int64_t* target, long rows, int C, long ignore_index, float smooth,
                   int64_t* stats, float* lse, float* loss_out, void* workspace, size_t workspace_bytes,
                   segmi_stream_t stream);
int segmi_dice_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C,
                   const float* loss_out, const float* grad_out, float* dlogits, int lddl, segmi_stream_t stream);
/* The three stages of segmi_dice_fwd as separate calls, for data-parallel training: the reference evaluates DiceLoss on the
 * GATHERED global batch (trainer.py:56-66 under nn.DataParallel), so target.min()/max()/ignored-count and the three sums are
 * whole-job quantities.  A rank runs segmi_target_stats, all-reduces {min, max, count} (and re-derives stats[3]), runs
 * segmi_dice_sums with those global stats (performs the in-place target rewrite, writes lse and this rank's
 * sums[3] = {sum p*y, sum p, sum y} as doubles), all-reduces sums, and calls segmi_dice_finalize.  Workspaces as
 * segmi_dice_workspace(rows). */
int segmi_target_stats(const int64_t* target, long rows, long ignore_index, int64_t* stats, void* workspace,
                       size_t workspace_bytes, segmi_stream_t stream);
int segmi_dice_sums(const float* logits, int ld, int64_t* target, long rows, int C, long ignore_index, const int64_t* stats,
                    float* lse, double* sums, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_dice_finalize(const double* sums, float smooth, float* loss_out, segmi_stream_t stream);
/* FocalLoss (utils/losses.py:52-65): mean over ALL pixels of (1-exp(-ce))^gamma * ce with ce = alpha[t] * (-log p_t)
 * (alpha: C class weights of the inner nn.CrossEntropyLoss(reduce=False, weight=alpha), NULL = 1) and ce = 0 at ignored
 * pixels.  Workspace: segmi_ce_workspace(rows).  loss_out[3] = {loss, rows, sum}; bwd divides by loss_out[1]. */
int segmi_focal_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float gamma,
                    const float* alpha, float* lse, float* loss_out, void* workspace, size_t workspace_bytes,
                    segmi_stream_t stream);
int segmi_focal_bwd(const float* logits, int ld, const int64_t* target, const float* lse, long rows, int C,
                    long ignore_index, float gamma, const float* alpha, const float* loss_out, const float* grad_out,
                    float* dlogits, int lddl, segmi_stream_t stream);

/* LovaszSoftmax (utils/losses.py:79-89 -> utils/lovasz_losses.py:153-218, lovasz_grad :19-31), classes='present',
 * per_image=False: softmax; ignored pixels dropped; per present class sort |fg - p_c| descending, Jaccard-gradient dot;
 * mean over present classes.  The C per-class sorts run as ONE segmented radix sort over the elements that can matter: an
 * element ranked after its class's last foreground element has Jaccard difference exactly 0 (lovasz_grad :19-31), so only
 * elements with error >= the class's smallest foreground error need sorting.  The "survivors" are the foreground elements plus
 * the background elements with z - lse >= xthr[c] = log(that error) minus a rounding margin — a slight superset of that prefix of
 * the full order, selected without evaluating exp; the extra elements rank where the Jaccard difference is 0, so loss and gradient
 * are bit-identical to the full sort.
 * G (rows*ldg floats, ldg >= round_up(C,4), pixel-major) receives d loss_c / d p (un-normalised) for SURVIVOR entries only; it
 * is neither cleared nor read elsewhere: the backward re-derives the survivor set from logits, lse, target and the thresholds.
 * loss_out[4 + C] = {loss, n_present, survivors, n_present * n_valid (= keys of the full sort), xthr[C] (keep thresholds on z - lse)}
 * must reach segmi_lovasz_bwd unchanged together with lse and G.  At most 1820 classes.
 * rows < 2^24 (the reference's fp32 cumsums are exact only below that) and log2(C) + log2(rows) <= 32.  workspace must be
 * 256-byte aligned; its size covers the worst case (every element survives). */
/* Process-wide switch, 1 (default) = tail pruning as above, 0 = every valid pixel of every present class is sorted (A/B and the
 * bit-identity reference of the tests).  Also SEGMI_LOVASZ_PRUNE=0 at first use. */
int segmi_lovasz_set_prune(int on);
size_t segmi_lovasz_workspace(long rows, int C);
int segmi_lovasz_fwd(const float* logits, int ld, const int64_t* target, long rows, int C, long ignore_index, float* lse,
                     float* G, int ldg, float* loss_out, void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_lovasz_bwd(const float* logits, int ld, const int64_t* target, long ignore_index, const float* lse, const float* G,
                     int ldg, long rows, int C, const float* loss_out, const float* grad_out, float* dlogits, int lddl,
                     segmi_stream_t stream);
/* The same loss on bilinearly UPSAMPLED logits without materialising them: the reference feeds F.interpolate(low-resolution
 * logits, input size, bilinear, align_corners=True) to the loss (models/deeplabv3_plus.py:361, models/pspnet.py:85-91 ->
 * trainer.py:56-66 -> utils/losses.py:86-89).  logits_lo [N, H, W, C] (row stride ld); target / lse [N, OH, OW]; G
 * [N*OH*OW, ldg] and loss_out as above.  Every pass interpolates its pixel's four low-resolution neighbours on the fly with
 * the operation order of segmi_bilinear_fwd, so loss, lse, G and dlogits_lo are BIT-IDENTICAL to segmi_bilinear_fwd ->
 * segmi_lovasz_fwd / segmi_lovasz_bwd -> segmi_bilinear_bwd; the backward reduces along the width while it evaluates the
 * gradient, so d loss / d logits exists only at [N, OH, W] and [N, H, W].  N*OH*OW < 2^24.  One workspace size serves both
 * calls (256-byte aligned). */
size_t segmi_upsample_lovasz_workspace(int N, int H, int W, int C, int OH, int OW);
int segmi_upsample_lovasz_fwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                              const int64_t* target, long ignore_index, float* lse, float* G, int ldg, float* loss_out,
                              void* workspace, size_t workspace_bytes, segmi_stream_t stream);
int segmi_upsample_lovasz_bwd(const float* logits_lo, int ld, int N, int H, int W, int C, int OH, int OW, int align_corners,
                              const int64_t* target, long ignore_index, const float* lse, const float* G, int ldg,
                              const float* loss_out, const float* grad_out, float* dlogits_lo, int lddl, void* workspace,
                              size_t workspace_bytes, segmi_stream_t stream);

/* eval_metrics (utils/metrics.py:42-67; trainer.py:84-86,128-129): argmax (first maximal class) + pixel-accuracy counts +
 * per-class intersection / prediction / label areas, ACCUMULATED into acc[2 + 3*C] int64 =
 * {correct, labeled, inter[C], pred_area[C], label_area[C]} (union = pred_area + label_area - inter).  The caller zeroes acc
 * at the start of an epoch and reads it when it wants numbers: no per-iteration host synchronisation. */
int segmi_seg_metrics(const float* logits, int ld, const int64_t* target, long rows, int C, int64_t* acc, segmi_stream_t stream);

/* ------------------------------------------------------------------ fused SGD
 * torch.optim.SGD.step as the reference configures it (config.json:48-52; differential learning rates
 * base/base_trainer.py:46-57): g = grad + wd*p; buf = momentum*buf + g; p -= lr*buf, over every tensor in ONE launch.
 * `table` is a DEVICE array of chunks the caller builds once (a tensor is cut into ranges of segmi_sgd_chunk_elems()
 * elements; `vec4` = 1 when the three pointers are 16-byte aligned and count % 4 == 0); per-group hyper-parameters are HOST
 * arrays read at call time (schedulers change them every iteration; ngroups <= 8).  Momentum buffers start zeroed. */
typedef struct segmi_sgd_chunk {
    float* param;
    const float* grad;
    float* momentum;
    long count;
    int group;
    int vec4;
} segmi_sgd_chunk;
int segmi_sgd_chunk_elems(void);
int segmi_sgd_step(const segmi_sgd_chunk* table, int nchunks, const float* lr_host, const float* weight_decay_host,
                   const float* momentum_host, int ngroups, segmi_stream_t stream);
/* Same update with the hyper-parameters in DEVICE memory: `hyper_dev` holds segmi_sgd_hyper_floats() floats,
 * lr[8] | weight_decay[8] | momentum[8] indexed by group.  For steps captured into a hipGraph (kernel arguments are frozen
 * at capture; a captured copy from a pinned host buffer refreshes hyper_dev on every replay). */
int segmi_sgd_hyper_floats(void);
int segmi_sgd_step_dev(const segmi_sgd_chunk* table, int nchunks, const float* hyper_dev, segmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGMI_H */
