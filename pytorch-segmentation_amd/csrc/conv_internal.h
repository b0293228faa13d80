// Entry points shared between translation units of libsegmi.so with C++ linkage — NOT part of the C ABI (include/segmi.h).
#pragma once
#include "segmi_common.h"

// D_b[M, Cd] (row stride ldd) = A_b[M, Cs] (row stride lda) x W_b[Cd, Cs]^T for b < batch, as ONE launch of the LDS-DMA
// implicit-GEMM kernel (conv_igemm.hip) under the process-wide convolution arithmetic; operands of consecutive problems lie
// bs_a / bs_w / bs_d floats apart.  Returns SEGMI_ERR_BADARG when the operands do not fit the kernel's 32-bit buffer offsets.
int segmi_internal_gemm_batched(const float* a, int lda, const float* w, float* d, int ldd, int M, int Cs, int Cd, int batch,
                                long bs_a, long bs_w, long bs_d, hipStream_t st);
// Name of the kernel variant segmi_internal_gemm_batched launches for this shape (as a rocprofv3 trace shows it).
int segmi_internal_gemm_variant(int M, int Cs, int Cd, char* buf, size_t len);
// Whether segmi_internal_gemm_batched can run this shape (LDS-DMA kernels enabled, operands within 32-bit buffer offsets).
bool segmi_internal_gemm_ok(long M, int lda, int Cs, int Cd);

// dW_b[K, C] = DY_b[M, K]^T x X_b[M, C] for b < batch: the batched 1x1 filter gradient (M % 32 == 0; x planes M*C floats apart,
// dy planes M*round_up(K,4) apart), ONE launch of the LDS-DMA filter-gradient kernel with blockIdx.z = b.  The pixel axis is
// split `segmi_internal_wgrad_batched_splits` ways (0: shape not supported) and the partial sums land in
// ws[nsplit][batch][K][C]; the caller reduces them in slice order.
int segmi_internal_wgrad_batched_splits(int M, int C, int K, int batch);
int segmi_internal_wgrad_batched(const float* x, const float* dy, float* ws, int M, int C, int K, int batch, hipStream_t st);
int segmi_internal_wgrad_batched_variant(int M, int C, int K, int batch, char* buf, size_t len);
