// Entry points shared between translation units of libsegmi.so with C++ linkage — NOT part of the C ABI (include/segmi.h).
#pragma once
#include "segmi_common.h"

// D_b[M, Cd] (row stride ldd) = A_b[M, Cs] (row stride lda) x W_b[Cd, Cs]^T for b < batch, as ONE launch of the LDS-DMA
// implicit-GEMM kernel (conv_igemm.hip) under the process-wide convolution arithmetic; operands of consecutive problems lie
// bs_a / bs_w / bs_d floats apart.  Returns SEGMI_ERR_BADARG when the operands do not fit the kernel's 32-bit buffer offsets.
int segmi_internal_gemm_batched(const float* a, int lda, const float* w, float* d, int ldd, int M, int Cs, int Cd, int batch,
                                long bs_a, long bs_w, long bs_d, hipStream_t st);
// Name of the kernel variant segmi_internal_gemm_batched launches for this shape (as a rocprofv3 trace shows it).
int segmi_internal_gemm_variant(int M, int Cd, char* buf, size_t len);
// Whether segmi_internal_gemm_batched can run this shape (LDS-DMA kernels enabled, operands within 32-bit buffer offsets).
bool segmi_internal_gemm_ok(long M, int lda, int Cs, int Cd);
