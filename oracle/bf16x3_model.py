"""TEST INFRASTRUCTURE — numpy model of the MATH_BF16X3 arithmetic of csrc/conv_igemm.hip (split_pair / mma_bf16x3).

Not a restatement of anything in the reference (which computes in fp32, trainer.py:56): it pins what the
kernel's three-plane split must guarantee so that the bf16 matrix pipe may stand in for the fp32 one:
  * h = bf16_rne(x), m = bf16_rne(x - h), l = bf16_rne(x - h - m) with fp32 subtractions  =>  h + m + l == x exactly;
  * the six products kept (l*h', h*l', m*m', m*h', h*m', h*h'), each exact in fp32 (8-bit x 8-bit significands), summed
    over k in fp32, differ from the exact dot product by about one fp32 rounding per product.
Only tests/ import this module; the product path never does.
"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite inputs)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    s = (r - m).astype(np.float32)
    l = bf16_rne(s)
    return h, m, l


def dot_bf16x3(a, b, kstep=16):
    """sum_k a[..., k] * b[..., k] the way mma_bf16x3 accumulates it: per 16-wide k step six matrix instructions
    (smallest plane products first), each adding its exactly-computed 16 products to the fp32 accumulator."""
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), dtype=np.float32)
    K = a.shape[-1]
    for k0 in range(0, K, kstep):
        sl = slice(k0, k0 + kstep)
        for pa, pb in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            # products of two bf16 values are exact in fp32; the instruction's internal sum is modelled in fp64 and
            # rounded once into the fp32 accumulator
            part = (pa[..., sl].astype(np.float64) * pb[..., sl].astype(np.float64)).sum(-1)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def dot_f32_chain(a, b):
    """The MATH_F32 path: v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fmaf chain."""
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), dtype=np.float32)
    for k in range(a.shape[-1]):
        acc = (acc.astype(np.float64) + a[..., k].astype(np.float64) * b[..., k].astype(np.float64)).astype(np.float32)
    return acc
