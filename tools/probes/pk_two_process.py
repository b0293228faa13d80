#!/usr/bin/env python
"""Second-generation reproducer for the round-2 observation (profiles/r02_two_process_repeatability.txt), with the REAL kernels:

  victim    : `segmi_bilinear_fwd` on the shapes of the failing test (logits [2, 5(ld 8), 9, 11] -> 72 x 88), looped, every output
              compared bit for bit with the first one.  Two builds of csrc/pool_resize.hip: the shipped one (-fno-slp-vectorize) and
              a copy built WITH the SLP vectoriser (the round-2 form: v_pk_mov_b32 / v_pk_mul_f32 ... op_sel / v_pk_fma_f32).
  aggressor : another PROCESS on the same GPU looping one dense convolution under bf16x3 (v_mfma_f32_32x32x16_bf16 + split VALU +
              LDS-DMA), under fp32 MFMA, or absent.

    python tools/probes/pk_two_process.py [--seconds 12]

The synthetic kernels of tools/probes/pk_mfma_repro.hip (a packed FMA stream next to a register-only MFMA loop) did NOT reproduce
the corruption; this one keeps everything that was present when it was seen."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "pytorch-segmentation_amd")
SLP_SO = "/tmp/libsegmi_pool_slp.so"


def build_slp_copy():
    src = os.path.join(PKG, "csrc", "pool_resize.hip")
    obj = "/tmp/pool_resize_slp.o"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-c", src, "-o", obj], check=True,
                   stderr=subprocess.DEVNULL)
    subprocess.run(["g++", "-shared", "-o", SLP_SO, obj, "-L/opt/rocm/lib", "-lamdhip64"], check=True)


def victim(variant, seconds):
    sys.path.insert(0, PKG)
    import torch
    import segmi  # noqa: F401  (loads the shipped library and torch's HIP runtime)
    from segmi._lib import lib as shipped
    lib = shipped
    if variant == "slp":
        lib = C.CDLL(SLP_SO, mode=os.RTLD_LOCAL)
        lib.segmi_bilinear_fwd.restype = C.c_int
        lib.segmi_bilinear_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 7 + [C.c_void_p]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    N, Cc, H, W, OH, OW = 2, 5, 9, 11, 72, 88
    x = torch.zeros(N, H, W, 8)
    x[..., :Cc] = torch.randn(N, H, W, Cc, generator=g) * 2
    x = x.to(dev)
    ys = [torch.empty(N, OH, OW, 8, device=dev) for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    # reference: the same interpolation by torch on the CPU in float64 (the kernel's fp32 result must be within rounding of it);
    # `first` = the first launch of the victim itself (the round-2 stress compared with that)
    ref = torch.nn.functional.interpolate(x[..., :Cc].permute(0, 3, 1, 2).double().cpu(), size=(OH, OW), mode="bilinear",
                                          align_corners=False).permute(0, 2, 3, 1).float().to(dev)
    first, bad, wrong, iters, t0 = None, 0, 0, 0, time.time()
    patterns = set()
    while time.time() - t0 < seconds:
        for _ in range(50):
            y = ys[iters & 1]
            y.zero_()
            rc = lib.segmi_bilinear_fwd(x.data_ptr(), 8, y.data_ptr(), 8, N, H, W, Cc, OH, OW, 0, st)
            assert rc == 0, rc
            err = (y[..., :Cc] - ref).abs()
            if float(err.max()) > 1e-4:                 # far outside fp32 rounding of a 4-tap blend of O(1) values
                wrong += 1
                idx = (err > 1e-4).nonzero()
                patterns.add(tuple(idx[0].tolist()) + (idx.shape[0],))
                if wrong <= 3:
                    print("   [victim %s] launch %d WRONG vs the fp64 reference: %d elements, channels %s, first at %s, max|err| %.3e" % (
                        variant, iters, idx.shape[0], sorted(set(idx[:, 3].tolist())), idx[0].tolist(), float(err.max())), flush=True)
            if first is None:
                first = y.clone()
            elif not torch.equal(y, first):
                bad += 1
            iters += 1
    print("victim %-5s: %d launches, %d wrong against the fp64 reference (%d distinct (location, size) patterns), %d differed from the first launch"
          % (variant, iters, wrong, len(patterns), bad), flush=True)


def aggressor(math, seconds):
    sys.path.insert(0, PKG)
    import torch
    from segmi import ops
    ops.set_conv_math(math)
    ops.set_conv_winograd(False)
    dev = torch.device("cuda:0")
    x = torch.randn(2, 512, 32, 32, device=dev)
    w = torch.randn(512, 512, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    t0, n = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < seconds:
            for _ in range(20):
                ops.conv2d(x, w, None, 1, 1, 1)
                n += 1
            torch.cuda.synchronize()
    print("aggressor %s: %d convolutions" % (math, n), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--only", default=None, help="run one configuration, e.g. slp:bf16x3")
    ap.add_argument("--cross", action="store_true", help="synthetic victims x real aggressor, real victim x synthetic aggressor")
    ap.add_argument("--role", default=None)
    ap.add_argument("--variant", default=None)
    args = ap.parse_args()
    if args.role == "victim":
        return victim(args.variant, args.seconds)
    if args.role == "aggressor":
        return aggressor(args.variant, args.seconds)
    build_slp_copy()
    me = [sys.executable, os.path.abspath(__file__)]
    if args.cross:
        # which side carries the ingredient?  synthetic victims (tools/probes/pk_mfma_repro.hip: compiler-packed, two hand-written
        # packed forms, scalar) next to the REAL bf16x3 convolution; the REAL SLP-built bilinear kernel next to the synthetic
        # register-only bf16 MFMA loop
        synth = "/tmp/pk_mfma_repro"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", synth, os.path.join(ROOT, "tools", "probes", "pk_mfma_repro.hip")],
                       check=True, stderr=subprocess.DEVNULL)
        for v in ("pk", "asm", "asm2", "scalar"):
            print("== synthetic victim %s, REAL aggressor bf16x3" % v, flush=True)
            a = subprocess.Popen(me + ["--role", "aggressor", "--variant", "bf16x3", "--seconds", str(args.seconds + 8)])
            time.sleep(6)
            subprocess.run([synth, "victim", str(args.seconds), v])
            a.wait()
        print("== REAL victim slp, synthetic aggressor bf16 (register-only MFMA loop)", flush=True)
        a = subprocess.Popen([synth, "aggressor", str(args.seconds + 14), "bf16"])
        subprocess.run(me + ["--role", "victim", "--variant", "slp", "--seconds", str(args.seconds)])
        a.wait()
        return
    for v in ("slp", "noslp"):
        for agg in (None, "f32", "bf16x3"):
            if args.only and args.only != "%s:%s" % (v, agg or "none"):
                continue
            print("== victim %s, aggressor %s" % (v, agg or "none"), flush=True)
            procs = []
            if agg:
                procs.append(subprocess.Popen(me + ["--role", "aggressor", "--variant", agg, "--seconds", str(args.seconds + 8)]))
                time.sleep(6)          # let the aggressor finish importing torch and start launching
            procs.append(subprocess.Popen(me + ["--role", "victim", "--variant", v, "--seconds", str(args.seconds)]))
            for p in procs:
                p.wait()


if __name__ == "__main__":
    main()
