"""Build libsegmi.so (hand-written HIP kernels for gfx950) in-tree.

    python pytorch-segmentation_amd/build.py [--force]

Every csrc/*.hip is compiled with hipcc for gfx950 only and linked with g++ (not `hipcc --hip-link`,
which would bake an /opt/rocm RUNPATH into the library): libsegmi.so must bind to the HIP runtime
that torch has already loaded into the process (same SONAME libamdhip64.so.7), so that torch's
streams and device pointers are meaningful to our kernels.  No GPU is needed to build.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "segmi")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libsegmi.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]
# per-file additions: no compiler-formed packed-fp32 arithmetic (v_pk_mul/fma/add_f32) in the HBM-bound kernels.  The SLP-packed
# form of bilinear_fwd_kernel returns wrong high halves while a bf16x3 convolution kernel of ANOTHER process shares the GPU
# (tools/probes/pk_two_process.py reproduces it with exactly these two kernels: 32 % of the launches wrong against an fp64 reference,
# profiles/r03_pk_two_process_ref.txt);
# the scalar form never does.  These kernels are HBM-bound: no cost.
EXTRA_FLAGS = {f: ["-fno-slp-vectorize", "-fno-vectorize"] for f in ("pool_resize.hip", "conv_winograd.hip", "bn.hip", "loss.hip", "misc.hip", "optim.hip",
                                                    "lovasz.hip", "dwconv_shuffle.hip", "pyramid_bottleneck.hip", "augment.hip")}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "segmi.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs, hdrs = _sources(), _headers()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        return o

    if jobs:
        if verbose:
            print("[segmi.build] hipcc %s: %s" % (ARCH, ", ".join(os.path.basename(j[0]) for j in jobs)), flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = ["g++", "-shared", "-o", LIB] + objs + ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("[segmi.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
