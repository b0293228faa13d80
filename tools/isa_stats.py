#!/usr/bin/env python
"""Static look at the compiled conv kernels (no GPU): registers / scratch per instantiation and, for the K loops, the
instruction mix and how VALU work is interleaved with the matrix instructions.

    python tools/isa_stats.py ["kernel name prefix" ...]     # default: the 128x128 fprop / dgrad / wgrad kernels

Compiles csrc/conv_igemm.hip to gfx950 assembly (hipcc -S --cuda-device-only) under /tmp and parses it.  Used to check the
K loops: every `M[v8 ...]` group = one MFMA with 8 VALU in its shadow.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pytorch-segmentation_amd", "csrc", "conv_igemm.hip")
ASM = "/tmp/segmi_conv_igemm.s"
DEFAULT = ["conv_dma_kernel<128, 128, 2, 2, 0, true, true>", "conv_dma_kernel<128, 128, 2, 2, 1, true, true>", "conv_wgrad_dma_kernel<128, 128, true, true>"]


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")


def main():
    want = sys.argv[1:] or DEFAULT
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value",
                    "-S", "--cuda-device-only", "-o", ASM, SRC], check=True, stderr=subprocess.DEVNULL)
    s = open(ASM).read()
    meta = {}
    for b in re.split(r"\n\s*\.amdhsa_kernel ", s)[1:]:
        name = demangle(b.split("\n")[0]).split("(")[0]
        meta[name] = (int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", b).group(1)), int(re.search(r"\.amdhsa_accum_offset (\d+)", b).group(1)),
                      int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", b).group(1)))
    for f in re.split(r"\n(?=_Z\w+:)", s):
        name = demangle(f.split(":", 1)[0]).split("(")[0]
        if not any(name.startswith("void " + w) or name.startswith(w) for w in want):
            continue
        key = name[5:] if name.startswith("void ") else name
        v = meta.get(name) or meta.get(key)
        print("== %s\n   registers %s (arch VGPRs up to %s, rest AGPRs), scratch %s B" % ((key,) + tuple(v or ("?", "?", "?"))))
        blocks, cur, lab = [], [], "entry"
        for l in f.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append((lab, cur)); cur, lab = [], l.split(":")[0]
            else:
                t = l.strip()
                if t and not t.startswith((";", ".")):
                    cur.append(t.split()[0])
        blocks.append((lab, cur))
        for lab, ops in blocks:
            n = sum(1 for o in ops if o.startswith("v_mfma"))
            if n < 8:
                continue
            c = collections.Counter()
            gaps, g = [], collections.Counter()
            for o in ops:
                if o.startswith("v_mfma"):
                    c["mfma"] += 1
                    gaps.append("M[v%d d%d n%d w%d]" % (g["v"], g["d"], g["n"], g["w"])); g = collections.Counter()
                elif o.startswith("v_"):
                    c["valu"] += 1; c["  " + o] += 1; g["v"] += 1
                elif o.startswith("ds_"):
                    c[o] += 1; g["d"] += 1
                elif o.startswith("buffer_"):
                    c["buffer"] += 1
                elif o.startswith("s_nop"):
                    c["s_nop"] += 1; g["n"] += 1
                elif o.startswith(("s_waitcnt", "s_barrier")):
                    c[o.split("_")[1] if False else o] += 1; g["w"] += 1
                elif o.startswith("s_"):
                    c["salu"] += 1
            print("   block %s: %s" % (lab, dict(sorted(c.items()))))
            print("   VALU / LDS / nop / wait counts before each MFMA:\n     " + " ".join(gaps))


if __name__ == "__main__":
    main()
