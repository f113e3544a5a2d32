#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch footprint of the HIP sources, read from the code-object metadata hipcc emits.

Usage: python tools/kernel_resources.py [file.hip ...]      (default: every .hip under gpax_amd/csrc)

Why it matters on this path (gpax/models/gp.py:160-164 -> the blocked Cholesky): a chain kernel is placed on a CU at
once only if it fits in what two resident trailing-update workgroups leave free there (512 - 2 x 200 = 112 VGPRs per
SIMD lane, 160 - 2 x 64 = 32 KB of LDS); this prints the numbers that decide it.  Runs without a GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True,
                             text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def resources(src, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "--cuda-device-only", "-S", src, "-o", out, *extra]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    md = text[text.index("amdhsa.kernels:"):]
    rows = []
    for blk in re.split(r"\n  - \.agpr_count:", md)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        row = {k: (get(k).group(1) if get(k) else "?") for k in
               ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")}
        rows.append(row)
    return rows


def kernel_asm(src, name_part, extra=()):
    """The gfx950 assembly of the ONE kernel of `src` whose mangled name contains `name_part` (label to end of function)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "--cuda-device-only", "-S", src, "-o", out, *extra]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    labels = [m.group(1) for m in re.finditer(r"^(\S*%s\S*):\s" % re.escape(name_part), text, re.M)]
    assert len(labels) == 1, labels
    start = text.index("\n" + labels[0] + ":")
    end = text.index(".Lfunc_end", start)
    return text[start:end]


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(ROOT, "gpax_amd", "csrc", f)
                                  for f in os.listdir(os.path.join(ROOT, "gpax_amd", "csrc")) if f.endswith(".hip"))
    print("%-78s %5s %5s %5s %7s %7s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds(st)"))
    for s in srcs:
        rows = resources(s)
        dm = demangle([r["name"] for r in rows])
        for r in rows:
            n = dm[r["name"]]
            n = re.sub(r"\(.*", "", n).replace("gpx::", "")
            print("%-78s %5s %5s %5s %7s %7s %8s %8s" % (n[:78], r["vgpr_count"], r["agpr_count"], r["sgpr_count"],
                                                       r["vgpr_spill_count"], r["sgpr_spill_count"],
                                                       r["private_segment_fixed_size"], r["group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
