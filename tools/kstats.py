#!/usr/bin/env python3
"""Per-kernel resource usage of one translation unit of the library, without a GPU: compiles csrc/<unit>.hip with
--save-temps into a scratch directory and prints, for every kernel whose demangled name contains <pattern>, VGPRs, spilled
VGPRs, scratch bytes, SGPRs, LDS and the code size.  Usage: tools/kstats.py <unit> [pattern]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    unit = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    d = tempfile.mkdtemp(prefix="kstats_")
    src = os.path.join(ROOT, "zk_evm_amd", "csrc", unit + ".hip")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src, "-o",
                    os.path.join(d, "x.o"), "--save-temps"], cwd=d, check=True, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    sizes = {}
    obj = os.path.join(d, unit + "-hip-amdgcn-amd-amdhsa-gfx950.o")
    for ln in subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", obj], capture_output=True, text=True).stdout.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            sizes[f[7]] = int(f[2])
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?\n(?:.*\n)*?\s+\.wavefront_size:", asm):
        blk = m.group(0)
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]          # noqa: E731
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat in dem:
            rows.append((dem.split("(")[0][-70:], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                         g("sgpr_count"), g("group_segment_fixed_size"), sizes.get(name, 0)))
    print("%-70s %5s %5s %7s %5s %6s %8s" % ("kernel", "vgpr", "spill", "scratch", "sgpr", "lds", "code B"))
    for r in rows:
        print("%-70s %5s %5s %7s %5s %6s %8d" % r)


if __name__ == "__main__":
    main()
