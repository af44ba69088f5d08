"""Build libzkstark_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "zkstark.hip")
OUT = os.path.join(_HERE, "libzkstark_hip.so")


def _deps():
    d = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    inc = os.path.join(os.path.dirname(_HERE), "include")
    # zk_all_stark.h is generated data for callers; the library does not include it
    d += [os.path.join(inc, f) for f in os.listdir(inc) if f != "zk_all_stark.h"]
    return d


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-o", OUT, SRC]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
