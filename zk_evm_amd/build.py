"""Build libzkstark_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is seven translation units (csrc/*.hip: core, four table-AIR groups, the PLONK prover, the witness-table
generators) compiled in parallel and linked once; a TU is rebuilt when one of the files it includes (transitively)
is newer than its object."""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(_HERE, "csrc", "build")
OUT = os.path.join(_HERE, "libzkstark_hip.so")
TUNE = os.path.join(_HERE, "zk_ntt_tune")          # the offline tuner of the plan table (csrc/ntt_tune_main.c); the library never runs it
UNITS = ["zkstark", "zk_airs_a", "zk_airs_b", "zk_airs_c", "zk_airs_d", "zk_plonk", "zk_tracegen"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path, seen=None):
    """`path` and every file it #includes with quotes, transitively."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in _INC.findall(f.read()):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _stale(unit):
    obj = os.path.join(OBJ, unit + ".o")
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in _deps(os.path.join(CSRC, unit + ".hip")))


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(TUNE) or os.path.getmtime(TUNE) < os.path.getmtime(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_stale(u) or os.path.getmtime(os.path.join(OBJ, u + ".o")) > t for u in UNITS)


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in UNITS if force or _stale(u)]

    def compile_unit(u):
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, u + ".hip"), "-o", os.path.join(OBJ, u + ".o")]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        return u, r
    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 1)) as ex:
        for u, r in ex.map(compile_unit, todo):
            if verbose or r.returncode != 0:
                sys.stderr.write(r.stderr)
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, "hipcc -c %s.hip" % u)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] +
                   [os.path.join(OBJ, u + ".o") for u in UNITS], check=True)
    subprocess.run([shutil.which("gcc") or "gcc", "-O1", "-o", TUNE, os.path.join(CSRC, "ntt_tune_main.c"),
                    "-L" + _HERE, "-lzkstark_hip", "-Wl,-rpath,$ORIGIN"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
