"""TEST INFRASTRUCTURE.  Builds tests/emu/build/libzkstark_emu[_<san>].so: the WHOLE library (zk_evm_amd/csrc/*.hip with every
kernel and every host-side include, from the sources where they lie) compiled as host C++ against the CPU stand-in for the HIP
runtime (tests/emu/hipemu/), after tests/emu/translate.py has rewritten the `<<<...>>>` launches.  Same C ABI, same symbols; "device"
pointers are host pointers.  Used by tests/test_emu_*.py (pytest -m "not gpu") and tools/emu_sanitizers.sh.

    python tests/emu/build_emu.py [--san asan|ubsan|tsan] [--force]
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from translate import translate  # noqa: E402

CSRC = os.path.join(ROOT, "zk_evm_amd", "csrc")
UNITS = ["zkstark", "zk_airs_a", "zk_airs_b", "zk_airs_c", "zk_airs_d", "zk_plonk", "zk_tracegen"]
CXX = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else (shutil.which("clang++") or "g++")
SAN_FLAGS = {
    "": [],
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],      # ASan + UBSan
    "ubsan": ["-fsanitize=undefined", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
    "tsan": ["-fsanitize=thread", "-fno-omit-frame-pointer"],
}


def paths(san=""):
    build = os.path.join(HERE, "build" + ("_" + san if san else ""))
    return build, os.path.join(build, "libzkstark_emu%s.so" % ("_" + san if san else ""))


def _sources():
    for d in (CSRC, os.path.join(ROOT, "include"), os.path.join(HERE, "hipemu")):
        for dp, _, fs in os.walk(d):
            if os.sep + "build" in dp:
                continue
            for f in fs:
                if f.endswith((".hip", ".cuh", ".inc", ".hpp", ".h", ".cpp", ".c")):
                    yield os.path.join(dp, f)
    yield os.path.join(HERE, "translate.py")
    yield os.path.abspath(__file__)


def needs_build(san=""):
    _, out = paths(san)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in _sources())


def build(san="", force=False, verbose=False):
    bdir, out = paths(san)
    if not force and not needs_build(san):
        return out
    src = os.path.join(bdir, "src")
    shutil.rmtree(src, ignore_errors=True)
    os.makedirs(os.path.join(src, "zk_evm_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(src, "include"))
    for f in os.listdir(CSRC):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and f.endswith((".hip", ".cuh", ".inc", ".hpp", ".h")):
            with open(p) as fh:
                text = translate(fh.read())
            with open(os.path.join(src, "zk_evm_amd", "csrc", f), "w") as fh:
                fh.write('#line 1 "%s"\n' % p + text)
    flags = ["-std=c++17", "-O2" if not san else "-O1", "-g", "-fPIC", "-pthread", "-DZK_NTT_EMULATE", "-DZK_HIPEMU", "-I", os.path.join(HERE, "hipemu"),
             "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-pass-failed", "-Wno-unknown-attributes",
             "-Wno-ignored-attributes"] + SAN_FLAGS[san]

    def cc(job):
        name, path, extra = job
        obj = os.path.join(bdir, name + ".o")
        r = subprocess.run([CXX, *flags, *extra, "-c", path, "-o", obj], capture_output=True, text=True)
        return name, obj, r
    jobs = [(u, os.path.join(src, "zk_evm_amd", "csrc", u + ".hip"), ["-x", "c++"]) for u in UNITS]
    # the emulator itself is NOT instrumented for ThreadSanitizer (its scheduler's bookkeeping is shared by the fibers by design); it
    # still tells TSan about fibers and barriers through the annotation calls (HIPEMU_TSAN)
    jobs.append(("hipemu", os.path.join(HERE, "hipemu", "hipemu.cpp"), ["-fno-sanitize=thread", "-DHIPEMU_TSAN=1"] if san == "tsan" else []))
    objs = []
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
        for name, obj, r in ex.map(cc, jobs):
            if verbose or r.returncode != 0:
                sys.stderr.write(r.stderr[-20000:])
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, "%s -c %s" % (CXX, name))
            objs.append(obj)
    link = [CXX, "-shared", "-fPIC", "-pthread", "-o", out, *objs, *SAN_FLAGS[san], "-ldl"]
    if san:
        link.append("-shared-libsan")
    subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    san = sys.argv[sys.argv.index("--san") + 1] if "--san" in sys.argv else ""
    print(build(san, force="--force" in sys.argv, verbose="--verbose" in sys.argv))
