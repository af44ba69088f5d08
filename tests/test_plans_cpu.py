"""CPU-only: the plan table of libzkstark_hip.so (csrc/ntt_host.inc "the plan table") is DATA -- a string parsed per ctx -- and the
library neither starts a process, nor runs a timing trial, nor writes the environment on any path (no global
state except the zk_ctx, no hidden threads).  The parsing is exercised through device-free internal exports; the "nothing is spawned"
part by looking at what the shared object imports and at the sources."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zk_evm_amd", "csrc")


def _lib():
    from zk_evm_amd import build
    lib = C.CDLL(build.build())
    lib.zki_plans_ntt.restype = C.c_int
    lib.zki_plans_ntt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.zki_plans_batch.restype = C.c_int
    lib.zki_plans_batch.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.zki_plans_tree_tops.restype = C.c_int
    lib.zki_plans_tree_tops.argtypes = [C.c_char_p]
    lib.zki_builtin_plans.restype = C.c_char_p
    return lib


def test_plan_string_items():
    lib = _lib()
    s = b"v20f0=2;d21f1=2;d20f0=1;b20r1=96x2;b19r1=0x1;T=1;"
    assert lib.zki_plans_ntt(s, 0, 20, 0) == 2 and lib.zki_plans_ntt(s, 1, 21, 1) == 2 and lib.zki_plans_ntt(s, 1, 20, 0) == 1
    assert lib.zki_plans_ntt(s, 0, 19, 0) == 0                      # no item: the tile kernels
    assert lib.zki_plans_ntt(s, 0, 9, 0) == -1 and lib.zki_plans_ntt(s, 1, 30, 1) == -1 and lib.zki_plans_ntt(s, 1, 20, 3) == -1   # no second plan at all
    assert lib.zki_plans_batch(s, 20, 1) == 96 * 4 + 2 and lib.zki_plans_batch(s, 19, 1) == 1 and lib.zki_plans_batch(s, 18, 1) == -1
    assert lib.zki_plans_tree_tops(s) == 1 and lib.zki_plans_tree_tops(b"v20f0=2;") == -1 and lib.zki_plans_tree_tops(b"T=0;") == 0
    # an item is matched at an item boundary only, and junk values are no verdict
    assert lib.zki_plans_ntt(b"xv20f0=2;", 0, 20, 0) == 0 and lib.zki_plans_ntt(b"v20f0=7;", 0, 20, 0) == 0
    assert lib.zki_plans_batch(b"b20r1=96x3;", 20, 1) == -1 and lib.zki_plans_batch(b"b20r1=99999x1;", 20, 1) == -1
    assert lib.zki_plans_tree_tops(b"XT=1;") == -1
    # a later duplicate does not matter: the first well-placed item wins (the tuner replaces, never appends twice)
    assert lib.zki_plans_ntt(b"v20f0=1;v20f0=2;", 0, 20, 0) == 1


def test_the_compiled_in_table_is_what_the_profiles_justify():
    """kBuiltinPlans may name a second form only with hardware evidence committed under profiles/ (a `*_plan_table_*` file holding
    the tuner's report with that exact string as its first line).  Empty is always fine: the r01-r04 kernels everywhere."""
    builtin = _lib().zki_builtin_plans().decode("ascii")
    assert re.fullmatch(r"([vd][0-9]+f[01]=[12];|b[0-9]+r[0-9]=[0-9]+x[12];|T=[01];)*", builtin), builtin
    if builtin:
        prof = os.path.join(ROOT, "profiles")
        evidence = [f for f in os.listdir(prof) if "plan_table" in f]
        assert any(open(os.path.join(prof, f)).readline().strip() == builtin for f in evidence), "no profiles/*plan_table* file starts with the compiled-in table"


def test_the_library_spawns_nothing_and_leaves_the_environment_alone():
    from zk_evm_amd import build
    so = build.build()
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    for banned in ("posix_spawn", "setenv", "putenv", "fork", "vfork", "execv", "system", "popen", "waitpid", "dladdr"):
        assert not re.search(r"\b%s\w*\b" % banned, syms), banned
    text = ""
    for f in os.listdir(CSRC):
        if f.endswith((".inc", ".hip", ".cuh", ".hpp")):
            text += open(os.path.join(CSRC, f)).read()
    for banned in ("posix_spawn", "setenv", "putenv", "fork", "system", "popen"):
        assert not re.search(r"(?<![A-Za-z0-9_])%s\(" % banned, text), banned
    n_env = len(re.findall(r"\benv_int\(", open(os.path.join(CSRC, "ntt_host.inc")).read()))
    assert n_env <= 8, n_env                                          # (one of them is the definition)
