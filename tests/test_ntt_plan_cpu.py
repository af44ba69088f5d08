"""CPU: the pass plans of the batched NTT (zk_evm_amd/csrc/ntt_host.inc plan_passes / plan_passes_for) through an internal export of
the library -- no device involved.  Default: the r01-r04 plans.  With ZK_NTT_SWAP=1 (read once at load time: child processes):
transforms of 2^17 .. 2^21 points are planned around the lane-swap kernels' shapes (csrc/ntt_swap.cuh: 10 real stages per wave in the
contiguous pass, 7 .. 10 row bits in the strided one), everything else is untouched.  Always: the passes' stages add up to L, the
distances chain, the last pass is the contiguous one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from zk_evm_amd import build
lib = C.CDLL(build.build())
lib.zki_ntt_plan.restype = C.c_int
lib.zki_ntt_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
out = {}
for L in range(0, 29):
    for free in (0, 1, 3):
        for dit in (0, 1):
            if free and not dit:
                continue                                   # (only an extension has free stages)
            buf = (C.c_int * 16)()
            k = lib.zki_ntt_plan(L, free, dit, buf, 8)
            out["%%d,%%d%%s" %% (L, free, ",dit" if dit and not free else "")] = [[buf[2 * i], buf[2 * i + 1]] for i in range(k)] if k >= 0 else None
print("RESULT " + json.dumps(out))
""" % ROOT


def _plans(env):
    r = subprocess.run([sys.executable, "-c", _CHILD], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **env), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


def _well_formed(plans):
    for key, plan in plans.items():
        L, free = map(int, key.split(",")[:2])
        assert plan, key
        assert sum(r for _, r in plan) == L, (key, plan)
        assert plan[-1][0] == 0, (key, plan)
        d = L
        for log_d, r in plan:                      # largest distance first: each pass covers [log_d, log_d + r)
            assert log_d + r == d, (key, plan)
            d = log_d
        assert plan[-1][1] <= 13


def test_default_plans_are_the_old_ones():
    p = _plans({"ZK_NTT_SWAP": "0"})
    _well_formed(p)
    assert p["20,0"] == [[11, 9], [0, 11]] and p["21,1"] == [[12, 9], [0, 12]] and p["10,0"] == [[0, 10]]


def test_swap_plans_use_the_new_kernels_shapes():
    p = _plans({"ZK_NTT_SWAP": "1"})
    _well_formed(p)
    q = _plans({"ZK_NTT_SWAP": "1", "ZK_NTT_SWAP_CONTIG": "0"})
    _well_formed(q)
    d = _plans({"ZK_NTT_SWAP": "0"})
    for L in range(11, 21):                              # values -> coefficients: the wave kernel's 10 + a strided pass of 1 .. 10 row bits
        assert p["%d,0" % L] == [[10, L - 10], [0, 10]]
    for L in range(15, 22):                              # rate_bits = 1: the contiguous pass takes the free stage as well; the strided
        assert p["%d,1" % L] == [[11, L - 11], [0, 11]]     # pass of coefficients -> values needs >= 4 row bits
    for L in range(14, 21):                              # coefficients -> values without extension
        assert p["%d,0,dit" % L] == [[10, L - 10], [0, 10]]
    for key in ("12,1", "13,1", "14,1", "11,0,dit", "12,0,dit", "13,0,dit"):
        assert p[key] == d[key], key
    assert p["21,0"] == [[11, 10], [0, 11]] and p["22,0"] == [[12, 10], [0, 12]] and p["22,1"] == [[12, 10], [0, 12]]
    for key in d:                                        # everything the new kernels do not cover keeps the old plan
        L, free = map(int, key.split(",")[:2])
        if free > 1 or L < 11 or L > 22:
            assert p[key] == d[key] and q[key] == d[key], key
        if L < 17:
            assert q[key] == d[key], key                 # (strided kernels only: nothing below 2^17)
    assert q["20,0"] == [[11, 9], [0, 11]] and q["21,1"] == [[11, 10], [0, 11]]      # strided kernels only: contiguous 11 | 12 by the tile kernel
