"""CPU: the kernel that makes the trial segment's synthetic traces provable (zk_evm_amd/csrc/tune_trace.cuh, written without GPU
access) run from its own source under the tests/emu shim: it must touch exactly the columns tools/benchlib.py's
synthetic_segment_traces makes binary, write only 0 / 1 there, keep the one-hot groups one-hot and the KeccakSponge block pattern
monotone -- otherwise the prover rejects a filter and the library's tree-top trial silently answers "per tree"."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu_output(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("emu") / "tune_trace_emu")
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-Wno-attributes", "-DZK_NTT_EMULATE", "-I", os.path.join(ROOT, "tests", "emu"),
                        "-I", os.path.join(ROOT, "zk_evm_amd", "csrc"), os.path.join(ROOT, "tests", "emu", "tune_trace_emu.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _python_generator_binary_columns():
    sys.path.insert(0, ROOT)
    from tools.benchlib import synthetic_segment_traces
    out = []
    for x in synthetic_segment_traces([10] * 9, "cpu", seed=3):
        out.append({c for c in range(x.shape[0]) if bool(((x[c] == 0) | (x[c] == 1)).all())})
    return out


def test_touched_columns_are_the_generators_binary_columns_and_hold_bits(emu_output):
    touched = [set() for _ in range(9)]
    for m in re.finditer(r"table (\d+) col (\d+) touched (\d+) min (\d+) max (\d+)", emu_output):
        t, c, n, mn, mx = map(int, m.groups())
        assert n == 1000 and mn >= 0 and mx <= 1, m.group(0)          # every row written, bits only
        touched[t].add(c)
    assert touched == _python_generator_binary_columns()


def test_row_patterns(emu_output):
    rows = {}
    for ln in emu_output.splitlines():
        m = re.match(r"table (\d+) row (\d+) :(.*)", ln)
        if m:
            rows.setdefault(int(m.group(1)), []).append({int(k): int(v) for k, v in (kv.split("=") for kv in m.group(3).split())})
    assert sorted(rows) == list(range(9)) and all(len(v) == 64 for v in rows.values())
    one_hot = {0: [range(0, 17)], 1: [range(1, 33)], 2: [range(6, 24)], 5: [range(0, 3)], 6: [range(15, 17)]}
    for t, groups in one_hot.items():
        for g in groups:
            sums = [sum(r[c] for c in g) for r in rows[t]]
            assert max(sums) <= 1 and 1 in sums, (t, g)               # at most one flag per row, and not all rows empty
    for r in rows[4]:                                                  # KeccakSponge: full block | final block of length ln | neither
        tail = [r[6 + i] for i in range(136)]
        assert tail == sorted(tail)                                    # 0 .. 0 1 .. 1: bytes from position ln on are flagged
        assert not (r[0] and any(tail))
    for r in rows[6]:
        assert r[1] == r[2]                                            # Memory: timestamp = its inverse, in {0, 1}
    for t in range(9):                                                 # the flags vary over the rows (not a constant column)
        for c in rows[t][0]:
            vals = {r[c] for r in rows[t]}
            if t == 4 and c >= 6:
                continue
            assert vals == {0, 1} or c in range(6, 24) or t in (0, 1), (t, c, vals)
