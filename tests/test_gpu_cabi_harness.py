"""-m gpu: the C ABI driven from plain C, no Python in the process (tests/cabi/harness.c, the role SURVEY 8(b) gives a
"C++ harness" and INTEGRATION.md gives the Rust `-sys` crate): built with gcc against include/zkstark.h, linked to the
in-tree library, its caps compared with tests/golden/commit_caps.json."""
import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "cabi_harness")
    libdir = os.path.join(ROOT, "zk_evm_amd")
    cmd = [shutil.which("gcc") or "gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cabi", "harness.c"),
           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lzkstark_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_c_harness_matches_golden_caps(tmp_path):
    exe = _build(tmp_path)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))["cases"]
    for c in cases:
        if c["log_n"] > 10:
            continue
        out = subprocess.run([exe] + [str(c[k]) for k in ("n_cols", "log_n", "rate_bits", "cap_height", "hasher", "seed")],
                             check=True, capture_output=True, text=True, timeout=300).stdout.splitlines()
        assert out[0].startswith("version zkstark-hip")
        cap = [int(x) for x in out[1].split()[1:]]
        assert cap == [x for row in c["cap"] for x in row], c["name"]
        assert out[2] == "memory log_n 3 unpadded 4"          # three operations + one timestamp-gap dummy, padded to 8 rows
        assert out[3].startswith('error "') and "cfg" in out[3]
