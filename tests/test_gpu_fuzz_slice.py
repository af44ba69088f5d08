"""-m gpu: a bounded, fixed-seed slice of tests/fuzz_parity.py (randomised differential parity: device prover vs oracle
prover word for word at random heights / hashers / FRI shapes / live-table sets, and the two witness generators with real
control flow cell for cell) so the driver's GPU run executes it.  ~20 s per mode; the open-ended walk stays a script."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode,seed", [("tables", 31337), ("segment", 31338), ("tracegen", 31339)])
def test_fuzz_slice(mode, seed):
    cmd = [sys.executable, "-m", "tests.fuzz_parity", "18", str(seed)] + ([] if mode == "tables" else [mode])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["mismatches"] == 0 and "FAILED" not in out
    assert (out.get("cases") or out.get("segment_cases") or out.get("memory_logs")) > 0


# Paths the library picks by size or by a load-time switch, forced the other way for a fuzz slice of the per-table provers
# (the device proofs must equal the oracle's word for word on either path): the FRI batch combination on the coefficients
# (default only from ~2^27 opened coefficients, i.e. never at fuzz heights), one lane instead of two.
@pytest.mark.parametrize("env", [{"ZK_FRI_COEFF_COMBINE_MIN_LOG": "0"}, {"ZK_FRI_COEFF_COMBINE": "0", "ZK_LANES": "0"}],
                         ids=["fri_coeff_combine", "value_combine_one_lane"])
def test_fuzz_slice_alternate_paths(env):
    cmd = [sys.executable, "-m", "tests.fuzz_parity", "12", "4242"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["mismatches"] == 0 and "FAILED" not in out and out["cases"] > 0
