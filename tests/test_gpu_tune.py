"""GPU: the library's plan trials (csrc/ntt_host.inc, csrc/tune_host.inc) through the real helper process.

By default only the contract is asserted -- whatever the helper did (verdicts, a crash), this process has a verdict for every
transform shape and for the tree tops, and a default commitment still equals the oracle's.  What the helper FOUND on this chip
(identical outputs or not, which plan is faster) is printed; asserting on it is opt-in (ZK_TEST_UNVALIDATED_PLANS=1), like
the other tests of the kernels that no hardware run had pinned when they were written."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_shape_has_a_verdict_after_settling():
    from zk_evm_amd._lib import settle_ntt_plans
    plans = settle_ntt_plans(0)
    if os.environ.get("ZK_NTT_SWAP", "2") != "2":
        pytest.skip("plans forced by ZK_NTT_SWAP")
    items = dict(it.split("=") for it in plans.split(";") if it)
    for key in ["v%df0" % l for l in range(10, 23)] + ["d%df1" % l for l in range(11, 23)]:
        assert items.get(key, "1") in ("1", "2"), (key, plans)
    assert items.get("T") in ("0", "1"), plans
    for l in range(17, 22):                          # column batches of the shapes that have a trial: "<MiB>x<streams>"
        assert re.fullmatch(r"\d+x[12]", items.get("b%dr1" % l, "")), (l, plans)
    assert os.environ["ZK_NTT_SWAP_PLANS"] == plans


def test_commitment_with_settled_plans_equals_oracle(oracle):
    import torch
    from tests.oracle_lib import splitmix64
    from zk_evm_amd import PolynomialBatch
    for n_cols, log_n in ((40, 12), (33, 17)):
        vals = np.stack([splitmix64(99 + c, 1 << log_n) for c in range(n_cols)])
        ref = oracle.commit_values(vals, rate_bits=1, cap_height=4, hasher=0)
        batch = PolynomialBatch.from_values(torch.from_numpy(vals.view(np.int64)).to("cuda:0"), 1, False, 4, hasher=0)
        assert np.array_equal(batch.merkle_tree.cap.elements, ref["cap"])
        assert np.array_equal(batch.polynomial_coeffs(n_cols - 1), ref["coeffs"][n_cols - 1])
        batch.free()


def _helper(extra_env, timeout=600):
    from zk_evm_amd import build
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZK_NTT_")}
    env.update({"ZK_NTT_TUNE_INPROC": "1", "ZK_NTT_SWAP": "2"})
    env.update(extra_env)
    return subprocess.run([build.TUNE, "0"], env=env, capture_output=True, text=True, timeout=timeout)


def test_helper_process_report():
    """Runs the helper exactly as the library does and prints what it found (always passes unless the helper cannot even be
    started: the library's reaction to a dead helper is test_ntt_tune_isolation.py's subject)."""
    r = _helper({})
    sys.stderr.write("zk_ntt_tune rc=%d\n%s\n%s\n" % (r.returncode, r.stdout[-6000:], r.stderr[-2000:]))
    assert r.returncode is not None


@pytest.mark.skipif(os.environ.get("ZK_TEST_UNVALIDATED_PLANS") != "1",
                    reason="asserts on what kernels written without GPU access do on the chip: opt in with ZK_TEST_UNVALIDATED_PLANS=1")
def test_helper_found_identical_outputs_everywhere():
    r = _helper({})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert re.fullmatch(r"([vdb][0-9a-z]+=[0-9x]+;|T=[01];)+", lines[0]), lines[0]
    assert not any("DIFFER" in ln or "failed" in ln for ln in lines[1:]), "\n".join(lines[1:])
    assert sum(ln.startswith("ntt plan") for ln in lines) >= 30 and any(ln.startswith("tree tops") for ln in lines)
