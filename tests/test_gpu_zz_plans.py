"""GPU: the SECOND implementations behind the plan table (csrc/ntt_host.inc "the plan table") against the ORACLE, unconditionally.

The library ships two equivalent forms of three decisions -- the NTT passes on the LDS tile kernels | the lane-swap kernels
(csrc/ntt_swap.cuh), from_values over all columns at once | in column batches, the trace trees' small levels per tree | batched
(merkle.cuh poseidon_merkle_level_multi_*).  The table compiled into the library is empty until a hardware run confirms a second
form, so the default path never reaches them; these tests FORCE them -- per ctx through zk_ctx_set_plans, per process through
ZK_NTT_SWAP / ZK_TREE_BATCH / ZK_NTT_COL_BATCH_MB -- and compare with the C oracle and with the whole-segment oracle proofs.  A
failure here is a parity failure of shipped code: fix or delete that kernel.

The file sorts after every other GPU test on purpose: `pytest -x` then reports the default path's parity (the rows of SURVEY
section 8) before it reaches kernels that had never executed on hardware when this was written (rounds 5 and 6 had no GPU)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.oracle_lib import splitmix64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every transform shape on the lane-swap kernels, 8 MiB column batches on two streams for every from_values shape, tree tops batched
ALL_SECOND = ("".join("v%df0=2;d%df0=2;d%df1=2;" % (l, l, l) for l in range(10, 23)) +
              "".join("b%dr1=8x2;" % l for l in range(10, 22)) + "T=1;")
SWAP_ONLY = "".join("v%df0=2;d%df0=2;d%df1=2;" % (l, l, l) for l in range(10, 23))


def _pythonpath():
    """the repository first, whatever the parent had after it (tests/emu/site when this suite runs on the emulation build)"""
    return ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")


@pytest.fixture
def default_ctx():
    import zk_evm_amd as zk
    ctx = zk.default_context(0)
    yield ctx
    ctx.set_plans(None)


def test_plan_table_round_trip(default_ctx):
    import zk_evm_amd as zk
    ctx = default_ctx
    initial = ctx.get_plans()
    ctx.set_plans("v20f0=2;T=1;")
    assert ctx.get_plans() == "v20f0=2;T=1;"
    with pytest.raises(zk.ZkStarkError):
        ctx.set_plans("v20f0=2; rm -rf")
    assert ctx.get_plans() == "v20f0=2;T=1;"
    ctx.set_plans(None)
    assert ctx.get_plans() == initial


@pytest.mark.parametrize("plans", [SWAP_ONLY, ALL_SECOND], ids=["lane_swap", "lane_swap+batches"])
@pytest.mark.parametrize("hasher", [0, 1])
def test_config2_full_size_on_the_second_forms_equals_oracle(oracle, default_ctx, plans, hasher):
    """BASELINE.json configs[1] at FULL size (116 x 2^20, rate_bits 1, cap_height 4) with every transform on the lane-swap
    kernels (and, second case, in 8 MiB column batches on two streams) EQUAL to the C oracle's from_values: the cap, every
    coefficient of three columns, 64 sampled leaves with their Merkle paths -- against the oracle, not against the tile kernels."""
    import torch
    from zk_evm_amd import PolynomialBatch
    n_cols, log_n, rate_bits, cap_height = 116, 20, 1, 4
    log_N = log_n + rate_bits
    vals = np.stack([splitmix64(0x6FEB51B7EC230F25 + c, 1 << log_n) for c in range(n_cols)])
    ref = oracle.commit_values(vals, rate_bits=rate_bits, cap_height=cap_height, hasher=hasher)
    default_ctx.set_plans(plans)
    batch = PolynomialBatch.from_values(torch.from_numpy(vals.view(np.int64)).cuda(), rate_bits, False, cap_height, hasher=hasher)
    assert np.array_equal(batch.merkle_tree.cap.elements, ref["cap"])
    for c in (0, 57, 115):
        assert np.array_equal(batch.polynomial_coeffs(c), ref["coeffs"][c])
    rng = np.random.default_rng(177 + hasher)
    for leaf in [0, 1, (1 << log_N) - 1] + [int(x) for x in rng.integers(0, 1 << log_N, size=61)]:
        assert np.array_equal(batch.merkle_tree.get(leaf), ref["leaves"][leaf])
        assert np.array_equal(batch.merkle_tree.prove(leaf).siblings, oracle.merkle_prove(ref["digests"], log_N, cap_height, leaf))
    batch.free()


@pytest.mark.parametrize("n_cols,log_n,rate_bits", [(20, 10, 1), (3, 10, 0), (5, 9, 1), (4, 11, 1), (9, 11, 0), (10, 12, 0), (6, 13, 3),
                                                    (300, 13, 1), (64, 14, 1), (40, 15, 1), (20, 16, 1), (7, 17, 1), (33, 18, 1), (6, 19, 1),
                                                    (2, 19, 0), (130, 19, 1), (1, 20, 0), (2, 21, 1), (1, 22, 1)])
def test_every_lane_swap_plan_equals_oracle(oracle, default_ctx, n_cols, log_n, rate_bits):
    """Every pairing of new / old kernels the planner produces -- single-wave transforms (2^10), one wave per tile in both passes
    (2^11 .. 2^16), the LDS-exchange strided kernel (2^17 .. 2^22), one and many columns, rate_bits 0 / 1 / 3, with and without
    column batches -- against the C oracle: cap, two coefficient columns, four leaves."""
    import torch
    from zk_evm_amd import PolynomialBatch
    vals = np.stack([splitmix64(1000 * log_n + c, 1 << log_n) for c in range(n_cols)])
    ref = oracle.commit_values(vals, rate_bits=rate_bits, cap_height=4, hasher=0)
    dev = torch.from_numpy(vals.view(np.int64)).cuda()
    N = 1 << (log_n + rate_bits)
    for plans in (SWAP_ONLY, ALL_SECOND):
        default_ctx.set_plans(plans)
        b = PolynomialBatch.from_values(dev, rate_bits, False, 4)
        assert np.array_equal(b.merkle_tree.cap.elements, ref["cap"]), plans[:12]
        for c in sorted({0, n_cols - 1}):
            assert np.array_equal(b.polynomial_coeffs(c), ref["coeffs"][c]), (plans[:12], c)
        for i in (0, 1, N - 1, 12345 % N):
            assert np.array_equal(b.merkle_tree.get(i), ref["leaves"][i]), (plans[:12], i)
        b.free()


_COMMIT_CHILD = r"""
import sys, json, hashlib
import numpy as np, torch
from zk_evm_amd import PolynomialBatch
from tests.oracle_lib import splitmix64
out = {}
for n_cols, log_n, rate_bits in json.loads(sys.argv[1]):
    vals = np.stack([splitmix64(1000 * log_n + c, 1 << log_n) for c in range(n_cols)])
    b = PolynomialBatch.from_values(torch.from_numpy(vals.view(np.int64)).cuda(), rate_bits, False, 4)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(b.merkle_tree.cap.elements).tobytes())
    for c in sorted({0, n_cols - 1}):
        h.update(b.polynomial_coeffs(c).tobytes())
    for i in (0, 1, (1 << (log_n + rate_bits)) - 1, 12345 % (1 << (log_n + rate_bits))):
        h.update(b.merkle_tree.get(i).tobytes())
    out["%d,%d,%d" % (n_cols, log_n, rate_bits)] = h.hexdigest()
    b.free()
print("RESULT " + json.dumps(out))
"""


def test_process_wide_switches_give_the_same_commitments():
    """The load-time switches (ZK_NTT_SWAP=0|1, ZK_NTT_SWAP_CONTIG, ZK_NTT_COL_BATCH_MB: read once, so a child process each):
    tile kernels, the strided lane-swap kernels alone, the whole family, the family in column batches -- same caps, coefficients
    and leaves."""
    shapes = [[20, 10, 1], [7, 17, 1], [33, 18, 1], [6, 19, 1], [5, 20, 1], [2, 21, 1], [6, 13, 3], [64, 14, 1], [130, 19, 1]]
    res = {}
    for key, env in (("tile", {"ZK_NTT_SWAP": "0"}), ("strided", {"ZK_NTT_SWAP": "1", "ZK_NTT_SWAP_CONTIG": "0"}),
                     ("all", {"ZK_NTT_SWAP": "1", "ZK_NTT_SWAP_CONTIG": "1"}), ("all+batches", {"ZK_NTT_SWAP": "1", "ZK_NTT_COL_BATCH_MB": "64"})):
        r = subprocess.run([sys.executable, "-c", _COMMIT_CHILD, json.dumps(shapes)], capture_output=True, text=True, cwd=ROOT,
                           env=dict(os.environ, PYTHONPATH=_pythonpath(), **env), timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[key] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert len(res["tile"]) == len(shapes)
    for key in ("strided", "all", "all+batches"):
        assert res[key] == res["tile"], (key, [k for k in res["tile"] if res[key].get(k) != res["tile"][k]])


@pytest.mark.parametrize("plans", ["T=1;", SWAP_ONLY, ALL_SECOND], ids=["tree_tops", "lane_swap", "all"])
def test_second_forms_prove_the_same_segments(oracle, default_ctx, plans):
    """Whole nine-table segment proofs with the second forms switched on, word for word against the oracle's segment prover
    (tests/test_gpu_segment.py's comparisons: 2^4 .. 2^5 rows both hashers, 2^12 .. 2^17 rows, the golden fixtures)."""
    from tests import test_gpu_segment as tgs
    default_ctx.set_plans(plans)
    tgs.test_segment_proof_matches_oracle(oracle, 0, [True] * 9)
    tgs.test_segment_proof_matches_oracle(oracle, 1, [True, False, True, False, False, False, True, True, False])
    if plans == ALL_SECOND:          # (2^12 .. 2^17 rows: the oracle's proof takes a minute; once, with everything switched on)
        tgs.test_segment_proof_matches_oracle_at_scale(oracle, 0, [16, 13, 15, 12, 13, 14, 17, 12, 13])
    for idx in (0, 1):
        tgs.test_segment_matches_golden_fixture(idx)


@pytest.mark.parametrize("switches", [{"ZK_TREE_BATCH": "1", "ZK_LANES": "0"},
                                      {"ZK_NTT_SWAP": "1", "ZK_NTT_COL_BATCH_MB": "8", "ZK_TREE_BATCH": "1", "ZK_TREE_BATCH_TOP_LOG": "12"}],
                         ids=["tree_batch_lanes_off", "all_switches"])
def test_process_wide_switches_prove_the_same_segments(switches):
    """The same with the load-time switches (a child pytest): the batched tree tops with the side lane off, and everything at
    once with the batch boundary moved down to 2^12 nodes."""
    which = "test_segment_proof_matches_oracle or test_segment_matches_golden_fixture"
    if "ZK_NTT_SWAP" not in switches:          # (the 2^12 .. 2^17-row cases once, in the variant that switches everything on)
        which = "(%s) and not at_scale" % which
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_segment.py"), "-m", "gpu", "-x", "-q",
                        "-k", which, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=_pythonpath(), **switches), timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_offline_tuner_finds_identical_outputs_everywhere(oracle, default_ctx):
    """`zk_ntt_tune` (the offline tool that produces a plan table for a device): exit status 0 = no second form produced a
    different word anywhere (7 = a DIFFER, i.e. a parity failure of shipped code); its first line is a well-formed plan string;
    and a commitment made under THAT table equals the oracle's."""
    import torch
    from zk_evm_amd import PolynomialBatch, build
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZK_NTT_") and k != "ZK_TREE_BATCH"}
    r = subprocess.run([build.TUNE, "0"], env=env, capture_output=True, text=True, timeout=900)
    sys.stderr.write("zk_ntt_tune rc=%d\n%s\n%s\n" % (r.returncode, r.stdout[-8000:], r.stderr[-2000:]))
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])
    lines = r.stdout.splitlines()
    assert re.fullmatch(r"([vd][0-9]+f[01]=[12];|b[0-9]+r1=[0-9]+x[12];|T=[01];)+", lines[0]), lines[0]
    assert not any("DIFFER" in ln or "failed" in ln for ln in lines[1:]), "\n".join(lines[1:])
    assert sum(ln.startswith("ntt plan") for ln in lines) >= 30 and any(ln.startswith("tree tops") for ln in lines)
    default_ctx.set_plans(lines[0])
    for n_cols, log_n in ((40, 12), (33, 17), (70, 18)):
        vals = np.stack([splitmix64(99 + c, 1 << log_n) for c in range(n_cols)])
        ref = oracle.commit_values(vals, rate_bits=1, cap_height=4, hasher=0)
        batch = PolynomialBatch.from_values(torch.from_numpy(vals.view(np.int64)).to("cuda:0"), 1, False, 4, hasher=0)
        assert np.array_equal(batch.merkle_tree.cap.elements, ref["cap"])
        assert np.array_equal(batch.polynomial_coeffs(n_cols - 1), ref["coeffs"][n_cols - 1])
        batch.free()
