"""CPU: the oracle's restatement of the per-segment driver (oracle/segment.py): it reproduces the committed
self-golden segment digests, and its cross-table-lookup verification accepts a hand-built balanced witness
(MemBefore rows looked up by Memory and looking into Memory) and rejects a perturbed one."""
import json
import os

import numpy as np
import pytest

import tests.oracle_lib as ol
from oracle import all_stark as oas
from oracle import segment as oseg
from oracle import stark as orc

P = 0xFFFFFFFF00000001
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "segment_proof.json")


@pytest.mark.parametrize("idx", [0, 1])
def test_oracle_reproduces_golden_segment(oracle, idx):
    from tests.golden.gen_segment_proof import oracle_case
    ol.setup_fri_api(oracle)
    case = json.load(open(GOLDEN))["cases"][idx]
    got = oracle_case(oracle, case["hasher"], case["in_use"], case["seed"])
    assert got == case["proof"]


def _balanced_traces(rng, k=5):
    tr = [np.zeros((c, 16), dtype=np.uint64) for c in oas.TABLE_COLUMNS]
    mb = tr[oas.MEM_BEFORE]
    mb[0, :k] = 1
    mb[1:4, :k] = rng.integers(0, 50, size=(3, k))
    mb[4:12, :k] = rng.integers(0, 1 << 32, size=(8, k))
    m = tr[oas.MEMORY]
    m[1, :] = 1          # padding rows: timestamp = timestamp_inv = 1 -> filter_mem_before = 0
    m[2, :] = 1
    m[0, :k] = 1         # the k initialisation rows: filter on, is_read 0, timestamp 0, same address / value
    m[1, :k] = 0
    m[2, :k] = 0
    m[4:7, :k] = mb[1:4, :k]
    m[7:15, :k] = mb[4:12, :k]
    return tr


def _ctl_zs_first(traces, ctls, challenges):
    per_table = oseg.cross_table_lookup_data(traces, ctls, challenges, 3)
    zs = []
    for t, zds in enumerate(per_table):
        cols = [[int(x) for x in c] for c in traces[t]]
        zs.append([orc.partial_sums(cols, zd.columns_filters, zd.challenge, 3)[-1][0] for zd in zds])
    return zs


def test_cross_table_lookups_balance_and_detect_tampering():
    rng = np.random.default_rng(7)
    ctls = oas.build_ctls()
    challenges = [orc.GrandProductChallenge(int(rng.integers(1, P, dtype=np.uint64)), int(rng.integers(1, P, dtype=np.uint64)))
                  for _ in range(2)]
    traces = _balanced_traces(rng)
    zs = _ctl_zs_first(traces, ctls, challenges)
    assert [len(z) for z in zs] == [2, 4, 12, 4, 10, 2, 8, 4, 2]      # Z columns per table (two challenges)
    ok, why = oseg.verify_cross_table_lookups(ctls, zs, None, 2)
    assert ok, why
    assert any(z != 0 for z in zs[oas.MEM_BEFORE]) and any(z != 0 for z in zs[oas.MEMORY])
    traces[oas.MEMORY][7, 2] ^= 1                                      # one value limb of one looked-up row
    ok, why = oseg.verify_cross_table_lookups(ctls, _ctl_zs_first(traces, ctls, challenges), None, 2)
    assert not ok and why.startswith("CTL")


def test_mem_cap_from_merkle_cap_keccak_matches_to_vec(oracle):
    """`MemCap::from_merkle_cap` (proof.rs:606-621) takes `h.to_vec()`: for KeccakHash<25> that is four elements from the
    7,7,7,4-byte little-endian chunks of the 25-byte digest, not the four 8-byte words of the padded 32-byte slot."""
    from tests.oracle_lib import splitmix64
    from zk_evm_amd.segment import MemCap
    vals = np.stack([splitmix64(900 + k, 64) for k in range(12)])
    for hasher in (0, 1):
        cap = oracle.commit_values(vals, rate_bits=1, cap_height=4, hasher=hasher, want_leaves=False)["cap"]
        exp = oseg.mem_cap_from_merkle_cap(cap, hasher)
        got = MemCap.from_merkle_cap(cap, hasher).mem_cap
        assert got == [[int(x) for x in h] for h in exp]
        assert MemCap.from_elements(exp).mem_cap == got
    digest = bytes(range(1, 26)) + bytes(7)                      # a hand-checkable 25-byte digest in its 32-byte slot
    slot = np.frombuffer(digest, dtype=np.uint64).reshape(1, 4)
    got = MemCap.from_merkle_cap(slot, 1).mem_cap[0]
    assert got == [int.from_bytes(bytes(range(1, 8)), "little"), int.from_bytes(bytes(range(8, 15)), "little"),
                   int.from_bytes(bytes(range(15, 22)), "little"), int.from_bytes(bytes(range(22, 26)), "little")]
    assert got != [int(x) for x in slot[0]]


@pytest.mark.parametrize("erigon", [False, True])
def test_product_extra_looking_values_match_oracle_restatement(erigon):
    """zk_evm_amd.segment.get_memory_extra_looking_values (the rows check_ctls adds to the Memory CTL) == the oracle's
    restatement of verifier.rs `get_memory_extra_looking_sum`'s write list, as multisets, for both feature sets."""
    import zk_evm_amd.segment as sg
    from tests.test_gpu_segment import make_pv, to_public_values
    rng = np.random.default_rng(5 + erigon)
    pvd = make_pv(rng)
    pvd["registers_before"] = dict(program_counter=77, is_kernel=1, stack_len=3, stack_top=1 << 200, context=2, gas_used=9)
    pvd["registers_after"] = dict(program_counter=1234, is_kernel=0, stack_len=1, stack_top=5, context=0, gas_used=1 << 40)
    if erigon:
        pvd["burn_addr"] = 0xDEADBEEF << 64
    got = sg.get_memory_extra_looking_values(to_public_values(pvd), 0xABCDEF << 100, 54321)
    exp = [[0, 0, seg, idx] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] + [2]
           for seg, idx, val in oseg.public_memory_writes(pvd, 0xABCDEF << 100, 54321)]
    assert sorted(got) == sorted(exp) and len(got) == len(exp) > 290
