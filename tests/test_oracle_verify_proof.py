"""CPU: the consistent nine-table segment (tests/consistent_segment.py) and the oracle's restatement of the
reference's top-level verifier pieces -- `get_memory_extra_looking_sum` (verifier.rs:319-512) and
`verify_cross_table_lookups` -- without proving anything: the CTL sums are taken straight from the rows."""
import numpy as np

from oracle import airs as oairs
from oracle import all_stark as A
from oracle import mem_trace as mt
from oracle import segment as oseg
from oracle import stark as S
from tests import consistent_segment as cs
from tests.test_oracle_tracegen import _check_air

KH = 0x1F2E3D4C5B6A79880102030405060708090A0B0C0D0E0F101112131415161718


def test_public_memory_writes_shape():
    pv = cs.make_public_values(np.random.default_rng(0))
    w = oseg.public_memory_writes(pv, KH, 1234)
    assert len(w) == 25 + 8 + 256 + 12                      # verifier.rs:330-494 (eth_mainnet)
    assert len({(s, i) for s, i, _ in w}) == len(w)         # distinct addresses
    assert (5, 45, KH) in w and (5, 46, 1234) in w and (33, 6, 31337) in w
    assert all(0 <= v < 1 << 256 for _, _, v in w)


def test_consistent_segment_rows_and_ctls_balance():
    rng = np.random.default_rng(1)
    code = rng.bytes(300)
    traces, pv, before = cs.build(rng, oairs.CPU_TEST_CONSTS[0], code, KH)
    assert [t.shape[0] for t in traces] == list(A.TABLE_COLUMNS)
    _check_air(oairs.make_eval_cpu(*oairs.CPU_TEST_CONSTS), traces[A.CPU])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    _check_air(oairs.AIRS[1][0], traces[A.MEM_BEFORE])
    _check_air(oairs.AIRS[1][0], traces[A.MEM_AFTER])
    m = traces[A.MEMORY]
    assert int(m[mt.FILTER].sum()) == len(before) + 301    # initialisation + public-value writes; dummies unfiltered
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(0x1234567890ABCDEF % S.P, 987654321987), S.GrandProductChallenge(31, 0xFFFF0000FFFF)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    ok, why = oseg.verify_cross_table_lookups(ctls, zf, extra, 2)
    assert not ok and why.startswith("CTL 6")               # without the public-value writes Memory does not balance
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    # a different public value (one block hash) breaks exactly the Memory CTL
    pv2 = dict(pv, prev_hashes=[bytes(32)] + pv["prev_hashes"][1:])
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv2, c, KH, len(code)) for c in ch]
    ok, why = oseg.verify_cross_table_lookups(ctls, zf, extra, 2)
    assert not ok and why.startswith("CTL 6")
