"""The oracle's fast evaluation of Poseidon (oracle/poseidon_fast.c: blocked partial rounds, lazy arithmetic, AVX2 matrix
products -- what bench.py's cpu_baseline times) against the plain 4 + 22 + 4 round definition (oracle/poseidon.c) that the
reference-tree KATs pin: the same permutation on edge values and random states, and the same hashes / caps through both."""
import numpy as np

from tests.oracle_lib import load_oracle, splitmix64

P = 0xFFFFFFFF00000001


def _both(o, st):
    a, b = np.array(st, dtype=np.uint64), np.array(st, dtype=np.uint64)
    o.lib.orc_poseidon_permute(a)
    o.lib.orc_poseidon_permute_fast(b)
    return a, b


def test_fast_permutation_equals_plain_definition():
    o = load_oracle()
    edge = [0, 1, P - 1, P, P + 1, (1 << 64) - 1, 1 << 32, (1 << 32) - 1, 0xFFFFFFFF00000000, 7, P - 2, 1 << 63]
    cases = [[0] * 12, list(range(12)), edge, edge[::-1], [P - 1] * 12, [(1 << 64) - 1] * 12]
    rnd = splitmix64(0xC0DE, 12 * 2000).reshape(-1, 12)
    cases += [list(map(int, r)) for r in rnd]
    for st in cases:
        a, b = _both(o, st)
        assert np.array_equal(a, b), st
        assert all(int(x) < P for x in b)
    # upstream vectors quoted in SURVEY 8(c): perm(0..0)[0], perm(0,1,..,11)[0]
    assert int(_both(o, [0] * 12)[1][0]) == 0x3c18a9786cb0b359
    assert int(_both(o, list(range(12)))[1][0]) == 0xd64e1e3efc5b8e9e


def test_hashes_and_commitment_agree_through_both_evaluations():
    o = load_oracle()
    vals = np.stack([splitmix64(99 + c, 1 << 7) for c in range(21)])
    try:
        o.lib.orc_poseidon_use_fast(1)
        fast = o.commit_values(vals, rate_bits=1, cap_height=4, hasher=0)
        o.lib.orc_poseidon_use_fast(0)
        plain = o.commit_values(vals, rate_bits=1, cap_height=4, hasher=0)
    finally:
        o.lib.orc_poseidon_use_fast(1)
    for k in ("cap", "digests", "leaves", "coeffs"):
        assert np.array_equal(fast[k], plain[k]), k
