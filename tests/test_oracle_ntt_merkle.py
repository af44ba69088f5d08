"""Oracle self-checks for the parts no reference KAT pins (SURVEY 8(c) 'parity unpinned'):
NTT ordering/coset conventions are checked against direct polynomial evaluation, the Merkle
restatement against its own verifier (verify_merkle_proof_to_cap semantics)."""
import numpy as np
import pytest

from tests.oracle_lib import P, splitmix64

G = 14293326489335486720


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9])
def test_fft_is_evaluation_on_subgroup(oracle, log_n):
    L = oracle.lib
    n = 1 << log_n
    coeffs = splitmix64(7 + log_n, n) % np.uint64(P)
    vals = coeffs.copy()
    L.orc_fft(vals, log_n)
    w = L.orc_gl_root_of_unity(log_n)
    for i in range(n):
        assert int(vals[i]) == L.orc_eval_poly(coeffs, n, L.orc_gl_pow(w, i))
    back = vals.copy()
    L.orc_ifft(back, log_n)
    assert np.array_equal(back, coeffs)


def test_noncanonical_inputs_are_reduced(oracle):
    L = oracle.lib
    a = np.array([P, P + 5, (1 << 64) - 1, 3], dtype=np.uint64)
    b = (a % np.uint64(P)).copy()
    L.orc_ifft(a, 2)
    L.orc_ifft(b, 2)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("log_n,rate_bits", [(3, 1), (6, 1), (4, 3)])
def test_lde_is_evaluation_on_coset(oracle, log_n, rate_bits):
    L = oracle.lib
    n = 1 << log_n
    N = n << rate_bits
    coeffs = splitmix64(99, n) % np.uint64(P)
    out = np.zeros(N, dtype=np.uint64)
    L.orc_lde(coeffs, log_n, rate_bits, out)
    wN = L.orc_gl_root_of_unity(log_n + rate_bits)
    for j in range(N):
        x = L.orc_gl_mul(G, L.orc_gl_pow(wN, j))
        assert int(out[j]) == L.orc_eval_poly(coeffs, n, x)


@pytest.mark.parametrize("hasher", [0, 1])
@pytest.mark.parametrize("n_cols,log_n,cap_height", [(3, 4, 2), (12, 5, 4), (5, 3, 4), (20, 6, 0)])
def test_commit_layout_and_proofs(oracle, hasher, n_cols, log_n, cap_height):
    L = oracle.lib
    n = 1 << log_n
    vals = splitmix64(5, n_cols * n).reshape(n_cols, n)
    r = oracle.commit_values(vals, rate_bits=1, cap_height=cap_height, hasher=hasher)
    log_N = log_n + 1
    N = 1 << log_N
    # coeffs: ifft of each column
    for c in range(n_cols):
        col = vals[c].copy()
        L.orc_ifft(col, log_n)
        assert np.array_equal(col, r["coeffs"][c])
    # leaves: bit-reversed rows of the natural-order LDE
    wN = L.orc_gl_root_of_unity(log_N)
    for j in (0, 1, 2, N // 2 + 1, N - 1):
        x = L.orc_gl_mul(G, L.orc_gl_pow(wN, j))
        row = r["leaves"][bitrev(j, log_N)]
        for c in range(n_cols):
            assert int(row[c]) == L.orc_eval_poly(r["coeffs"][c], n, x)
    # every leaf opens to the cap
    nsib = log_N - cap_height
    for idx in range(N):
        sib = oracle.merkle_prove(r["digests"], log_N, cap_height, idx)
        ok = L.orc_merkle_verify(np.ascontiguousarray(r["leaves"][idx]), n_cols, idx,
                                 sib if sib.size else np.zeros((1, 4), np.uint64), nsib,
                                 r["cap"], hasher)
        assert ok == 1
    # tampering breaks it
    bad = r["leaves"][1].copy()
    bad[0] ^= np.uint64(1)
    sib = oracle.merkle_prove(r["digests"], log_N, cap_height, 1)
    assert L.orc_merkle_verify(bad, n_cols, 1, sib if sib.size else np.zeros((1, 4), np.uint64),
                               nsib, r["cap"], hasher) == 0


def test_poseidon_hash_or_noop_small_leaves(oracle):
    L = oracle.lib
    a = np.array([5, P + 1, 7], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    L.orc_poseidon_hash_or_noop(a, 3, out)
    assert out.tolist() == [5, 1, 7, 0]
    a5 = np.arange(5, dtype=np.uint64)
    L.orc_poseidon_hash_or_noop(a5, 5, out)
    st = oracle.poseidon_permute(list(range(5)) + [0] * 7)
    assert out.tolist() == st[:4].tolist()
    # 9 elements: two permutations, second absorb overwrites only lane 0
    a9 = np.arange(1, 10, dtype=np.uint64)
    L.orc_poseidon_hash_or_noop(a9, 9, out)
    st = oracle.poseidon_permute(list(range(1, 9)) + [0] * 4)
    st[0] = 9
    st = oracle.poseidon_permute(st)
    assert out.tolist() == st[:4].tolist()
