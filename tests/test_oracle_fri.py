"""Oracle FRI prover vs oracle FRI verifier (restating plonky2's verify_fri_proof): no reference
golden vector pins FRI, so prover/verifier consistency + tamper rejection is the available check."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_lib import (P, make_cfg, new_challenger, oracle_fri_prove, oracle_fri_verify,
                              setup_fri_api, splitmix64, stark_fri_instance)


def _setup(oracle, degree_bits, hasher, n_trace, n_aux, n_quot, seed, with_ctl=False, **cfgkw):
    setup_fri_api(oracle)
    L = oracle.lib
    n = 1 << degree_bits
    cfg = make_cfg(hasher=hasher, **cfgkw)
    commits = []
    for k, c in enumerate([n_trace, n_aux, n_quot]):
        if c == 0:
            continue
        vals = np.stack([splitmix64(seed * 100 + k * 1000 + j, n) for j in range(c)])
        commits.append(oracle.commit_values(vals, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height,
                                            hasher=hasher))
    ch = new_challenger(oracle, hasher)
    for r in commits:
        L.orc_challenger_observe_cap(C.byref(ch), r["cap"], r["cap"].shape[0])
    zeta = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(ch), zeta)
    w = L.orc_gl_root_of_unity(degree_bits)
    gz = (L.orc_gl_mul(int(zeta[0]), w), L.orc_gl_mul(int(zeta[1]), w))
    inst = stark_fri_instance((int(zeta[0]), int(zeta[1])), gz, n_trace, n_aux, n_quot,
                              ctl_zs_range=(max(n_aux - 2, 0), n_aux) if with_ctl else None)
    return cfg, commits, ch, inst


@pytest.mark.parametrize("hasher", [0, 1])
@pytest.mark.parametrize("degree_bits,n_trace,n_aux,n_quot,with_ctl,kw", [
    (6, 5, 3, 2, True, dict(pow_bits=4, queries=5)),
    (9, 12, 4, 4, True, dict(pow_bits=6, queries=7)),
    (10, 7, 0, 2, False, dict(pow_bits=3, queries=4)),
    (5, 3, 2, 2, False, dict(pow_bits=1, queries=1)),          # TEST_STARK_CONFIG-like
    (13, 4, 2, 4, True, dict(pow_bits=8, queries=10)),
])
def test_prove_then_verify(oracle, hasher, degree_bits, n_trace, n_aux, n_quot, with_ctl, kw):
    cfg, commits, ch, inst = _setup(oracle, degree_bits, hasher, n_trace, n_aux, n_quot, 3, with_ctl, **kw)
    ch_v = type(ch).from_buffer_copy(ch)
    opn, proof = oracle_fri_prove(oracle, cfg, degree_bits, commits, inst, ch)
    caps = [r["cap"] for r in commits]
    cols = [r["coeffs"].shape[0] for r in commits]
    ok, why = oracle_fri_verify(oracle, cfg, degree_bits, caps, cols, inst, opn, proof,
                                type(ch).from_buffer_copy(ch_v))
    assert ok == 1, why
    # proof-of-work witness is the smallest valid one and the shape header is right
    L = oracle.lib
    ab = (C.c_uint32 * 32)()
    R = L.orc_fri_reduction_arity_bits(degree_bits, C.byref(cfg), ab, 32)
    assert int(proof[0]) == R and int(proof[2]) == kw["queries"]
    # tampering anywhere must be rejected
    rng = np.random.default_rng(1)
    for _ in range(12):
        bad = proof.copy()
        i = int(rng.integers(6 + R + len(cols), bad.size))
        bad[i] ^= np.uint64(1)
        ok2, _ = oracle_fri_verify(oracle, cfg, degree_bits, caps, cols, inst, opn, bad,
                                   type(ch).from_buffer_copy(ch_v))
        # flipping a bit of an unused leaf column is still caught by the Merkle check
        assert ok2 == 0, i
    bad_opn = opn.copy()
    bad_opn[0] ^= np.uint64(1)
    ok3, _ = oracle_fri_verify(oracle, cfg, degree_bits, caps, cols, inst, bad_opn, proof,
                               type(ch).from_buffer_copy(ch_v))
    assert ok3 == 0


def test_reduction_arity_bits(oracle):
    setup_fri_api(oracle)
    cfg = make_cfg()
    ab = (C.c_uint32 * 32)()
    # SURVEY 8(c'): degree_bits 20 -> [4,4,4,4], final poly 2^4
    assert oracle.lib.orc_fri_reduction_arity_bits(20, C.byref(cfg), ab, 32) == 4
    assert list(ab[:4]) == [4, 4, 4, 4]
    assert oracle.lib.orc_fri_reduction_arity_bits(5, C.byref(cfg), ab, 32) == 0
    # degree_bits + rate_bits - arity_bits >= cap_height must also hold: 6 + 1 - 4 < 4
    assert oracle.lib.orc_fri_reduction_arity_bits(6, C.byref(cfg), ab, 32) == 0
    assert oracle.lib.orc_fri_reduction_arity_bits(7, C.byref(cfg), ab, 32) == 1
    assert oracle.lib.orc_fri_reduction_arity_bits(16, C.byref(cfg), ab, 32) == 3


def test_challenger_semantics(oracle):
    setup_fri_api(oracle)
    L = oracle.lib
    ch = new_challenger(oracle, 0)
    # fresh challenger: first challenge = state[7] of permute(0)
    st = oracle.poseidon_permute([0] * 12)
    assert L.orc_challenger_get(C.byref(ch)) == int(st[7])
    assert L.orc_challenger_get(C.byref(ch)) == int(st[6])
    # observing clears the output buffer; 3 inputs then a challenge -> duplex with overwrite
    e = np.array([5, 6, 7], dtype=np.uint64)
    L.orc_challenger_observe(C.byref(ch), e, 3)
    st2 = st.copy()
    st2[:3] = e
    st2 = oracle.poseidon_permute(st2)
    assert L.orc_challenger_get(C.byref(ch)) == int(st2[7])
    # compact returns the state and drops buffered outputs
    out = np.zeros(12, dtype=np.uint64)
    L.orc_challenger_compact(C.byref(ch), out)
    assert out.tolist() == st2.tolist()
    st3 = oracle.poseidon_permute(st2)
    assert L.orc_challenger_get(C.byref(ch)) == int(st3[7])
    # keccak hash -> 7,7,7,4-byte elements
    slot = np.frombuffer(bytes(range(1, 26)) + bytes(7), dtype=np.uint64).copy()
    el = np.zeros(4, dtype=np.uint64)
    L.orc_hash_to_elements.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.uint64), np.ctypeslib.ndpointer(np.uint64)]
    L.orc_hash_to_elements(1, slot, el)
    b = bytes(range(1, 26))
    assert el.tolist() == [int.from_bytes(b[0:7], "little"), int.from_bytes(b[7:14], "little"),
                           int.from_bytes(b[14:21], "little"), int.from_bytes(b[21:25], "little")]
