"""-m gpu: openings + FRI on the GPU (value-domain combination, bit-reversed folding) must be
word-for-word identical to the oracle's coefficient-domain restatement of plonky2, and the
oracle's verify_fri_proof restatement must accept the GPU proof."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_lib import (make_cfg, new_challenger, oracle_fri_prove, oracle_fri_verify,
                              setup_fri_api, splitmix64)
from tests.oracle_lib import stark_fri_instance as oracle_instance

pytestmark = pytest.mark.gpu


def _host_challenger_matches_oracle(oracle, hasher):
    from zk_evm_amd import Challenger
    setup_fri_api(oracle)
    L = oracle.lib
    och = new_challenger(oracle, hasher)
    ch = Challenger(hasher)
    rng = np.random.default_rng(9)
    for step in range(40):
        k = int(rng.integers(0, 12))
        e = rng.integers(0, 1 << 64, size=k, dtype=np.uint64)
        if k:
            L.orc_challenger_observe(C.byref(och), e, k)
            ch.observe_elements(e)
        if step % 3 == 0:
            cap = rng.integers(0, 1 << 63, size=(4, 4), dtype=np.uint64)
            if hasher == 1:
                cap[:, 3] &= np.uint64(0xFF)
            L.orc_challenger_observe_cap(C.byref(och), cap, 4)
            ch.observe_cap(cap)
        for _ in range(int(rng.integers(0, 11))):
            assert ch.get_challenge() == L.orc_challenger_get(C.byref(och))
        if step % 7 == 0:
            st = np.zeros(12, dtype=np.uint64)
            L.orc_challenger_compact(C.byref(och), st)
            assert np.array_equal(ch.compact(), st)
    e2 = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(och), e2)
    assert ch.get_extension_challenge() == (int(e2[0]), int(e2[1]))


@pytest.mark.parametrize("hasher", [0, 1])
def test_challenger_matches_oracle(oracle, hasher):
    _host_challenger_matches_oracle(oracle, hasher)


@pytest.mark.parametrize("hasher", [0, 1])
@pytest.mark.parametrize("degree_bits,n_trace,n_aux,n_quot,with_ctl,kw", [
    (6, 5, 3, 2, True, dict(pow_bits=4, queries=5)),
    (5, 3, 2, 2, False, dict(pow_bits=1, queries=1)),
    (9, 12, 4, 4, True, dict(pow_bits=6, queries=7)),
    (10, 7, 0, 2, False, dict(pow_bits=3, queries=4)),
    (13, 9, 5, 4, True, dict(pow_bits=10, queries=12)),
    (16, 6, 2, 4, True, dict(pow_bits=16, queries=84)),   # production FRI parameters
    # the FRI shape of the recursion layer (`standard_recursion_config`, rate_bits 3, 28 queries)
    (12, 7, 3, 2, True, dict(pow_bits=16, queries=28, rate_bits=3)),
    (13, 4, 0, 8, False, dict(pow_bits=8, queries=9, rate_bits=2)),
])
def test_prove_openings_matches_oracle(oracle, hasher, degree_bits, n_trace, n_aux, n_quot, with_ctl, kw):
    import zk_evm_amd as zk
    setup_fri_api(oracle)
    L = oracle.lib
    n = 1 << degree_bits
    kw = dict(kw)
    rate_bits = kw.pop("rate_bits", 1)
    cfg = make_cfg(hasher=hasher, rate_bits=rate_bits, **kw)
    seed = 11
    mats = []
    for k, c in enumerate([n_trace, n_aux, n_quot]):
        if c:
            mats.append(np.stack([splitmix64(seed * 100 + k * 1000 + j, n) for j in range(c)]))
    commits = [oracle.commit_values(m, rate_bits=rate_bits, cap_height=4, hasher=hasher) for m in mats]
    batches = [zk.PolynomialBatch.from_values(m, rate_bits, False, 4, hasher=hasher) for m in mats]
    for r, b in zip(commits, batches):
        assert np.array_equal(b.merkle_tree.cap.elements, r["cap"])
    # transcript up to zeta, both sides
    och = new_challenger(oracle, hasher)
    ch = zk.Challenger(hasher)
    for r in commits:
        L.orc_challenger_observe_cap(C.byref(och), r["cap"], 16)
        ch.observe_cap(r["cap"])
    zeta = ch.get_extension_challenge()
    z2 = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(och), z2)
    assert zeta == (int(z2[0]), int(z2[1]))
    w = L.orc_gl_root_of_unity(degree_bits)
    gz = (L.orc_gl_mul(zeta[0], w), L.orc_gl_mul(zeta[1], w))
    ctl = (max(n_aux - 2, 0), n_aux) if with_ctl else None
    oinst = oracle_instance(zeta, gz, n_trace, n_aux, n_quot, ctl_zs_range=ctl)
    inst = zk.stark_fri_instance(zeta, gz, n_trace, n_aux, n_quot, ctl_zs_range=ctl)
    och_v = type(och).from_buffer_copy(och)
    o_open, o_proof = oracle_fri_prove(oracle, cfg, degree_bits, commits, oinst, och)
    # GPU
    g_open = zk.fri_openings(inst, batches)
    assert np.array_equal(g_open.reshape(-1), o_open)
    ch.observe_extension_elements(g_open)
    scfg = zk.StarkConfig(hasher=hasher, fri_config=zk.FriConfig(rate_bits=rate_bits, proof_of_work_bits=kw["pow_bits"],
                                                                 num_query_rounds=kw["queries"]))
    g_proof = zk.prove_openings(inst, batches, ch, scfg, g_open)
    assert g_proof.shape == o_proof.shape
    bad = np.nonzero(g_proof != o_proof)[0]
    assert bad.size == 0, (bad[:8], g_proof[bad[:4]], o_proof[bad[:4]])
    # transcripts stay in lock step afterwards
    assert ch.get_challenge() == L.orc_challenger_get(C.byref(och))
    # and the oracle's verifier accepts the GPU proof
    ok, why = oracle_fri_verify(oracle, cfg, degree_bits, [r["cap"] for r in commits],
                                [m.shape[0] for m in mats], oinst, g_open.reshape(-1).copy(), g_proof,
                                type(och).from_buffer_copy(och_v))
    assert ok == 1, why


def test_fri_rejects_bad_arguments():
    import zk_evm_amd as zk
    b1 = zk.PolynomialBatch.from_values(np.zeros((2, 16), np.uint64), 1, False, 2)
    b2 = zk.PolynomialBatch.from_values(np.zeros((2, 32), np.uint64), 1, False, 2)
    inst = zk.FriInstanceInfo([zk.FriBatchInfo((3, 4), [(0, 0), (1, 1)])])
    with pytest.raises(zk.ZkStarkError):
        zk.fri_openings(inst, [b1, b2])          # mixed degrees
    inst2 = zk.FriInstanceInfo([zk.FriBatchInfo((3, 4), [(0, 5)])])
    with pytest.raises(zk.ZkStarkError):
        zk.fri_openings(inst2, [b1])             # polynomial index out of range
