"""CPU: the oracle's restatement of the plonky2 PLONK prover / verifier (oracle/plonk.py) is
self-consistent: proofs of valid circuits verify, and a changed witness cell, public input, opening or circuit constant
is rejected at the check plonky2 would fail.  (Parity unpinned: no reference-generated PLONK proof exists in the tree.)"""
import numpy as np
import pytest

import tests.oracle_lib as ol
from oracle import plonk as PK

P = PK.P


def _cfg(**kw):
    return PK.CircuitConfig(proof_of_work_bits=4, num_query_rounds=6, **kw)


@pytest.fixture(scope="module")
def proven(oracle):
    ol.setup_fri_api(oracle)
    circ, wires, pis = PK.build_arithmetic_circuit(7, seed=11, cfg=_cfg())
    wires, _ = PK.set_public_input_wires(oracle, circ, wires, pis)
    return circ, wires, pis, PK.prove(oracle, ol, circ, wires, pis)


def test_circuit_shape_matches_standard_recursion_config():
    circ, wires, _ = PK.build_arithmetic_circuit(6, seed=1)
    cfg = circ.config
    assert (cfg.num_wires, cfg.num_routed_wires, cfg.num_constants, cfg.num_challenges) == (135, 80, 2, 2)
    assert (cfg.rate_bits, cfg.cap_height, cfg.num_query_rounds, cfg.proof_of_work_bits) == (3, 4, 28, 16)
    assert [g.id for g in circ.gates] == ["NoopGate", "ConstantGate { num_consts: 2 }", "PublicInputGate",
                                          "ArithmeticGate { num_ops: 20 }"]
    assert circ.num_selectors == 1 and circ.groups == [(0, 4)]        # 3 + 4 - 1 <= 9: one selector polynomial
    assert circ.num_partial_products == 9 and circ.num_gate_constraints == 20
    assert circ.constants.shape == (3, 64) and circ.sigmas.shape == (80, 64) and wires.shape == (135, 64)
    # sigma is a permutation of the cells k_j * w^i
    w = PK.S.root_of_unity(6)
    ids = {circ.k_is[j] * pow(w, i, P) % P for j in range(80) for i in range(64)}
    assert {int(x) for x in circ.sigmas.reshape(-1)} == ids


def test_selector_groups_when_one_polynomial_is_not_enough():
    class G:
        def __init__(self, d, i): self.degree, self.id = d, i
    gates = sorted([G(7, "a"), G(6, "b"), G(3, "c"), G(1, "d"), G(0, "e")], key=lambda g: (g.degree, g.id))
    sel, idx, groups = PK.selector_polynomials(gates, [0, 1, 2, 3, 4, 0], 9)
    # greedy: a group grows while (its size + the next gate's degree) stays below max_degree = 9
    U = PK.UNUSED_SELECTOR
    assert groups == [(0, 3), (3, 5)] and idx == [0, 0, 0, 1, 1]
    assert sel[0] == [0, 1, 2, U, U, 0] and sel[1] == [U, U, U, 3, 4, U]


def test_valid_proof_verifies(oracle, proven):
    circ, wires, pis, proof = proven
    ok, why = PK.verify(oracle, ol, circ, proof)
    assert ok, why
    # Z starts at 1 and the grand product closes: Z(g^(n-1)) * (last row's quotient) == 1 is implied by the identity;
    # the partial products of a satisfied permutation are non-trivial
    assert int(proof["zs_pp"][0][0]) == 1 and int(proof["zs_pp"][1][0]) == 1
    assert len({int(x) for x in proof["zs_pp"][0]}) > 8


def test_tampering_is_rejected(oracle, proven):
    circ, wires, pis, proof = proven
    bad = dict(proof)
    bad["public_inputs"] = [pis[0] ^ 1] + pis[1:]
    ok, why = PK.verify(oracle, ol, circ, bad)
    assert not ok
    bad = dict(proof)
    o2 = np.array(proof["openings"], dtype=np.uint64).copy()
    o2[2 * (3 + 80 + 5)] ^= 1                                        # one wire opening
    bad["openings"] = o2
    ok, why = PK.verify(oracle, ol, circ, bad)
    assert not ok and "vanishing" in why
    # a witness that breaks a copy constraint: the quotient is no longer a polynomial of degree < 8n, so the proof the
    # (honest) prover code emits fails verification
    w2 = wires.copy()
    row = next(r for r in range(circ.n) if int(circ.constants[0][r]) == 3)
    w2[3, row] = (int(w2[3, row]) + 1) % P                           # an Arithmetic output: gate AND copy constraints
    p2 = PK.prove(oracle, ol, circ, w2, pis)
    ok, why = PK.verify(oracle, ol, circ, p2)
    assert not ok


def test_extension_arithmetic():
    a, b = PK.Ext(3, 5), PK.Ext(11, 7)
    assert (a * b) == PK.Ext(3 * 11 + 7 * 5 * 7, 3 * 7 + 5 * 11)
    assert a * a.inverse() == PK.Ext(1)
    assert (2 - a) == PK.Ext(P - 1, P - 5) and a.pow(5) == a * a * a * a * a


def test_mixed_gate_circuit_verifies_and_every_gate_bites(oracle):
    """Eleven gate kinds in three selector groups (extension arithmetic, base sums, reducing chains, exponentiation,
    Poseidon): the valid witness proves and verifies; breaking one cell of each new gate's row is rejected."""
    ol.setup_fri_api(oracle)
    circ, wires, pis = PK.build_mixed_circuit(7, seed=21, cfg=_cfg())
    assert circ.num_selectors == 4 and circ.num_gate_constraints == 123
    assert [g.KIND for g in circ.gates] == [0, 1, 12, 2, 6, 8, 7, 4, 3, 5, 9, 11, 13, 10]   # sorted by (degree, id string)
    assert circ.groups == [(0, 7), (7, 11), (11, 13), (13, 14)]  # greedy: size + next degree < 9
    g = next(g for g in circ.gates if g.KIND == 13)
    assert (g.degree, g.n_inter, g.num_constraints, g.start_inter + 4 * g.n_inter + 2) == (6, 2, 12, 47)
    wires, _ = PK.set_public_input_wires(oracle, circ, wires, pis)
    # every gate's constraints vanish on its own rows (generate <-> eval, the reference's test style)
    for r in range(circ.n):
        g = circ.gates[min(int(circ.constants[s][r]) for s in range(circ.num_selectors))]
        vals = g.eval_unfiltered([int(circ.constants[circ.num_selectors + k][r]) for k in range(2)],
                                 [int(wires[w][r]) for w in range(135)],
                                 [int(x) for x in oracle.poseidon_hash_no_pad(np.array(pis, dtype=np.uint64))])
        assert all(v % P == 0 for v in vals), (r, g.id)
    proof = PK.prove(oracle, ol, circ, wires, pis)
    ok, why = PK.verify(oracle, ol, circ, proof)
    assert ok, why
    for kind, col in ((4, 6), (5, 4), (6, 3), (7, 0), (8, 1), (9, 100), (10, 70), (11, 75), (12, 30), (13, 40)):
        gi = next(i for i, g in enumerate(circ.gates) if g.KIND == kind)
        row = next(r for r in range(circ.n) if min(int(circ.constants[s][r]) for s in range(circ.num_selectors)) == gi)
        w2 = wires.copy()
        w2[col, row] = (int(w2[col, row]) + 1) % P
        ok, why = PK.verify(oracle, ol, circ, PK.prove(oracle, ol, circ, w2, pis))
        assert not ok, (kind, col)


def test_coset_interpolation_gate_interpolates():
    """the gate's evaluation_value is the value at `evaluation_point` of the degree < 16 polynomial through the 16 values on
    shift * H (what FRI's arity-16 fold needs), not merely something self-consistent"""
    rng = np.random.default_rng(3)
    g = PK.CosetInterpolationGate(4, 8)
    coeffs = [PK.Ext(int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))) for _ in range(16)]

    def f(x):
        acc = PK.Ext(0)
        for c in reversed(coeffs):
            acc = acc * x + c
        return acc
    shift = int(rng.integers(1, P, dtype=np.uint64))
    point = PK.Ext(int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    vals = [f(PK.Ext(shift * h % P)) for h in g.domain]
    shifted = point * PK.Ext(pow(shift, P - 2, P))
    ev, prod = g._partial(0, 16, [(v.a, v.b) for v in vals], (shifted.a, shifted.b), (0, 0), (1, 0))
    assert PK.Ext(ev[0], ev[1]) == f(point)
