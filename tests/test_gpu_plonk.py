"""-m gpu: the PLONK prover slice (zk_plonk_prove) against the oracle's restatement of plonky2's
`prove` (oracle/plonk.py), word for word -- wires / Zs+partial-products / quotient caps, the whole opening set, the FRI
proof over the four oracles, the transcript state afterwards -- on circuits of 2^6 .. 2^14 rows under
`standard_recursion_config` (135 wires, 80 routed, rate_bits 3, quotient degree factor 8), and the oracle's verifier
accepts the device proof."""
import ctypes as C

import numpy as np
import pytest

import tests.oracle_lib as ol
from oracle import plonk as PK

pytestmark = pytest.mark.gpu


def _device_circuit(circ):
    import zk_evm_amd.plonk as zp
    from tests.gpu_util import to_dev
    cfg = circ.config
    pcfg = zp.CircuitConfig(num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires, num_challenges=cfg.num_challenges,
                            rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, proof_of_work_bits=cfg.proof_of_work_bits,
                            num_query_rounds=cfg.num_query_rounds, arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits,
                            hasher=cfg.hasher)
    cs = to_dev(np.concatenate([circ.constants, circ.sigmas]))
    return zp.CircuitData(pcfg, circ.degree_bits, circ.gate_descriptors(), circ.num_selectors, cs, circ.k_is,
                          circ.circuit_digest, circ.num_gate_constraints, circ.quotient_degree_factor)


@pytest.mark.parametrize("degree_bits,seed,kw", [(6, 1, dict(proof_of_work_bits=3, num_query_rounds=4)),
                                                 (9, 2, dict(proof_of_work_bits=5, num_query_rounds=7)),
                                                 (12, 3, dict()),           # standard_recursion_config: 28 queries, 16 PoW bits
                                                 (14, 4, dict(num_query_rounds=12))])
def test_plonk_proof_matches_oracle(oracle, degree_bits, seed, kw):
    from tests.gpu_util import to_dev
    ol.setup_fri_api(oracle)
    circ, wires, pis = PK.build_arithmetic_circuit(degree_bits, seed=seed, cfg=PK.CircuitConfig(**kw))
    wires, pi_hash = PK.set_public_input_wires(oracle, circ, wires, pis)
    exp = PK.prove(oracle, ol, circ, wires, pis)
    cd = _device_circuit(circ)
    assert np.array_equal(cd.constants_sigmas_cap(), PK.commit_circuit(oracle, circ)["cap"])
    got = cd.prove(to_dev(wires), pis)
    assert got.public_inputs_hash == pi_hash
    assert np.array_equal(got.wires_cap, exp["wires_cap"])
    assert np.array_equal(got.plonk_zs_partial_products_cap, exp["zs_pp_cap"])
    assert np.array_equal(got.quotient_polys_cap, exp["quotient_cap"])
    assert np.array_equal(got.openings.reshape(-1), exp["openings"])
    assert np.array_equal(got.opening_proof, exp["fri"])
    ok, why = PK.verify(oracle, ol, circ, dict(wires_cap=got.wires_cap, zs_pp_cap=got.plonk_zs_partial_products_cap,
                                               quotient_cap=got.quotient_polys_cap, openings=got.openings,
                                               fri=got.opening_proof, public_inputs=pis))
    assert ok, why
    # the same circuit proves a second witness (other public inputs) without re-committing constants / sigmas
    pis2 = [p ^ 5 for p in pis]
    w2, _ = PK.set_public_input_wires(oracle, circ, wires.copy(), pis2)
    got2 = cd.prove(to_dev(w2), pis2)
    exp2 = PK.prove(oracle, ol, circ, w2, pis2)
    assert np.array_equal(got2.opening_proof, exp2["fri"]) and not np.array_equal(got2.wires_cap, got.wires_cap)
    cd.free()


def test_plonk_rejects_bad_circuit_descriptions():
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.plonk as zp
    cs = torch.zeros((83, 64), dtype=torch.int64, device="cuda")
    good = [(0, 0, 0, 0, 4), (1, 2, 0, 0, 4), (2, 0, 0, 0, 4), (3, 20, 0, 0, 4)]
    with pytest.raises(zk.ZkStarkError, match="gate kind 99"):
        zp.CircuitData(zp.CircuitConfig(), 6, good[:3] + [(99, 0, 0, 0, 4)], 1, cs, [1] * 80, [0] * 4, 4)
    with pytest.raises(zk.ZkStarkError, match="num_gate_constraints"):
        zp.CircuitData(zp.CircuitConfig(), 6, good, 1, cs, [1] * 80, [0] * 4, 7)
    with pytest.raises(zk.ZkStarkError, match="selector group"):
        zp.CircuitData(zp.CircuitConfig(), 6, [(0, 0, 0, 1, 4)] + good[1:], 1, cs, [1] * 80, [0] * 4, 20)
    cd = zp.CircuitData(zp.CircuitConfig(), 6, good, 1, cs, [1] * 80, [0] * 4, 20)
    with pytest.raises(zk.ZkStarkError, match="witness must be"):
        cd.prove(torch.zeros((135, 32), dtype=torch.int64, device="cuda"), [])


@pytest.mark.parametrize("degree_bits,seed,kw", [(8, 31, dict(proof_of_work_bits=4, num_query_rounds=5)), (12, 32, dict())])
def test_plonk_mixed_gates_match_oracle(oracle, degree_bits, seed, kw):
    """Fourteen gate kinds in four selector groups -- extension arithmetic, BaseSum, Reducing(+Extension), Exponentiation, RandomAccess, PoseidonMds, CosetInterpolation,
    Poseidon (123 constraints, plain-round evaluation) next to the four base gates -- with a VALID witness: device proof ==
    oracle proof word for word, and the oracle's verifier accepts it."""
    from tests.gpu_util import to_dev
    ol.setup_fri_api(oracle)
    circ, wires, pis = PK.build_mixed_circuit(degree_bits, seed=seed, cfg=PK.CircuitConfig(**kw))
    wires, pi_hash = PK.set_public_input_wires(oracle, circ, wires, pis)
    exp = PK.prove(oracle, ol, circ, wires, pis)
    cd = _device_circuit(circ)
    got = cd.prove(to_dev(wires), pis)
    assert np.array_equal(got.plonk_zs_partial_products_cap, exp["zs_pp_cap"])
    assert np.array_equal(got.quotient_polys_cap, exp["quotient_cap"])
    assert np.array_equal(got.openings.reshape(-1), exp["openings"])
    assert np.array_equal(got.opening_proof, exp["fri"])
    ok, why = PK.verify(oracle, ol, circ, dict(wires_cap=got.wires_cap, zs_pp_cap=got.plonk_zs_partial_products_cap,
                                               quotient_cap=got.quotient_polys_cap, openings=got.openings,
                                               fri=got.opening_proof, public_inputs=pis))
    assert ok, why
    # random (unsatisfying) wires exercise every constraint with non-zero values on both sides
    rng = np.random.default_rng(seed + 1)
    junk = rng.integers(0, 1 << 64, size=wires.shape, dtype=np.uint64)
    got2 = cd.prove(to_dev(junk), pis)
    exp2 = PK.prove(oracle, ol, circ, junk, pis)
    assert np.array_equal(got2.quotient_polys_cap, exp2["quotient_cap"]) and np.array_equal(got2.opening_proof, exp2["fri"])
    cd.free()


@pytest.mark.parametrize("degree_bits,mixed", [(8, False), (10, True)])
def test_plonk_keccak_config_matches_oracle(oracle, degree_bits, mixed):
    """`KeccakGoldilocksConfig`: Keccak Merkle trees and challenger (hash-onion permutation, device-side proof-of-work grind),
    Poseidon for the public-input hash (C::InnerHasher); device proof == oracle proof."""
    from tests.gpu_util import to_dev
    ol.setup_fri_api(oracle)
    build = PK.build_mixed_circuit if mixed else PK.build_arithmetic_circuit
    circ, wires, pis = build(degree_bits, seed=5, cfg=PK.CircuitConfig(hasher=1, proof_of_work_bits=9, num_query_rounds=6))
    wires, _ = PK.set_public_input_wires(oracle, circ, wires, pis)
    exp = PK.prove(oracle, ol, circ, wires, pis)
    cd = _device_circuit(circ)
    got = cd.prove(to_dev(wires), pis)
    assert np.array_equal(got.wires_cap, exp["wires_cap"]) and np.array_equal(got.quotient_polys_cap, exp["quotient_cap"])
    assert np.array_equal(got.openings.reshape(-1), exp["openings"]) and np.array_equal(got.opening_proof, exp["fri"])
    cd.free()



def test_plonk_prove_batch_equals_single_proofs(oracle):
    """zk_plonk_prove_batch: twelve witnesses of one mixed-gate circuit through four worker contexts of the library, from one
    call -- every proof equals `prove` of the same witness word for word (and the first the oracle's), a second batch reuses
    the workers, and a bad witness fails the whole call."""
    import torch
    from tests.gpu_util import to_dev
    from zk_evm_amd._lib import ZkStarkError
    ol.setup_fri_api(oracle)
    circ, wires, pis = PK.build_mixed_circuit(10, seed=41, cfg=PK.CircuitConfig(proof_of_work_bits=6, num_query_rounds=6))
    wires, _ = PK.set_public_input_wires(oracle, circ, wires, pis)
    cd = _device_circuit(circ)
    rng = np.random.default_rng(9)
    ws, ps = [wires], [pis]
    for k in range(11):                                  # junk wires: every constraint non-zero; own public inputs
        p = [int(x) for x in rng.integers(0, 1 << 60, size=len(pis))]
        w = rng.integers(0, 1 << 64, size=wires.shape, dtype=np.uint64)
        w, _ = PK.set_public_input_wires(oracle, circ, w, p)
        ws.append(w)
        ps.append(p)
    dev = [to_dev(w) for w in ws]
    single = [cd.prove(d, p) for d, p in zip(dev, ps)]
    exp0 = PK.prove(oracle, ol, circ, ws[0], ps[0])
    assert np.array_equal(single[0].opening_proof, exp0["fri"])
    for rep in range(2):
        batch = cd.prove_batch(dev, ps, in_flight=4)
        assert len(batch) == 12
        for a, b in zip(single, batch):
            assert np.array_equal(a.wires_cap, b.wires_cap) and np.array_equal(a.quotient_polys_cap, b.quotient_polys_cap)
            assert np.array_equal(a.openings, b.openings) and np.array_equal(a.opening_proof, b.opening_proof)
            assert a.public_inputs_hash == b.public_inputs_hash
    assert cd.prove_batch([], []) == []
    with pytest.raises(ZkStarkError):
        cd.prove_batch(dev[:2] + [torch.zeros((135, 32), dtype=torch.int64, device="cuda")], ps[:3])
    cd.free()
