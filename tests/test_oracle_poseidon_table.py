"""CPU: the cdk_erigon Poseidon table restatement (oracle/poseidon_table.py).  Mirrors the reference's own tests for
the table (poseidon_stark.rs:920-1004): `poseidon_correctness_test` (the digest columns of a general operation equal
`poseidon_hash_padded_byte_vec` of its input -- here additionally pinned to the reference's published
`hash_contract_bytecode` vector, smt_trie/src/code.rs:56-84) and `test_stark_degree`; plus the generate <-> eval
consistency every other table is held to."""
import numpy as np

from oracle import poseidon_table as pt
from oracle import stark as S
from tests.test_oracle_kat import SOME_CODE, _hash_contract_bytecode
from tests.test_oracle_tracegen import _check_air

P = pt.P


def pad(code: bytes) -> bytes:                      # poseidon_pad_byte_vec, smt_trie/src/code.rs:38-44
    b = bytearray(code) + b"\x01"
    while len(b) % 56:
        b.append(0)
    b[-1] |= 0x80
    return bytes(b)


def sample_ops(rng):
    ops = [("simple", [int(x) for x in rng.integers(0, P, 12, dtype=np.uint64)]),
           ("general", (3, 7, 1000), 55, pad(SOME_CODE), len(pad(SOME_CODE))),
           ("simple", [0] * 12),
           ("general", (0, 1, 0), 56, pad(b""), 56),
           ("general", (1, 2, 3), 90, rng.bytes(56 * 3), 56 * 3),
           ("simple", [P - 1] * 12)]
    return ops


def digest_of(row):
    return [int(row[pt.DIGEST_COL + 2 * i]) + (int(row[pt.DIGEST_COL + 2 * i + 1]) << 32) for i in range(4)]


def test_rows_satisfy_the_air_and_digests_match(oracle):
    rng = np.random.default_rng(5)
    ops = sample_ops(rng)
    t = pt.generate_trace(ops, 8)
    assert t.shape == (pt.NUM_COLUMNS, 32)
    _check_air(pt.eval_poseidon, t)
    rows = t.T
    # simple op: the whole permutation output is on the row
    st = oracle.poseidon_permute(ops[0][1])
    assert digest_of(rows[0]) == [int(x) for x in st[:4]]
    assert [int(v) for v in rows[0][pt.OUTPUT_PARTIAL:pt.OUTPUT_PARTIAL + 8]] == [int(x) for x in st[4:]]
    # general op over the padded 574-byte contract: 11 rows, last digest == the reference's KAT
    n_blocks = len(pad(SOME_CODE)) // 56
    last = rows[1 + n_blocks - 1]
    assert digest_of(last) == _hash_contract_bytecode(oracle, SOME_CODE) == [
        13311281292453978464, 8384462470517067887, 14733964407220681187, 13541155386998871195]
    assert int(rows[1][pt.IS_FIRST_ROW_GENERAL_OP]) == 1 and int(last[pt.IS_FINAL_INPUT_LEN]) == 1
    assert [int(rows[1 + k][pt.ALREADY_ABSORBED]) for k in range(n_blocks)] == [56 * k for k in range(n_blocks)]
    # empty code
    r = rows[1 + n_blocks + 1]
    assert digest_of(r) == [10052403398432742521, 15195891732843337299, 2019258788108304834, 4300613462594703212]
    # padding rows: permutation of zeros, no flag
    assert not any(int(t[c, -1]) for c in (pt.NOT_PADDING, pt.IS_SIMPLE_OP, pt.IS_FULL_INPUT_BLOCK)) and int(t[pt.CUBED_FULL, -1]) != 0


def test_corrupted_rows_are_caught():
    rng = np.random.default_rng(6)
    t = pt.generate_trace(sample_ops(rng), 8)
    for col, row in ((pt.PARTIAL_SBOX + 7, 2), (pt.CUBED_FULL + 50, 0), (pt.DIGEST_COL + 1, 3), (pt.ALREADY_ABSORBED, 2),
                     (pt.INPUT + 9, 4)):
        bad = t.copy()
        bad[col, row] ^= np.uint64(1)
        try:
            _check_air(pt.eval_poseidon, bad)
        except AssertionError:
            continue
        raise AssertionError(f"corruption of column {col} row {row} not detected")


def test_stark_degree():
    """poseidon_stark.rs:932-944 (`test_stark_low_degree`), as tests/test_oracle_stark_degree.py does for the nine
    eth_mainnet tables."""
    from tests.test_oracle_stark_degree import _degree_of_values, _interp_eval
    rng = np.random.default_rng(7)
    n, N = 4, 16
    w_n, w_N = S.root_of_unity(2), S.root_of_unity(4)
    xs = [7 * pow(w_N, j, P) % P for j in range(N)]
    trace = [_interp_eval([int(v) for v in rng.integers(0, P, n, dtype=np.uint64)], w_n, xs) for _ in range(pt.NUM_COLUMNS)]
    alpha = int(rng.integers(1, P, dtype=np.uint64))
    last, ninv = pow(w_n, P - 2, P), pow(n, P - 2, P)
    vals = []
    for j, x in enumerate(xs):
        zh = (pow(x, n, P) - 1) % P
        lf = zh * ninv % P * pow((x - 1) % P, P - 2, P) % P
        ll = zh * ninv % P * pow((x * w_n - 1) % P, P - 2, P) % P
        cons = S.ConstraintConsumer([alpha], (x - last) % P, lf, ll)
        pt.eval_poseidon([c[j] for c in trace], [c[(j + 4) % N] for c in trace], cons)
        vals.append(cons.accs[0])
    assert 9 <= _degree_of_values(vals, xs) < 3 * n
