"""-m gpu: the device Arithmetic-table generator (zk_arithmetic_generate_trace: fixed-width restatement of
`Operation::to_rows` and its per-operation generators) against oracle/arith_trace.py (Python big integers), cell for
cell; then the device-generated table is proven and accepted by the oracle verifier."""
import numpy as np
import pytest

from oracle import arith_trace as at

pytestmark = pytest.mark.gpu


def _to_product(ops):
    out = []
    for op in ops:
        if op[0] == "range_check":
            out.append((16,) + tuple(op[1:]))
        else:
            out.append(tuple(op[1:]))
    return out


def _check(ops):
    from zk_evm_amd.tracegen import arithmetic_generate_trace
    exp, n_rows = at.generate_trace(ops)
    trace, used = arithmetic_generate_trace(_to_product(ops))
    got = trace.cpu().numpy().view(np.uint64)
    assert used == n_rows and got.shape == exp.shape
    for c in range(116):
        bad = np.nonzero(got[c] != exp[c])[0]
        assert bad.size == 0, ("column", c, "rows", bad[:5], got[c, bad[:3]], exp[c, bad[:3]])
    return trace


def test_every_operation_kind_and_edge_case():
    from tests.test_oracle_tracegen import sample_arith_ops
    for seed in (3, 51):
        _check(sample_arith_ops(np.random.default_rng(seed)))
    m = (1 << 256) - 1
    edge = [("bin", at.IS_ADD, m, m), ("bin", at.IS_ADD, m, 1), ("bin", at.IS_SUB, 0, m), ("bin", at.IS_SUB, 5, 5),
            ("bin", at.IS_LT, 5, 5), ("bin", at.IS_GT, 5, 5), ("bin", at.IS_MUL, m, m), ("bin", at.IS_MUL, 0, 0),
            ("bin", at.IS_DIV, m, 1), ("bin", at.IS_DIV, 1, m), ("bin", at.IS_DIV, m, m), ("bin", at.IS_MOD, m, m - 1),
            ("bin", at.IS_DIV, 0, 0), ("bin", at.IS_MOD, 0, 0), ("bin", at.IS_DIV, 1 << 255, 1 << 128),
            ("ter", at.IS_MULMOD, m, m, 0), ("ter", at.IS_MULMOD, m, m, 1), ("ter", at.IS_MULMOD, m, m, m),
            ("ter", at.IS_MULMOD, m, m, 1 << 255), ("ter", at.IS_ADDMOD, m, m, m), ("ter", at.IS_ADDMOD, m, m, 0),
            ("ter", at.IS_ADDMOD, m, m, 3), ("ter", at.IS_SUBMOD, 0, m, 1), ("ter", at.IS_SUBMOD, 0, m, 0),
            ("ter", at.IS_SUBMOD, 0, m, m), ("ter", at.IS_SUBMOD, 0, m, 7), ("ter", at.IS_SUBMOD, 3, 3, 7),
            ("ter", at.IS_SUBMOD, 0, 14, 7), ("ter", at.IS_SUBMOD, m, 0, m - 1),
            ("bin", at.IS_SUBFP254, 0, at.BN_BASE - 1), ("bin", at.IS_ADDFP254, at.BN_BASE - 1, at.BN_BASE - 1),
            ("bin", at.IS_MULFP254, at.BN_BASE - 1, at.BN_BASE - 1), ("bin", at.IS_MULFP254, 0, 0),
            ("bin", at.IS_SHL, 255, m), ("bin", at.IS_SHR, 255, m), ("bin", at.IS_SHL, 64, m), ("bin", at.IS_SHR, 64, m),
            ("bin", at.IS_SHL, 63, m), ("bin", at.IS_SHR, 65, m), ("bin", at.IS_SHL, m, m), ("bin", at.IS_SHR, m, m),
            ("bin", at.IS_SHL, 128, 1), ("bin", at.IS_SHR, 0, 0),
            ("bin", at.IS_BYTE, 0, m), ("bin", at.IS_BYTE, 31, 0x0102), ("bin", at.IS_BYTE, 1 << 16, m),
            ("bin", at.IS_BYTE, 1 << 255, m), ("bin", at.IS_BYTE, 33, m),
            ("range_check", m, 0, 1, 0xFF, m)]
    _check(edge)
    _check([])


def random_operations(rng, count):
    def r(bits): return int.from_bytes(rng.bytes(32), "little") >> (256 - bits) if bits else 0
    sizes = (0, 1, 16, 17, 64, 65, 128, 200, 255, 256)
    ops = []
    for _ in range(count):
        f = int(rng.integers(0, 16))
        a, b, c = (r(int(rng.choice(sizes))) for _ in range(3))
        if f in (at.IS_ADDMOD, at.IS_MULMOD, at.IS_SUBMOD):
            ops.append(("ter", f, a, b, c))
        elif f in (at.IS_ADDFP254, at.IS_MULFP254, at.IS_SUBFP254):
            ops.append(("bin", f, a % at.BN_BASE, b % at.BN_BASE))
        elif f in (at.IS_SHL, at.IS_SHR):
            ops.append(("bin", f, int(rng.integers(0, 300)), b))
        elif f == at.IS_BYTE:
            ops.append(("bin", f, int(rng.integers(0, 40)), b))
        else:
            ops.append(("bin", f, a, b))
    return ops


def test_many_random_operations():
    _check(random_operations(np.random.default_rng(8), 1500))


def test_generated_arithmetic_table_is_proven_and_accepted(oracle):
    from tests.test_gpu_stark_verify import _prove_and_verify, _registry_descs
    from tests.test_oracle_tracegen import sample_arith_ops
    trace = _check(sample_arith_ops(np.random.default_rng(77)))
    zlist, lookups = _registry_descs(0)
    ok, why = _prove_and_verify(oracle, 5, trace.cpu().numpy().view(np.uint64), 0, zlist, lookup_spec=lookups)
    assert ok, why


def test_error_paths():
    from zk_evm_amd import ZkStarkError
    from zk_evm_amd.tracegen import arithmetic_generate_trace
    with pytest.raises(ZkStarkError):
        arithmetic_generate_trace([(17, 1, 2)])
    with pytest.raises(ZkStarkError):
        arithmetic_generate_trace([(0, 1 << 256, 2)])


def test_packed_array_is_passed_through():
    import torch
    from tests.test_oracle_tracegen import sample_arith_ops
    from zk_evm_amd.tracegen import arithmetic_generate_trace
    ops = _to_product(sample_arith_ops(np.random.default_rng(5)))
    m64 = (1 << 64) - 1
    flat = np.zeros((len(ops), 18), dtype=np.uint64)
    for r, op in enumerate(ops):
        if op[0] == 16:
            _, a, b, c, opcode, res = op
        else:
            a, b, c, opcode, res = op[1], op[2], (op[3] if len(op) > 3 else 0), 0, 0
        flat[r, 0], flat[r, 1] = op[0], opcode
        for k, v in enumerate((a, b, c, res)):
            flat[r, 2 + 4 * k:6 + 4 * k] = [(v >> (64 * l)) & m64 for l in range(4)]
    t1, u1 = arithmetic_generate_trace(ops)
    t2, u2 = arithmetic_generate_trace(flat)
    assert torch.equal(t1, t2) and u1 == u2


def test_reference_basic_and_big_traces_on_device():
    """The reference's own `basic_trace` / `big_traces` (arithmetic_stark.rs:373-519) through the device generator:
    the asserted output limbs, the RANGE_MAX floor, and the doubling at RANGE_MAX two-row operations -- at the
    reference's full counts (2^16 MULs, 2^16 MULMODs)."""
    from tests.test_oracle_tracegen import check_reference_basic_trace, reference_basic_trace_ops
    from zk_evm_amd.tracegen import arithmetic_generate_trace
    ops, _ = reference_basic_trace_ops()
    trace = _check(ops)
    check_reference_basic_trace(trace.cpu().numpy().view(np.uint64), 14)
    rng = np.random.default_rng(7)
    words = rng.integers(0, 1 << 64, size=(1 << 16, 3, 4), dtype=np.uint64)
    r256 = lambda i, k: sum(int(words[i, k, j]) << (64 * j) for j in range(4))
    t, used = arithmetic_generate_trace([(at.IS_MUL, r256(i, 0), r256(i, 1)) for i in range(1 << 16)])
    assert tuple(t.shape) == (116, 1 << 16) and used == 1 << 16
    t, used = arithmetic_generate_trace([(at.IS_MULMOD, r256(i, 0), r256(i, 1), r256(i, 2)) for i in range(1 << 16)])
    assert tuple(t.shape) == (116, 1 << 17) and used == 1 << 17
    # spot-check the last MULMOD against Python integers: output register of its second... first row holds the result
    got = t.cpu().numpy().view(np.uint64)
    i = (1 << 16) - 1
    res = (r256(i, 0) * r256(i, 1)) % r256(i, 2) if r256(i, 2) else 0
    assert sum(int(got[at.OUT + k, 2 * i]) << (16 * k) for k in range(16)) == res
