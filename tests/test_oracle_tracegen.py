"""CPU: the oracle's BytePacking / KeccakSponge generators (oracle/tracegen.py) against the restated AIRs -- the
reference's own generate<->eval test style: every constraint vanishes on every generated row --
and the sponge digest against keccak256."""
import numpy as np

from oracle import airs as oairs
from oracle import tracegen as otg

P = 0xFFFFFFFF00000001


def _check_air(air, t):
    n = t.shape[1]

    class Cons:
        def __init__(self, i): self.i, self.bad = i, 0
        def constraint(self, c): self.bad += 1 if c % P else 0
        def constraint_transition(self, c): self.bad += 1 if (self.i != n - 1 and c % P) else 0
        def constraint_first_row(self, c): self.bad += 1 if (self.i == 0 and c % P) else 0
        def constraint_last_row(self, c): self.bad += 1 if (self.i == n - 1 and c % P) else 0
    rows = t.T
    for i in range(n):
        c = Cons(i)
        air([int(v) for v in rows[i]], [int(v) for v in rows[(i + 1) % n]], c)
        assert c.bad == 0, (i, c.bad)


def _keccak_f(oracle):
    def f(words):
        a = np.array(words, dtype=np.uint64)
        oracle.lib.orc_keccak_f1600(a)
        return [int(x) for x in a]
    return f


def sample_byte_packing_ops(rng, n):
    ops = []
    for _ in range(n):
        ln = int(rng.integers(1, 33))
        ops.append((bool(rng.integers(0, 2)), (int(rng.integers(0, 5)), int(rng.integers(0, 30)), int(rng.integers(0, 1 << 20))),
                    int(rng.integers(1, 1 << 20)), rng.bytes(ln)))
    ops.insert(3, (True, (0, 0, 0), 9, b""))            # an empty op produces no row (byte_packing_stark.rs:209-213)
    return ops


def sample_sponge_ops(rng):
    lens = [0, 1, 55, 135, 136, 137, 272, 300]
    return [((int(rng.integers(0, 5)), int(rng.integers(0, 30)), int(rng.integers(0, 1 << 20))), int(rng.integers(1, 1 << 20)),
             rng.bytes(ln)) for ln in lens]


def test_byte_packing_rows_satisfy_the_air():
    rng = np.random.default_rng(1)
    t = otg.byte_packing_generate_trace(sample_byte_packing_ops(rng, 40), 16)
    assert t.shape == (71, 256)
    _check_air(oairs.eval_byte_packing, t)
    assert int(t[70].sum()) == 32 * 256 and np.array_equal(t[69], np.arange(256, dtype=np.uint64))


def test_keccak_sponge_rows_satisfy_the_air_and_hash(oracle):
    rng = np.random.default_rng(2)
    ops = sample_sponge_ops(rng)
    t = otg.keccak_sponge_generate_trace(ops, 16, _keccak_f(oracle))
    assert t.shape == (438, 256)
    _check_air(oairs.eval_keccak_sponge, t)
    r = 0
    for _, _, data in ops:                               # final row of each op holds keccak256(input)
        r += len(data) // 136
        digest = bytes(int(t[404 + i, r]) for i in range(32))
        assert digest == oracle.keccak256(data)
        r += 1


def test_reference_sponge_generation(oracle):
    """`test_generation` (keccak_sponge_stark.rs:994-1022): the op with input [1, 2, 3] at (0, Segment::Code, 0),
    timestamp 0, is ONE row whose updated_digest_state_bytes are keccak256([1, 2, 3])."""
    t = otg.keccak_sponge_generate_trace([((0, 0, 0), 0, bytes([1, 2, 3]))], 1, _keccak_f(oracle))
    digest = bytes(int(t[404 + i, 0]) for i in range(32))
    assert digest == oracle.keccak256(bytes([1, 2, 3]))
    assert digest.hex() == "f1885eda54b7a053318cd41e2093220dab15d65381b1157a3633a83bfd5c9239"


def sample_arith_ops(rng):
    from oracle import arith_trace as at
    def r256(): return int.from_bytes(rng.bytes(32), "little")
    ops = []
    for f in (at.IS_ADD, at.IS_SUB, at.IS_LT, at.IS_GT, at.IS_MUL, at.IS_DIV, at.IS_MOD):
        ops += [("bin", f, r256(), r256()), ("bin", f, r256(), r256() >> 200)]
    ops += [("bin", at.IS_DIV, r256(), 0), ("bin", at.IS_MOD, r256(), 0), ("bin", at.IS_MUL, 0, r256())]
    for f in (at.IS_ADDMOD, at.IS_SUBMOD, at.IS_MULMOD):
        ops += [("ter", f, r256(), r256(), r256()), ("ter", f, r256(), r256(), r256() >> 100), ("ter", f, r256(), r256(), 0)]
    ops += [("ter", at.IS_SUBMOD, 5, 7, 1000), ("ter", at.IS_SUBMOD, 7, 5, 1000)]
    for f in (at.IS_ADDFP254, at.IS_MULFP254, at.IS_SUBFP254):
        ops += [("bin", f, r256() % at.BN_BASE, r256() % at.BN_BASE), ("bin", f, 3, at.BN_BASE - 1)]
    for sh in (0, 1, 17, 255, 256, 1 << 40):
        ops += [("bin", at.IS_SHL, sh, r256()), ("bin", at.IS_SHR, sh, r256())]
    for idx in (0, 1, 13, 30, 31, 32, 77, 1 << 100):
        ops.append(("bin", at.IS_BYTE, idx, r256()))
    ops.append(("range_check", r256(), r256(), r256(), 0x49, r256()))
    return ops


def test_arithmetic_rows_satisfy_the_air():
    from oracle import arith_trace as at
    rng = np.random.default_rng(3)
    t, n_rows = at.generate_trace(sample_arith_ops(rng))
    assert t.shape == (116, 1 << 16) and n_rows > 80
    n = t.shape[1]

    class Cons:
        def __init__(self, i): self.i, self.bad = i, 0
        def constraint(self, c): self.bad += 1 if c % P else 0
        def constraint_transition(self, c): self.bad += 1 if (self.i != n - 1 and c % P) else 0
        def constraint_first_row(self, c): self.bad += 1 if (self.i == 0 and c % P) else 0
        def constraint_last_row(self, c): self.bad += 1 if (self.i == n - 1 and c % P) else 0
    rows = t.T
    for i in list(range(n_rows + 2)) + [65534, 65535]:     # the operation rows, the first padding rows, the wrap-around
        c = Cons(i)
        oairs.eval_arithmetic([int(v) for v in rows[i]], [int(v) for v in rows[(i + 1) % n]], c)
        assert c.bad == 0, (i, c.bad)


def sample_memory_ops(rng, n_addr=40):
    """A consistent log: per address a run of writes / reads at increasing timestamps (reads return the last value, a
    read before any write returns 0), some addresses preloaded through mem_before, one context marked stale."""
    ops, before = [], []
    ts = 1
    for _ in range(n_addr):
        addr = (int(rng.integers(0, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 60)))
        val = 0
        if rng.random() < 0.25 and not any(a == addr for a, _ in before):
            val = int.from_bytes(rng.bytes(32), "little")
            before.append((addr, val))
        elif any(a == addr for a, _ in before) or any((o["ctx"], o["seg"], o["virt"]) == addr for o in ops):
            continue
        for _ in range(int(rng.integers(1, 5))):
            ts += int(rng.integers(1, 4))
            if rng.random() < 0.5:
                val = int.from_bytes(rng.bytes(32), "little")
                ops.append(dict(filter=True, timestamp=ts, ctx=addr[0], seg=addr[1], virt=addr[2], is_read=False, value=val))
            else:
                ops.append(dict(filter=True, timestamp=ts, ctx=addr[0], seg=addr[1], virt=addr[2], is_read=True, value=val))
    return ops, before, [2]


def test_memory_rows_satisfy_the_air():
    from oracle import mem_trace as mt
    rng = np.random.default_rng(4)
    ops, before, stale = sample_memory_ops(rng)
    t, mem_after = mt.generate_trace(ops, before, stale)
    assert t.shape[0] == 30 and t.shape[1] & (t.shape[1] - 1) == 0 and len(mem_after) > 0
    _check_air(oairs.eval_memory, t)


# ---- the reference's own Arithmetic-table tests (arithmetic_stark.rs:372-519) ----------------------------------------
def reference_basic_trace_ops():
    """`basic_trace` (arithmetic_stark.rs:373-457): ten operations and the (row, first OUTPUT_REGISTER limb) pairs the
    reference asserts; the other fifteen output limbs of those rows must be zero."""
    from oracle import arith_trace as at
    ops = [("bin", at.IS_ADD, 123, 456), ("ter", at.IS_MULMOD, 123, 456, 1007), ("ter", at.IS_ADDMOD, 1234, 567, 1007),
           ("bin", at.IS_MUL, 123, 456), ("bin", at.IS_MOD, 128, 13), ("bin", at.IS_LT, 128, 13),
           ("bin", at.IS_LT, 13, 128), ("bin", at.IS_LT, 128, 128), ("bin", at.IS_DIV, 128, 13),
           ("bin", at.IS_BYTE, 30, 0xABCD)]
    expected = [(0, 579), (1, 703), (3, 794), (5, 56088), (6, 11), (8, 0), (9, 1), (10, 0), (11, 9), (13, 0xAB)]
    return ops, expected


def check_reference_basic_trace(t, n_rows):
    from oracle import arith_trace as at
    _, expected = reference_basic_trace_ops()
    assert t.shape == (at.NUM_COLS, 1 << 16) and n_rows == 14      # NUM_ARITH_COLUMNS x RANGE_MAX; 4 two-row operations
    for row, value in expected:
        assert int(t[at.OUT, row]) == value, (row, value, int(t[at.OUT, row]))
        assert not t[at.OUT + 1:at.OUT + 16, row].any(), row


def test_reference_basic_trace():
    from oracle import arith_trace as at
    ops, _ = reference_basic_trace_ops()
    check_reference_basic_trace(*at.generate_trace(ops))


def test_reference_big_traces():
    """`big_traces` (arithmetic_stark.rs:460-519): RANGE_MAX one-row MULs keep the table at RANGE_MAX rows, RANGE_MAX
    two-row MULMODs double it.  (A reduced count keeps the Python generator to seconds: the height rule is what the
    reference asserts, and 2^15 + 1 two-row operations are the smallest log that must spill over 2^16 rows.)"""
    from oracle import arith_trace as at
    rng = np.random.default_rng(0x6FEB51B7EC230F25 % (1 << 32))
    r256 = lambda: int.from_bytes(rng.bytes(32), "little")
    t, n = at.generate_trace([("bin", at.IS_MUL, r256(), r256()) for _ in range(3000)])
    assert t.shape == (at.NUM_COLS, 1 << 16) and n == 3000
    t, n = at.generate_trace([("ter", at.IS_MULMOD, r256(), r256(), r256()) for _ in range((1 << 15) + 1)])
    assert t.shape == (at.NUM_COLS, 1 << 17) and n == (1 << 16) + 2
