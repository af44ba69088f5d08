"""-m gpu: the device Memory-table generator (zk_memory_trace_begin / _finish: radix sort, closed-form fill_gaps,
padding, flags, range-check / frequency / stale-context columns, final-memory extraction) against the oracle's
literal restatement of MemoryStark::generate_trace (oracle/mem_trace.py: the reference's while-loops and re-sorts),
cell for cell; then the generated table is proven and accepted by the oracle verifier."""
import numpy as np
import pytest

from oracle import mem_trace as mt

pytestmark = pytest.mark.gpu


def _to_product(ops, before):
    return ([(o["filter"], o["timestamp"], (o["ctx"], o["seg"], o["virt"]), o["is_read"], o["value"]) for o in ops],
            [(a, v) for a, v in before])


def _check(ops, before, stale):
    from zk_evm_amd.tracegen import memory_generate_trace
    exp, exp_after = mt.generate_trace(ops, before, stale)
    pops, pbefore = _to_product(ops, before)
    trace, after, final, unpadded = memory_generate_trace(pops, pbefore, stale)
    got = trace.cpu().numpy().view(np.uint64)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    for c in range(30):
        assert np.array_equal(got[c], exp[c]), ("column", c, np.nonzero(got[c] != exp[c])[0][:5])
    assert len(final) == len(exp_after)
    for ((c, s, v), val), row in zip(final, exp_after):
        assert [1, c, s, v] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] == row
    a = after.cpu().numpy().view(np.uint64)
    assert a.shape == (12, max(128, 1 << max(len(exp_after) - 1, 0).bit_length()))
    if exp_after:
        assert np.array_equal(a[:, :len(exp_after)], np.array(exp_after, dtype=np.uint64).T)
    assert not a[:, len(exp_after):].any()
    # unpadded_length: rows before pad_memory_ops = everything that is not the (identical) padding operation
    pad_rows = int(exp.shape[1] - unpadded)
    assert 1 <= pad_rows and np.all(exp[mt.FILTER, unpadded:] == 0)
    return trace


def test_random_logs_match_reference_generator():
    from tests.test_oracle_tracegen import sample_memory_ops
    for seed in (3, 4, 5):
        rng = np.random.default_rng(seed)
        ops, before, stale = sample_memory_ops(rng, 30 + 40 * seed)
        _check(ops, before, stale)


def _w(ts, ctx, seg, virt, val, read=False):
    return dict(filter=True, timestamp=ts, ctx=ctx, seg=seg, virt=virt, is_read=read, value=val)


def test_fill_gaps_all_three_cases_and_front_dummy():
    """Few operations, wide gaps: a first address with virt != 0 (front dummy), a virt gap inside a segment
    (ascending dummies, timestamps +1 each), a first virt of a new segment / context beyond max_rc (descending
    pushes), and a timestamp gap on one address (dummies repeat the value)."""
    v = 0x1234567890ABCDEF1122334455667788
    ops = [_w(3, 0, 1, 7, v), _w(5, 0, 1, 400, v + 1), _w(9, 0, 1, 400, v + 1, read=True),
           _w(900, 0, 1, 400, v + 1, read=True),                    # timestamp gap
           _w(12, 0, 4, 1000, v + 2),                               # new segment, first virt beyond max_rc
           _w(13, 2, 3, 77, 0), _w(14, 2, 3, 78, v + 3), _w(15, 5, 0, 33, v + 4)]
    before = [((0, 1, 2), v + 9), ((7, 12, 130), v + 10)]
    t = _check(ops, before, [2])
    assert t.shape[1] >= 64
    _check(ops[:1], [], [])                                          # one operation: a two-element list
    _check([_w(1, 0, 0, 0, 5)], [], [])                              # first address (0,0,0): no front dummy
    _check([_w(4, 0, 0, 0, 5), _w(1, 0, 0, 0, 6), _w(2, 1, 1, 0, 7)], [((3, 35, 0), 0)], [3])   # unsorted input


def random_log(rng, n_ops=6000, n_before=500, virt_range=3000):
    ops, ts = [], 1
    state = {}
    for _ in range(n_ops):
        addr = (int(rng.integers(0, 6)), int(rng.integers(0, 36)), int(rng.integers(0, virt_range)))
        ts += int(rng.integers(1, 3))
        if rng.random() < 0.5 or addr not in state:
            state[addr] = int.from_bytes(rng.bytes(32), "little") if rng.random() < 0.9 else 0
            ops.append(_w(ts, *addr, state[addr]))
        else:
            ops.append(_w(ts, *addr, state[addr], read=True))
    before = []
    for _ in range(n_before):
        addr = (int(rng.integers(0, 6)), int(rng.integers(0, 36)), int(rng.integers(virt_range, 3 * virt_range)))
        if addr not in state:
            state[addr] = 1
            before.append((addr, int.from_bytes(rng.bytes(32), "little")))
    return ops, before


def test_large_log_matches():
    ops, before = random_log(np.random.default_rng(9))
    _check(ops, before, [1, 4])


def test_generated_memory_table_is_proven_and_accepted(oracle):
    from tests.test_gpu_stark_verify import _prove_and_verify, _registry_descs
    from tests.test_oracle_tracegen import sample_memory_ops
    from zk_evm_amd.tracegen import memory_generate_trace
    rng = np.random.default_rng(12)
    ops, before, stale = sample_memory_ops(rng, 80)
    trace, _, _, _ = memory_generate_trace(*_to_product(ops, before), stale)
    zlist, lookups = _registry_descs(6)
    ok, why = _prove_and_verify(oracle, 3, trace.cpu().numpy().view(np.uint64), 0, zlist, lookup_spec=lookups)
    assert ok, why


def test_error_paths():
    from zk_evm_amd import ZkStarkError
    from zk_evm_amd.tracegen import memory_generate_trace
    with pytest.raises(ZkStarkError):
        memory_generate_trace([], [], [])                            # the reference indexes memory_ops[0]
    op = (True, 1, (0, 0, 0), False, 5)
    with pytest.raises(ZkStarkError):
        memory_generate_trace([op], [], [1, 1])                      # "Stale contexts are not unique."
    with pytest.raises(ZkStarkError):
        memory_generate_trace([(True, 1 << 32, (0, 0, 0), False, 5)], [], [])
    with pytest.raises(ZkStarkError):
        memory_generate_trace([(True, 1, (900, 0, 0), False, 5)], [], [])   # context beyond the trace height
    with pytest.raises(ZkStarkError):                                # 2^31 virt gap over a 2-operation log
        memory_generate_trace([op, (True, 2, (0, 0, 1 << 31), False, 5), (True, 3, (0, 0, (1 << 32) - 1), False, 5)], [], [])


def test_packed_arrays_are_passed_through():
    """(n, 9) / (n, 7) uint64 arrays in the C ABI's record layout give the same table as the tuple form."""
    import torch
    from tests.test_oracle_tracegen import sample_memory_ops
    from zk_evm_amd.tracegen import memory_generate_trace
    ops, before, stale = sample_memory_ops(np.random.default_rng(21), 50)
    pops, pbefore = _to_product(ops, before)
    m64 = (1 << 64) - 1
    a = np.array([[(1 if r else 0) | (2 if f else 0), ts, c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)]
                  for f, ts, (c, s, v), r, val in pops], dtype=np.uint64)
    b = np.array([[c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)] for (c, s, v), val in pbefore], dtype=np.uint64)
    t1, a1, f1, u1 = memory_generate_trace(pops, pbefore, stale)
    t2, a2, f2, u2 = memory_generate_trace(a, b, stale)
    assert torch.equal(t1, t2) and torch.equal(a1, a2) and f1 == f2 and u1 == u2


def test_equal_keys_keep_input_order():
    """`sort_by_key` is stable: operations with the same (context, segment, virt, timestamp) stay in log order
    (memory_ops first, then the mem_before writes).  The device radix sort must do the same, cell for cell."""
    v = 0xAABBCCDD00112233445566778899
    ops = [_w(5, 1, 2, 3, v, read=True), _w(5, 1, 2, 3, v + 1), _w(5, 1, 2, 3, v + 2, read=True),
           _w(0, 1, 2, 9, 7), _w(5, 0, 0, 0, 1), _w(5, 0, 0, 0, 2), _w(5, 0, 0, 0, 3)]
    before = [((1, 2, 9), 8), ((1, 2, 9), 9)]                   # same key as the timestamp-0 operation above
    _check(ops, before, [])
