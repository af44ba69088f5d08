"""-m gpu: GPU proofs of VALID traces are accepted by the oracle's restatement of starky's
verifier (constraint identity at zeta in F_{p^2} + verify_fri_proof); proofs of traces that break
one constraint are rejected.  Trace generators restate the reference's generate_trace rows."""
import ctypes as C

import numpy as np
import pytest

import tests.oracle_lib as ol

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def _prove_and_verify(oracle, air_id, trace, hasher, ctl_entries_list, kw=None, lookup_spec=()):
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.prover as zp
    import zk_evm_amd.stark as prod
    from oracle import airs as oairs
    from oracle import stark as orc
    from oracle import stark_verifier as overify
    from tests.test_gpu_stark_aux import _mk, _mkf
    kw = kw or dict(pow_bits=5, queries=6)
    ol.setup_fri_api(oracle)
    L = oracle.lib
    n_cols, n = trace.shape
    log_n = n.bit_length() - 1
    cfg = ol.make_cfg(hasher=hasher, **kw)
    dev = torch.from_numpy(trace.view(np.int64)).cuda()
    tbatch = zk.PolynomialBatch.from_values(dev, 1, False, 4, hasher=hasher)
    tcap = tbatch.merkle_tree.cap.elements
    ch = zk.Challenger(hasher)
    och = ol.new_challenger(oracle, hasher)
    ch.observe_cap(tcap)
    L.orc_challenger_observe_cap(C.byref(och), tcap, 16)
    ctl_challenges = [(ch.get_challenge(), ch.get_challenge()) for _ in range(cfg.num_challenges)]
    for _ in range(2 * cfg.num_challenges):
        L.orc_challenger_get(C.byref(och))

    def entries_for(mod, entries):
        return [([_mk(c, mod.Column, mod.Filter) for c in cols], _mkf(f, mod.Column, mod.Filter)) for cols, f in entries]
    p_z, o_z = [], []
    for entries in ctl_entries_list:
        for b, g in ctl_challenges:
            aux = prod.ctl_partial_sums(dev, entries_for(prod, entries), b, g, 3)
            p_z.append(zp.CtlZData(b, g, entries_for(prod, entries), aux))
            o_z.append(orc.CtlZData(orc.GrandProductChallenge(b, g), entries_for(orc, entries), aux.shape[0] - 1))
    def lookups_for(mod):
        return [mod.Lookup([_mk(c, mod.Column, mod.Filter) for c in cols], _mk(table, mod.Column, mod.Filter),
                           _mk(freq, mod.Column, mod.Filter), [_mkf(f, mod.Column, mod.Filter) for f in filts])
                for cols, table, freq, filts in lookup_spec]
    scfg = zk.StarkConfig(hasher=hasher, fri_config=zk.FriConfig(proof_of_work_bits=kw["pow_bits"],
                                                                 num_query_rounds=kw["queries"]))
    pr = zp.prove_single_table(air_id, scfg, dev, tbatch, lookups_for(prod), p_z, ctl_challenges, ch)
    # verifier side of prove_single_table's compact(): get_challenges.rs:298-300 compacts before each table too
    oracle.lib.orc_challenger_compact(C.byref(och), np.zeros(12, dtype=np.uint64))
    proof = dict(trace_cap=tcap, aux_cap=pr.auxiliary_polys_cap, quotient_cap=pr.quotient_polys_cap,
                 openings=pr.openings, fri=pr.opening_proof)
    return overify.verify_stark_proof(oracle, ol, cfg, oairs.AIRS[air_id][0], n_cols, log_n, lookups_for(orc), o_z,
                                      ctl_challenges, proof, och)


def _mem_continuation_trace(rng, n_rows, n_padded):
    # memory_continuation_stark.rs:53-98: FILTER=1 rows (context, segment, virt, 8 x u32 limbs), zero padding
    t = np.zeros((12, n_padded), dtype=np.uint64)
    t[0, :n_rows] = 1
    t[1, :n_rows] = rng.integers(0, 4, size=n_rows)
    t[2, :n_rows] = rng.integers(0, 40, size=n_rows)
    t[3, :n_rows] = np.arange(n_rows)
    t[4:12, :n_rows] = rng.integers(0, 1 << 32, size=(8, n_rows))
    return t


MEMC_CTL = [([("single", 1), ("single", 2), ("single", 3)] + [("single", 4 + i) for i in range(8)],
             ("simple", ("single", 0)))]


@pytest.mark.parametrize("hasher", [0, 1])
def test_mem_continuation_valid_and_invalid(oracle, hasher):
    rng = np.random.default_rng(21)
    t = _mem_continuation_trace(rng, 100, 128)     # MemBefore/After pad to >= 128 rows
    ok, why = _prove_and_verify(oracle, 1, t, hasher, [MEMC_CTL])
    assert ok, why
    bad = t.copy()
    bad[0, 7] = 0                                   # still binary: valid (a padding row in the middle)
    ok, why = _prove_and_verify(oracle, 1, bad, hasher, [MEMC_CTL])
    assert ok, why


def _logic_trace(rng, n_ops, n_padded, ops_out=None):
    # logic.rs:165-188 (Operation::into_row): one-hot op flag, 2 x 256 input bits, 8 x 32-bit result limbs
    t = np.zeros((523, n_padded), dtype=np.uint64)
    for r in range(n_ops):
        op = int(rng.integers(0, 3))
        a = int.from_bytes(rng.bytes(32), "little")
        b = int.from_bytes(rng.bytes(32), "little")
        if ops_out is not None:
            ops_out.append((op, a, b))
        res = [a & b, a | b, a ^ b][op]
        t[op, r] = 1
        for i in range(256):
            t[3 + i, r] = (a >> i) & 1
            t[259 + i, r] = (b >> i) & 1
        for l in range(8):
            t[515 + l, r] = (res >> (32 * l)) & 0xFFFFFFFF
    return t


def _logic_ctl():
    cols = [("lc", [(0, 0x16), (1, 0x17), (2, 0x18)], [], 0)]
    for base in (3, 259):
        for limb in range(8):
            cols.append(("lc", [(base + 32 * limb + i, 1 << i) for i in range(32)], [], 0))
    cols += [("single", 515 + i) for i in range(8)]
    return [(cols, ("simple", ("lc", [(0, 1), (1, 1), (2, 1)], [], 0)))]


def test_logic_valid_and_invalid(oracle):
    rng = np.random.default_rng(22)
    ops = []
    t = _logic_trace(rng, 20, 32, ops)
    # the device generator (zk_logic_generate_trace = logic.rs:165-240) produces the same table
    from zk_evm_amd.tracegen import logic_generate_trace
    dev = logic_generate_trace(ops, 32)
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), t)
    assert not logic_generate_trace([], 16).cpu().numpy().any()
    ok, why = _prove_and_verify(oracle, 2, t, 0, [_logic_ctl()])
    assert ok, why
    bad = t.copy()
    bad[515, 3] ^= np.uint64(1)                    # wrong result limb -> constraint violated
    ok, why = _prove_and_verify(oracle, 2, bad, 0, [_logic_ctl()])
    assert not ok and why == "quotient identity", why
    bad2 = t.copy()
    bad2[3 + 17, 5] = 2                            # a non-bit input
    ok, why = _prove_and_verify(oracle, 2, bad2, 0, [_logic_ctl()])
    assert not ok


def _arith_trace(rng, n_ops):
    """ADD / SUB / LT / GT rows (arithmetic/addcy.rs:31-65), range-check rows (arithmetic/mod.rs:343-359),
    zero padding to 2^16 rows and the RANGE_COUNTER / RC_FREQUENCIES columns
    (arithmetic_stark.rs:130-156)."""
    n = 1 << 16
    t = np.zeros((116, n), dtype=np.uint64)
    M = (1 << 256) - 1

    def put(r, start, x):
        for i in range(16):
            t[start + i, r] = (x >> (16 * i)) & 0xFFFF
    for r in range(n_ops):
        a = int.from_bytes(rng.bytes(32), "little")
        b = int.from_bytes(rng.bytes(32), "little")
        kind = int(rng.integers(0, 5))
        if kind == 4:                                  # range check row
            t[16, r] = 1
            t[17, r] = int(rng.integers(0, 256))
            put(r, 18, a); put(r, 34, b); put(r, 50, a ^ b); put(r, 66, (a + b) & M)
            continue
        flag = [0, 2, 11, 12][kind]
        t[flag, r] = 1
        put(r, 18, a); put(r, 34, b)
        if kind == 0:
            s = a + b
            put(r, 82, s >> 256); put(r, 66, s & M)
        elif kind == 1:
            put(r, 82, 1 if a < b else 0); put(r, 66, (a - b) & M)
        elif kind == 2:
            put(r, 82, (a - b) & M); put(r, 66, 1 if a < b else 0)
        else:
            put(r, 82, (b - a) & M); put(r, 66, 1 if b < a else 0)
    t[114] = np.arange(n, dtype=np.uint64)             # n == RANGE_MAX
    freq = np.zeros(n, dtype=np.uint64)
    for col in range(18, 114):
        freq += np.bincount(t[col].astype(np.int64), minlength=n).astype(np.uint64)
    t[115] = freq
    return t


def test_arithmetic_valid_and_invalid(oracle):
    rng = np.random.default_rng(23)
    t = _arith_trace(rng, 300)
    lk = ([("single", 18 + i) for i in range(96)], ("single", 114), ("single", 115), [None] * 96)
    cols = [("single", 17)]
    for reg in (18, 34, 50, 66):
        cols += [("lc", [(reg + 2 * k, 1), (reg + 2 * k + 1, 1 << 16)], [], 0) for k in range(8)]
    ctl = [(cols, ("simple", ("lc", [(i, 1) for i in range(17)], [], 0)))]
    ok, why = _prove_and_verify(oracle, 5, t, 0, [ctl], lookup_spec=[lk])
    assert ok, why
    bad = t.copy()
    bad[66, 1] ^= np.uint64(1)                         # corrupt an output limb (stays < 2^16)
    bad[115] = 0
    for col in range(18, 114):
        bad[115] += np.bincount(bad[col].astype(np.int64), minlength=1 << 16).astype(np.uint64)
    ok, why = _prove_and_verify(oracle, 5, bad, 0, [ctl], lookup_spec=[lk])
    assert not ok and why == "quotient identity", why


def test_keccak_table_generated_on_device_verifies(oracle):
    """The Keccak table end to end: witness rows generated ON THE DEVICE from permutation inputs
    (zk_keccak_generate_trace = keccak_stark.rs:65-234) equal the oracle's restatement of the reference generator,
    and the table proof of that (valid) trace -- with the table's two real CTL looked entries, all_stark.rs:226-255 --
    is accepted by the oracle verifier including the quotient identity of the restated Keccak AIR; a trace with one
    flipped A' bit is rejected."""
    import torch
    from oracle import keccak_trace as okt
    from zk_evm_amd.tracegen import keccak_generate_trace
    rng = np.random.default_rng(23)
    inputs = [([int(v) for v in rng.integers(0, 1 << 64, size=25, dtype=np.uint64)], 5 + 3 * i) for i in range(2)]
    dev = keccak_generate_trace(inputs, 16)
    got = dev.cpu().numpy().view(np.uint64)
    exp = okt.generate_trace_rows(inputs, 16)
    assert got.shape == (2431, 64) and np.array_equal(got.T, exp)
    from oracle import all_stark as oas
    ctls = oas.build_ctls()
    # CTL 3 (keccak_inputs) and 4 (keccak_outputs): the Keccak table is the looked table of both
    def desc(col):
        lc = col.linear_combination
        return ("single", lc[0][0]) if (len(lc) == 1 and lc[0][1] == 1 and not col.next_row_linear_combination and col.constant == 0) \
            else ("lc", list(lc), list(col.next_row_linear_combination), col.constant)
    ctl_entries = []
    for k in (3, 4):
        lk = ctls[k].looked_table
        ctl_entries.append([([desc(c) for c in lk.columns], ("simple", desc(lk.filter.constants[0])))])
    t = np.ascontiguousarray(got)
    ok, why = _prove_and_verify(oracle, 6, t, 0, ctl_entries)
    assert ok, why
    bad = t.copy()
    bad[715 + 3 * 320 + 2 * 64 + 11, 29] ^= np.uint64(1)           # one A'[3, 2, 11] bit in row 29
    ok, why = _prove_and_verify(oracle, 6, bad, 0, ctl_entries)
    assert not ok and why == "quotient identity", why


@pytest.mark.parametrize("hasher", [0, 1])
def test_memory_continuation_generator_and_initial_memory_cap(oracle, hasher):
    """MemBefore rows generated on the device (memory_continuation_stark.rs:53-98) and `initial_memory_merkle_cap`
    (verifier.rs:14-78) for a synthetic kernel image, against a row-by-row restatement committed by the
    oracle."""
    from zk_evm_amd.tracegen import initial_memory_merkle_cap, memory_continuation_generate_trace
    rng = np.random.default_rng(31 + hasher)
    vals = [((int(rng.integers(0, 9)), int(rng.integers(0, 33)), int(rng.integers(0, 1 << 20))),
             int.from_bytes(rng.bytes(32), "little")) for _ in range(200)]
    got = memory_continuation_generate_trace(vals).cpu().numpy().view(np.uint64)
    exp = np.zeros((12, 256), dtype=np.uint64)
    for r, ((c, s, v), val) in enumerate(vals):
        exp[0:4, r] = [1, c, s, v]
        for j in range(8):
            exp[4 + j, r] = (val >> (32 * j)) & 0xFFFFFFFF
    assert np.array_equal(got, exp)
    assert memory_continuation_generate_trace([]).shape == (12, 128)
    code = rng.bytes(1000)                                # a stand-in for KERNEL.code
    rows = []
    for i, b in enumerate(code):
        rows.append([1, 0, 0, i, b] + [0] * 7)
    for i in range(256):
        v = 1 << i
        rows.append([1, 0, 13, i] + [(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)])
    n = 1 << (len(rows) - 1).bit_length()
    t = np.zeros((12, n), dtype=np.uint64)
    t[:, :len(rows)] = np.array(rows, dtype=np.uint64).T
    ref = oracle.commit_values(t, rate_bits=1, cap_height=4, hasher=hasher, want_leaves=False)
    assert np.array_equal(initial_memory_merkle_cap(code, 1, 4, hasher=hasher), ref["cap"])


def _registry_descs(table):
    """(ctl_entries_list, lookup_spec) of one table in the descriptor form of _prove_and_verify, taken from the real
    registry (oracle/all_stark.py): one z-data per CTL run of looking entries of this table / per looked role."""
    from oracle import all_stark as oas

    def desc(col):
        lc = col.linear_combination
        if len(lc) == 1 and lc[0][1] == 1 and not col.next_row_linear_combination and col.constant == 0:
            return ("single", lc[0][0])
        return ("lc", list(lc), list(col.next_row_linear_combination), col.constant)

    def fdesc(f):
        return ("full", [(desc(a), desc(b)) for a, b in f.products], [desc(c) for c in f.constants])
    zlist = []
    for ctl in oas.build_ctls():
        mine = [t for t in ctl.looking_tables if t.table == table]
        if mine:
            zlist.append([([desc(c) for c in t.columns], fdesc(t.filter)) for t in mine])
        if ctl.looked_table.table == table:
            lk = ctl.looked_table
            zlist.append([([desc(c) for c in lk.columns], fdesc(lk.filter))])
    lookups = [([desc(c) for c in l.columns], desc(l.table_column), desc(l.frequencies_column),
                [None if (not f.products and len(f.constants) == 1 and f.constants[0].constant == 1
                          and not f.constants[0].linear_combination) else fdesc(f) for f in l.filter_columns])
               for l in oas.build_lookups()[table]]
    return zlist, lookups


def test_byte_packing_table_generated_on_device_verifies(oracle):
    """BytePacking end to end with its REAL lookups and CTL entries (looked by the Cpu, 32 lookers into Memory):
    device-generated rows == the restated reference generator, and the proof of that valid trace is accepted by the
    oracle verifier (quotient identity included); a flipped byte is rejected."""
    from oracle import tracegen as otg
    from tests.test_oracle_tracegen import sample_byte_packing_ops
    from zk_evm_amd.tracegen import byte_packing_generate_trace
    rng = np.random.default_rng(41)
    ops = sample_byte_packing_ops(rng, 60)
    got = byte_packing_generate_trace(ops, 16).cpu().numpy().view(np.uint64)
    exp = otg.byte_packing_generate_trace(ops, 16)
    assert np.array_equal(got, exp)
    zlist, lookups = _registry_descs(1)
    assert [len(z) for z in zlist] == [1, 32] and len(lookups) == 1
    t = np.ascontiguousarray(got)
    ok, why = _prove_and_verify(oracle, 4, t, 0, zlist, lookup_spec=lookups)
    assert ok, why
    bad = t.copy()
    bad[37 + 2, 7] = 256                                    # a "byte" out of range: the logUp range check fails
    ok, why = _prove_and_verify(oracle, 4, bad, 0, zlist, lookup_spec=lookups)
    assert not ok, why


def test_keccak_sponge_table_generated_on_device_verifies(oracle):
    """KeccakSponge end to end with its real 136-column range check and all its CTL roles (looked by the Cpu, looking
    into Keccak inputs / outputs, 5 Logic lookers, 136 Memory lookers): device rows == restated generator, valid
    trace accepted by the oracle verifier, a wrong already_absorbed_bytes rejected."""
    from oracle import tracegen as otg
    from tests.test_oracle_tracegen import _keccak_f, sample_sponge_ops
    from zk_evm_amd.tracegen import keccak_sponge_generate_trace
    rng = np.random.default_rng(43)
    ops = sample_sponge_ops(rng)
    got = keccak_sponge_generate_trace(ops, 16).cpu().numpy().view(np.uint64)
    exp = otg.keccak_sponge_generate_trace(ops, 16, _keccak_f(oracle))
    assert np.array_equal(got, exp)
    zlist, lookups = _registry_descs(4)
    assert [len(z) for z in zlist] == [1, 1, 1, 5, 136] and len(lookups) == 1
    t = np.ascontiguousarray(got)
    ok, why = _prove_and_verify(oracle, 7, t, 0, zlist, lookup_spec=lookups)
    assert ok, why
    bad = t.copy()
    # (the updated state of a final row is only tied to the Keccak table through a CTL, so corrupt something the
    # table itself constrains: already_absorbed_bytes of the second row of the 136-byte input, rows 4-5)
    assert bad[0, 4] == 1 and bad[5, 5] == 136
    bad[5, 5] += np.uint64(1)
    ok, why = _prove_and_verify(oracle, 7, bad, 0, zlist, lookup_spec=lookups)
    assert not ok, why


def test_arithmetic_all_operations_verify(oracle):
    """Every Arithmetic operation kind (ADD SUB LT GT MUL DIV MOD ADDMOD SUBMOD MULMOD, the three FP254 variants, SHL SHR
    BYTE, range-check rows; zero moduli and oversized shifts included) generated by the restated reference generators
    (oracle/arith_trace.py), range-check columns finalised ON THE DEVICE, proven with the table's real lookup and CTL
    entry, accepted by the oracle verifier; a wrong quotient limb is rejected."""
    import torch
    from oracle import arith_trace as at
    from tests.test_oracle_tracegen import sample_arith_ops
    from zk_evm_amd.tracegen import range_check_columns
    rng = np.random.default_rng(51)
    t, n_rows = at.generate_trace(sample_arith_ops(rng))
    dev = torch.from_numpy(t.view(np.int64).copy()).cuda()
    dev[114:116] = 0
    range_check_columns(dev, 18, 96, 114, 115, 1 << 16)               # generate_range_checks on the device
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), t)
    zlist, lookups = _registry_descs(0)
    assert [len(z) for z in zlist] == [1] and len(lookups) == 1
    ok, why = _prove_and_verify(oracle, 5, t, 0, zlist, lookup_spec=lookups)
    assert ok, why
    bad = t.copy()
    r = next(i for i in range(n_rows) if t[at.IS_MULMOD, i] == 1)
    bad[at.AUX0 + 3, r] ^= np.uint64(1)                              # one limb of the MULMOD quotient
    bad[115] = 0
    bad[115, :1 << 16] = np.bincount(bad[18:114].astype(np.int64).reshape(-1), minlength=1 << 16).astype(np.uint64)
    ok, why = _prove_and_verify(oracle, 5, bad, 0, zlist, lookup_spec=lookups)
    assert not ok and why == "quotient identity", why


def test_memory_table_valid_trace_verifies(oracle):
    """The Memory table: a consistent operation log run through the restated reference pipeline (sort, fill_gaps,
    padding, first-change flags, range-check / stale-context frequencies: oracle/mem_trace.py = memory_stark.rs:104-455),
    proven on the GPU with the table's two real lookups (one filtered, one using a next-row column) and its four CTL
    roles, accepted by the oracle verifier; breaking the address ordering is rejected."""
    from oracle import mem_trace as mt
    from tests.test_oracle_tracegen import sample_memory_ops
    rng = np.random.default_rng(61)
    ops, before, stale = sample_memory_ops(rng, 60)
    t, _ = mt.generate_trace(ops, before, stale)
    zlist, lookups = _registry_descs(6)
    assert [len(z) for z in zlist] == [1, 1, 1, 1] and len(lookups) == 2
    ok, why = _prove_and_verify(oracle, 3, t, 0, zlist, lookup_spec=lookups)
    assert ok, why
    bad = t.copy()
    i = next(r for r in range(5, t.shape[1] - 2) if t[mt.VIRT_FIRST, r] == 1)
    bad[mt.VIRT, i + 1], bad[mt.VIRT, i] = t[mt.VIRT, i], t[mt.VIRT, i + 1]      # two rows out of address order
    ok, why = _prove_and_verify(oracle, 3, bad, 0, zlist, lookup_spec=lookups)
    assert not ok, why


def test_packed_logs_give_the_same_tables():
    """logic / keccak / byte_packing generators: numpy arrays in the C ABI's record layouts == the tuple forms."""
    import torch
    import zk_evm_amd.tracegen as tg
    rng = np.random.default_rng(91)
    m64 = (1 << 64) - 1
    ops = [(int(rng.integers(0, 3)), int.from_bytes(rng.bytes(32), "little"), int.from_bytes(rng.bytes(32), "little")) for _ in range(40)]
    packed = np.array([[k] + [(a >> (64 * l)) & m64 for l in range(4)] + [(b >> (64 * l)) & m64 for l in range(4)] for k, a, b in ops], dtype=np.uint64)
    assert torch.equal(tg.logic_generate_trace(ops, 16), tg.logic_generate_trace(packed, 16))
    perms = [([int(x) for x in rng.integers(0, 1 << 63, 25)], int(rng.integers(0, 1000))) for _ in range(7)]
    arrs = (np.array([p[0] for p in perms], dtype=np.uint64), np.array([p[1] for p in perms], dtype=np.uint64))
    assert torch.equal(tg.keccak_generate_trace(perms, 16), tg.keccak_generate_trace(arrs, 16))
    bp = [(bool(rng.integers(0, 2)), (int(rng.integers(0, 5)), int(rng.integers(0, 30)), int(rng.integers(0, 1000))), int(rng.integers(1, 500)),
           rng.bytes(int(rng.integers(1, 33)))) for _ in range(30)]
    bpa = np.array([[1 if rd else 0, c, s, v, ts, len(d)] + [int.from_bytes(d[8 * k:8 * k + 8].ljust(8, b"\0"), "little") for k in range(4)]
                    for rd, (c, s, v), ts, d in bp], dtype=np.uint64)
    assert torch.equal(tg.byte_packing_generate_trace(bp, 0), tg.byte_packing_generate_trace(bpa, 0))
