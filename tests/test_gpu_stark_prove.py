"""-m gpu: per-table STARK proof (starky prove_with_commitment) on the GPU vs the oracle
restatement, word for word: auxiliary cap, quotient cap, openings, FRI proof.  Random traces
(the prover does not require the trace to satisfy the AIR; acceptance by a verifier does -- see
test_gpu_stark_verify.py for valid traces)."""
import ctypes as C

import numpy as np
import pytest

import tests.oracle_lib as ol

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def _descs_to(mod, col_descs):
    from tests.test_gpu_stark_aux import _mk
    return [_mk(d, mod.Column, mod.Filter) for d in col_descs]


# tests/fuzz_parity.py installs a callable here that re-draws (log_n, hasher, seed, kw) for every case, so the same
# table definitions are replayed at other heights / FRI shapes / hashers than the pinned ones below.
FUZZ = None


def _run_case(oracle, air_id, n_cols, log_n, hasher, lookup_spec, ctl_spec, seed, binary_cols=(), kw=None,
              trace_fix=None, air_consts=()):
    import torch
    if FUZZ is not None:
        log_n, hasher, seed, kw = FUZZ(n_cols, log_n, hasher, seed, kw)
    import zk_evm_amd as zk
    import zk_evm_amd.prover as zp
    import zk_evm_amd.stark as prod
    from oracle import airs as oairs
    from oracle import stark as orc
    from oracle import stark_prover as oprover
    from tests.test_gpu_stark_aux import _mk, _mkf
    kw = kw or dict(pow_bits=3, queries=3)
    ol.setup_fri_api(oracle)
    L = oracle.lib
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    trace = rng.integers(0, 1 << 64, size=(n_cols, n), dtype=np.uint64)
    for c in binary_cols:
        trace[c] = rng.integers(0, 2, size=n, dtype=np.uint64)
    if trace_fix:
        trace_fix(trace, rng)
    cfg = ol.make_cfg(hasher=hasher, **kw)
    nchal = cfg.num_challenges
    # --- commit + transcript prefix (both sides) ---
    tcommit = oracle.commit_values(trace, rate_bits=1, cap_height=4, hasher=hasher)
    dev = torch.from_numpy(trace.view(np.int64)).cuda()
    tbatch = zk.PolynomialBatch.from_values(dev, 1, False, 4, hasher=hasher)
    och = ol.new_challenger(oracle, hasher)
    ch = zk.Challenger(hasher)
    L.orc_challenger_observe_cap(C.byref(och), tcommit["cap"], 16)
    ch.observe_cap(tcommit["cap"])
    ctl_challenges = [(ch.get_challenge(), ch.get_challenge()) for _ in range(nchal)]
    for b, g in ctl_challenges:
        assert (b, g) == (L.orc_challenger_get(C.byref(och)), L.orc_challenger_get(C.byref(och)))

    def lookups_for(mod):
        out = []
        for cols, table, freq, filts in lookup_spec:
            out.append(mod.Lookup([_mk(c, mod.Column, mod.Filter) for c in cols], _mk(table, mod.Column, mod.Filter),
                                  _mk(freq, mod.Column, mod.Filter), [_mkf(f, mod.Column, mod.Filter) for f in filts]))
        return out

    def entries_for(mod, entries):
        return [([_mk(c, mod.Column, mod.Filter) for c in cols], _mkf(f, mod.Column, mod.Filter)) for cols, f in entries]

    # CTL z-data: for each CTL group in ctl_spec, one z-data per challenge (cross_table_lookup_data order)
    o_z, p_z = [], []
    for entries in ctl_spec:
        for b, g in ctl_challenges:
            o_z.append(orc.CtlZData(orc.GrandProductChallenge(b, g), entries_for(orc, entries), 0))
            aux = prod.ctl_partial_sums(dev, entries_for(prod, entries), b, g, 3)
            p_z.append(zp.CtlZData(b, g, entries_for(prod, entries), aux))
    # the product call is the reference's prove_single_table: it compacts the challenger first (prover.rs:318-320)
    init = np.zeros(12, dtype=np.uint64)
    L.orc_challenger_compact(C.byref(och), init)
    exp = oprover.prove_with_commitment(oracle, ol, cfg, oairs.AIRS[air_id][0], trace, tcommit, lookups_for(orc),
                                        o_z, ctl_challenges, och)
    scfg = zk.StarkConfig(hasher=hasher, num_challenges=nchal,
                          fri_config=zk.FriConfig(proof_of_work_bits=kw["pow_bits"], num_query_rounds=kw["queries"]))
    got = zp.prove_single_table(air_id, scfg, dev, tbatch, lookups_for(prod), p_z, ctl_challenges, ch,
                                   air_consts=air_consts)
    if exp["aux_cap"] is None:
        assert got.auxiliary_polys_cap is None
    else:
        assert np.array_equal(got.auxiliary_polys_cap, exp["aux_cap"])
    assert np.array_equal(got.init_challenger_state, init)
    assert np.array_equal(got.trace_cap, tcommit["cap"])
    assert np.array_equal(got.quotient_polys_cap, exp["quotient_cap"])
    assert np.array_equal(got.openings.reshape(-1), exp["openings"])
    assert np.array_equal(got.opening_proof, exp["fri"])
    assert ch.get_challenge() == L.orc_challenger_get(C.byref(och))
    return exp, got


LOOKUP_A = ([("single", 0), ("single", 1), ("next", 2)], ("single", 3), ("single", 4),
            [None, ("simple", ("single", 10)), ("full", [(("single", 10), ("single", 11))], [])])
CTL_MULTI = [([("single", 0), ("lc", [(1, 3), (2, 5)], [(3, 7)], 11)], ("simple", ("single", 10))),
             ([("single", 4), ("single", 5)], None),
             ([("next", 6), ("single", 7)], ("full", [(("single", 10), ("single", 11))], []))]
CTL_SINGLE = [([("single", 1), ("single", 2), ("single", 3)], ("simple", ("single", 11)))]


@pytest.mark.parametrize("hasher", [0, 1])
def test_generic_machinery_no_air(oracle, hasher):
    # AIR_NONE: only lookup + CTL constraints; 12 columns, cols 10/11 binary filters
    _run_case(oracle, 0, 12, 6, hasher, [LOOKUP_A], [CTL_MULTI, CTL_SINGLE], seed=5, binary_cols=(10, 11))


def test_minimum_height(oracle):
    # 2^3 rows: the LDE has 2^4 leaves, so the cap (height 4) IS the leaf digests and every Merkle path is empty;
    # one row fewer and plonky2 (and zk_commit_columns) refuse the cap height
    import zk_evm_amd as zk
    _run_case(oracle, 0, 12, 3, 0, [LOOKUP_A], [CTL_MULTI, CTL_SINGLE], seed=21, binary_cols=(10, 11))
    _run_case(oracle, 3, 30, 3, 1, [], [], seed=22)
    import torch
    with pytest.raises(zk.ZkStarkError, match="cap_height 4 exceeds tree height 3"):
        zk.PolynomialBatch.from_values(torch.zeros((5, 4), dtype=torch.int64, device="cuda"), 1, False, 4)


def test_no_aux_at_all(oracle):
    _run_case(oracle, 0, 5, 5, 0, [], [], seed=6)


def test_lookups_only_and_ctl_only(oracle):
    _run_case(oracle, 0, 12, 5, 0, [LOOKUP_A, LOOKUP_A], [], seed=7, binary_cols=(10, 11))
    _run_case(oracle, 0, 12, 7, 0, [], [CTL_SINGLE], seed=8, binary_cols=(10, 11))


@pytest.mark.parametrize("hasher", [0, 1])
def test_mem_continuation_table(oracle, hasher):
    # MemBefore / MemAfter: ctl_data + ctl_filter of memory_continuation_stark.rs:28-39 (looked side of CTL 7/8)
    ctl = [([("single", 1), ("single", 2), ("single", 3)] + [("single", 4 + i) for i in range(8)],
            ("simple", ("single", 0)))]
    _run_case(oracle, 1, 12, 7, hasher, [], [ctl], seed=9, binary_cols=(0,))


def test_logic_table(oracle):
    # Logic: ctl_data (logic.rs:84-113): opcode combination, 2 x 8 le_bits limbs, 8 result limbs
    cols = [("lc", [(0, 0x16), (1, 0x17), (2, 0x18)], [], 0)]
    for base in (3, 259):
        for limb in range(8):
            cols.append(("lc", [(base + 32 * limb + i, 1 << i) for i in range(32)], [], 0))
    cols += [("single", 515 + i) for i in range(8)]
    ctl = [(cols, ("simple", ("lc", [(0, 1), (1, 1), (2, 1)], [], 0)))]
    # make the op flags one-hot-or-zero so the CTL filter (their sum) is binary
    def fix(trace, rng):
        n = trace.shape[1]
        which = rng.integers(0, 4, size=n)
        for k in range(3):
            trace[k] = (which == k).astype(np.uint64)
    _run_case(oracle, 2, 523, 5, 0, [], [ctl], seed=10, trace_fix=fix)


def test_memory_table(oracle):
    # lookups(): memory_stark.rs:858-885; CTL looked data: memory_stark.rs:35-60 (is_read, ctx, seg, virt, 8 limbs, ts)
    lk1 = ([("single", 27), ("next", 6)], ("single", 28), ("single", 29),
           [None, ("simple", ("lc", [(15, 1), (16, 1)], [], 0))])
    lk2 = ([("lc", [(4, 1)], [], 1)], ("single", 21), ("single", 23), [("simple", ("single", 24))])
    ctl = [([("single", 3), ("single", 4), ("single", 5), ("single", 6)] + [("single", 7 + i) for i in range(8)] +
            [("single", 1)], ("simple", ("single", 0)))]

    def fix(trace, rng):
        n = trace.shape[1]
        which = rng.integers(0, 3, size=n)
        trace[15] = (which == 0).astype(np.uint64)
        trace[16] = (which == 1).astype(np.uint64)
        trace[24] = rng.integers(0, 2, size=n, dtype=np.uint64)
        trace[0] = rng.integers(0, 2, size=n, dtype=np.uint64)
    _run_case(oracle, 3, 30, 6, 0, [lk1, lk2], [ctl], seed=12, trace_fix=fix)


def test_byte_packing_table(oracle):
    # lookups(): byte_packing_stark.rs:426-437 (32 value bytes range-checked against range_counter)
    lk = ([("single", 37 + i) for i in range(32)], ("single", 69), ("single", 70), [None] * 32)
    # CTL looked side (ctl_looked_data, byte_packing_stark.rs:55-90): filter = sum of index_len
    ctl = [([("single", 0), ("single", 33), ("single", 34), ("single", 35),
             ("lc", [(1 + i, i + 1) for i in range(32)], [], 0), ("single", 36)] +
            [("lc", [(37 + 4 * l + k, 1 << (8 * k)) for k in range(4)], [], 0) for l in range(8)],
            ("simple", ("lc", [(1 + i, 1) for i in range(32)], [], 0)))]

    def fix(trace, rng):
        n = trace.shape[1]
        which = rng.integers(0, 33, size=n)
        for i in range(32):
            trace[1 + i] = (which == i).astype(np.uint64)
    _run_case(oracle, 4, 71, 5, 0, [lk], [ctl], seed=13, trace_fix=fix)


def test_arithmetic_table(oracle):
    # lookups(): arithmetic_stark.rs:320-327 (96 shared columns range-checked against RANGE_COUNTER);
    # CTL looked side: arithmetic_stark.rs:33-117 (opcode + 4 registers packed as x + 2^16 y limb pairs)
    lk = ([("single", 18 + i) for i in range(96)], ("single", 114), ("single", 115), [None] * 96)
    cols = [("single", 17)]
    for reg in (18, 34, 50, 66):
        cols += [("lc", [(reg + 2 * k, 1), (reg + 2 * k + 1, 1 << 16)], [], 0) for k in range(8)]
    ctl = [(cols, ("simple", ("lc", [(i, 1) for i in range(17)], [], 0)))]

    def fix(trace, rng):
        n = trace.shape[1]
        which = rng.integers(0, 18, size=n)
        for i in range(17):
            trace[i] = (which == i).astype(np.uint64)
    _run_case(oracle, 5, 116, 5, 0, [lk], [ctl], seed=14, trace_fix=fix)


def test_keccak_table(oracle):
    # CTL looked entries: keccak_stark.rs:38-49 (inputs: reg_input_limb x 50 + TIMESTAMP, filter = round flag 0;
    # outputs: reg_output_limb x 50 + TIMESTAMP, filter = round flag 23)
    from oracle.airs import k_reg_a, k_reg_a_ppp
    inputs = [("single", k_reg_a((i // 2) % 5, (i // 2) // 5) + (i % 2)) for i in range(50)] + [("single", 24)]
    outputs = [("single", k_reg_a_ppp((i // 2) % 5, (i // 2) // 5) + (i % 2)) for i in range(50)] + [("single", 24)]
    ctl_in = [(inputs, ("simple", ("single", 0)))]
    ctl_out = [(outputs, ("simple", ("single", 23)))]

    def fix(trace, rng):
        n = trace.shape[1]
        which = rng.integers(0, 25, size=n)
        for i in range(24):
            trace[i] = (which == i).astype(np.uint64)
    _run_case(oracle, 6, 2431, 4, 0, [], [ctl_in, ctl_out], seed=15, trace_fix=fix)


def test_keccak_sponge_table(oracle):
    # lookups(): keccak_sponge_stark.rs:946-953 -- block_bytes (136) + updated_digest_state_bytes ... range-checked
    # against range_counter; here: the 136 block bytes (the real definition is exercised in the
    # segment-level tests); one CTL looked entry (ctl_looked_data shape: 13 columns, filter = is_final block)
    lk = ([("single", 192 + i) for i in range(136)], ("single", 436), ("single", 437), [None] * 136)
    cols = [("single", 1), ("single", 2), ("single", 3),
            ("lc", [(5, 1)] + [(6 + i, 1) for i in range(136)], [], 0), ("single", 4)] + \
           [("lc", [(404 + 4 * k + i, 1 << (8 * i)) for i in range(4)], [], 0) for k in range(8)]
    ctl = [(cols, ("simple", ("single", 6 + 135)))]

    def fix(trace, rng):
        n = trace.shape[1]
        trace[6 + 135] = rng.integers(0, 2, size=n, dtype=np.uint64)
    _run_case(oracle, 7, 438, 4, 0, [lk], [ctl], seed=16, trace_fix=fix)


def test_cpu_table(oracle):
    # CpuStark (cpu/cpu_stark.rs:594-626): all 18 constraint modules, 85 columns; the four kernel-label
    # constants are passed as air_consts.  CTLs: two of the CPU's looking shapes -- a GP memory channel
    # (ctl_data_gp_memory: is_read, ctx, seg, virt, value[8], timestamp = clock*NUM_CHANNELS + channel;
    # filter = channel.used) and the logic CTL (opcode-derived column + 24 value limbs, filter = logic_op).
    from oracle import airs as oairs
    ch0 = 41
    mem_cols = [("single", ch0 + 1), ("single", ch0 + 2), ("single", ch0 + 3), ("single", ch0 + 4)] + \
               [("single", ch0 + 5 + i) for i in range(8)] + [("lc", [(40, 4)], [], 0)]
    ctl_mem = [(mem_cols, ("simple", ("single", ch0 + 0)))]
    logic_cols = [("lc", [(24 + i, 1 << i) for i in range(8)], [], 0)] + \
                 [("single", 41 + 13 * k + 5 + i) for k in range(2) for i in range(8)] + \
                 [("next", 41 + 5 + i) for i in range(8)]
    ctl_logic = [(logic_cols, ("simple", ("single", 10)))]

    def fix(trace, rng):
        n = trace.shape[1]
        for c in list(range(6, 24)) + [ch0, 4] + list(range(24, 32)):
            trace[c] = rng.integers(0, 2, size=n, dtype=np.uint64)
    _run_case(oracle, 8, 85, 5, 0, [], [ctl_mem, ctl_logic], seed=17, trace_fix=fix,
              air_consts=oairs.CPU_TEST_CONSTS)


def test_cpu_table_cdk_erigon(oracle):
    # The 86-column Cpu table of a `cdk_erigon` build (AIR id 10: `poseidon` flag at column 14, everything after it one
    # further; contextops / control_flow / decode / gas / stack gain their poseidon terms, jumps loses the
    # JUMPDEST-bit read).  CTLs: the three looking shapes this feature adds, cpu_stark.rs:465-544.
    from oracle import airs as oairs
    ch = lambda k: 42 + 13 * k
    pos_flag, bit0, clock = 14, 25, 41
    simple_cols = [("lc", [(ch(k) + 5 + 2 * i, 1), (ch(k) + 5 + 2 * i + 1, 1 << 32)], [], 0) for k in range(3) for i in range(4)] + \
                  [("next", ch(0) + 5 + i) for i in range(8)]
    f_simple = ("full", [(("single", pos_flag), ("lc", [(bit0, 0xFFFFFFFF00000001 - 1)], [], 1))], [])
    f_general = ("full", [(("single", pos_flag), ("single", bit0))], [])
    general_in = [("single", ch(0) + 5 + 2), ("single", ch(0) + 5 + 1), ("single", ch(0) + 5), ("single", ch(1) + 5),
                  ("lc", [(clock, 5)], [], 0)]
    general_out = [("next", ch(0) + 5 + i) for i in range(8)] + [("lc", [(clock, 5)], [], 0)]

    def fix(trace, rng):
        n = trace.shape[1]
        for c in list(range(6, 25)) + [ch(0), 4] + list(range(25, 33)):
            trace[c] = rng.integers(0, 2, size=n, dtype=np.uint64)
    _run_case(oracle, 10, 86, 5, 0, [], [[(simple_cols, f_simple)], [(general_in, f_general)], [(general_out, f_general)]],
              seed=19, trace_fix=fix, air_consts=oairs.CPU_TEST_CONSTS)
