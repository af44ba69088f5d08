"""CPU: the consistent nine-table segment (tests/consistent_segment.py) and the oracle's restatement of the
reference's top-level verifier pieces -- `get_memory_extra_looking_sum` (verifier.rs:319-512) and
`verify_cross_table_lookups` -- without proving anything: the CTL sums are taken straight from the rows."""
import numpy as np
import pytest

from oracle import airs as oairs
from oracle import all_stark as A
from oracle import mem_trace as mt
from oracle import segment as oseg
from oracle import stark as S
from tests import consistent_segment as cs
from tests.test_oracle_tracegen import _check_air

KH = 0x1F2E3D4C5B6A79880102030405060708090A0B0C0D0E0F101112131415161718


def test_public_memory_writes_shape():
    pv = cs.make_public_values(np.random.default_rng(0))
    w = oseg.public_memory_writes(pv, KH, 1234)
    assert len(w) == 25 + 8 + 256 + 12                      # verifier.rs:330-494 (eth_mainnet)
    assert len({(s, i) for s, i, _ in w}) == len(w)         # distinct addresses
    assert (5, 45, KH) in w and (5, 46, 1234) in w and (33, 6, 31337) in w
    assert all(0 <= v < 1 << 256 for _, _, v in w)


def test_consistent_segment_rows_and_ctls_balance():
    rng = np.random.default_rng(1)
    code = rng.bytes(300)
    traces, pv, before = cs.build(rng, oairs.CPU_TEST_CONSTS[0], code, KH)
    assert [t.shape[0] for t in traces] == list(A.TABLE_COLUMNS)
    _check_air(oairs.make_eval_cpu(*oairs.CPU_TEST_CONSTS), traces[A.CPU])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    _check_air(oairs.AIRS[1][0], traces[A.MEM_BEFORE])
    _check_air(oairs.AIRS[1][0], traces[A.MEM_AFTER])
    m = traces[A.MEMORY]
    assert int(m[mt.FILTER].sum()) == len(before) + 301    # initialisation + public-value writes; dummies unfiltered
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(0x1234567890ABCDEF % S.P, 987654321987), S.GrandProductChallenge(31, 0xFFFF0000FFFF)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    ok, why = oseg.verify_cross_table_lookups(ctls, zf, extra, 2)
    assert not ok and why.startswith("CTL 6")               # without the public-value writes Memory does not balance
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    # a different public value (one block hash) breaks exactly the Memory CTL
    pv2 = dict(pv, prev_hashes=[bytes(32)] + pv["prev_hashes"][1:])
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv2, c, KH, len(code)) for c in ch]
    ok, why = oseg.verify_cross_table_lookups(ctls, zf, extra, 2)
    assert not ok and why.startswith("CTL 6")


def test_segment_with_an_executing_cpu_table(oracle):
    """A Cpu table that really runs instructions (PC PC PC ADD XOR PC PC ADD KECCAK_GENERAL PUSH32 MSTORE_32BYTES POP
    in kernel mode, then halts): its rows satisfy all 514 constraints of the restated Cpu AIR, and its bus traffic --
    twelve code reads of the kernel image, stack writes through the partial channel, stack reads, two ADDs, one XOR,
    one KECCAK_GENERAL over three bytes of the kernel image (a KeccakSponge row, its Keccak-f permutation, its five
    block XORs, its byte reads), one MSTORE_32BYTES of the digest (a BytePacking row and its 32 byte writes) --
    balances all ten CTLs of the real wiring with ALL NINE tables live (nine CTLs with traffic; context pruning in
    the fourth kernel).  A wrong gas charge or stack pointer breaks the
    AIR; a wrong sum breaks exactly the Arithmetic CTL, a wrong XOR the Logic CTL, a wrong digest the KeccakSponge CTL,
    a wrong stored byte the BytePacking / Memory CTLs; executing a byte that is not in the kernel image breaks the
    Memory CTL."""
    rng = np.random.default_rng(2)
    traces, pv, code = cs.build_with_cpu_program(rng, oracle, KH)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_CONSTS)
    _check_air(air, traces[A.CPU])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    _check_air(oairs.AIRS[2][0], traces[A.LOGIC])
    _check_air(oairs.eval_keccak_sponge, traces[A.KECCAK_SPONGE])
    _check_air(oairs.AIRS[6][0], traces[A.KECCAK])
    _check_air(oairs.eval_byte_packing, traces[A.BYTE_PACKING])
    assert int(traces[A.BYTE_PACKING][32].sum()) == 1                          # one 32-byte operation
    assert int(traces[A.ARITHMETIC][0].sum()) == 2 and int(traces[A.CPU][6:24].sum()) == cs.CPU_EXECUTED
    assert int(traces[A.LOGIC][2].sum()) == 1 + 5 and int(traces[A.KECCAK][0].sum()) == 1
    # the digest the Cpu pushed is keccak256 of the three hashed bytes, read big-endian
    digest = sum(int(traces[A.CPU][41 + 5 + i, 9]) << (32 * i) for i in range(8))
    assert digest == int.from_bytes(oracle.keccak256(code[11:14]), "big")
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]

    def balance(trs):
        zf = cs.ctl_first_values(trs, ctls, ch)
        extra = [[0, 0] for _ in ctls]
        extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
        return oseg.verify_cross_table_lookups(ctls, zf, extra, 2), zf
    (ok, why), zf = balance(traces)
    assert ok, why
    assert all(z != 0 for z in zf[A.CPU][0:2]) and all(z != 0 for z in zf[A.KECCAK_SPONGE][0:2]) and all(zf[A.KECCAK])
    for col, row in ((5, 4), (3, 2), (40, 7)):                                # gas, stack_len, clock
        bad = traces[A.CPU].copy()
        bad[col, row] += np.uint64(1)
        with pytest.raises(AssertionError):
            _check_air(air, bad)

    def first_failure(table, col, row, value=None):
        bad = [t.copy() for t in traces]
        bad[table][col, row] = bad[table][col, row] ^ np.uint64(4) if value is None else value
        return balance(bad)[0]
    assert first_failure(A.CPU, 41 + 5, 4, 4) == (False, "CTL 0 challenge 0")          # 2 + 1 = 4
    assert first_failure(A.CPU, 41 + 5, 9)[1].startswith("CTL 2")                      # another digest on the stack
    assert first_failure(A.LOGIC, 515, 0)[1].startswith("CTL 5")                       # another XOR result
    assert first_failure(A.KECCAK, 2429, 23)[1].startswith("CTL 4")                    # another permutation output
    assert first_failure(A.BYTE_PACKING, 37, 0)[1].startswith("CTL 1")                 # another stored byte
    # the stored digest is in the final memory: 32 bytes at (7, 11, 5..36)
    after = traces[A.MEM_AFTER]
    stored = {int(after[3, r]): int(after[4, r]) for r in range(after.shape[1]) if after[0, r] and (after[1, r], after[2, r]) == (7, 11)}
    assert bytes(stored.get(5 + i, 0) for i in range(32)) == oracle.keccak256(code[11:14])
    assert first_failure(A.CPU, 24, 0)[1].startswith("CTL 6")                          # opcode 0x59 claimed at pc 0


def test_cdk_erigon_segment_with_an_executing_cpu_table(oracle):
    """The `cdk_erigon` feature set with a live Cpu table: PC PC PC POSEIDON POP in the 86-column layout satisfies the
    531-constraint variant of the Cpu AIR; the Poseidon table's PoseidonSimpleOp row matches it through CTL 10
    (ctl_poseidon_simple: twelve 64-bit input elements from three stack words, eight digest limbs); the extra looking
    sum of the Memory CTL is the cdk_erigon one (burn address written, no eth_mainnet metadata)."""
    from oracle import poseidon_table as pt
    traces, pv, code = cs.build_cdk_erigon_with_cpu_program(np.random.default_rng(3), oracle, KH)
    reg = A.Registry(True)
    assert [t.shape[0] for t in traces] == list(reg.TABLE_COLUMNS)
    air = oairs.make_eval_cpu(*cs.ERIGON_CONSTS, cdk_erigon=True)
    _check_air(air, traces[A.CPU])
    _check_air(pt.eval_poseidon, traces[9])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    words = [0, 1, 2][::-1]                                             # stack top first: PC pushed 0, 1, 2
    inp = [((w >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for w in words for i in range(4)]
    digest = [int(v) for v in oracle.poseidon_permute(inp)[:4]]
    pushed = sum(int(traces[A.CPU][42 + 5 + i, 4]) << (32 * i) for i in range(8))
    assert pushed == sum(v << (64 * i) for i, v in enumerate(digest))
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]

    def balance(trs, pvx=pv):
        zf = cs.ctl_first_values(trs, reg.ctls, ch)
        extra = [[0, 0] for _ in reg.ctls]
        extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pvx, c, KH, len(code)) for c in ch]
        return oseg.verify_cross_table_lookups(reg.ctls, zf, extra, 2), zf
    (ok, why), zf = balance(traces)
    assert ok, why
    assert all(zf[9][2:4]) and not any(zf[9][:2])                        # simple-op CTL live, no Poseidon memory reads
    bad = [t.copy() for t in traces]
    bad[9][pt.DIGEST_COL, 0] ^= np.uint64(1)
    assert balance(bad)[0] == (False, "CTL 10 challenge 0")
    # the mainnet reading of the same public values (blob-gas / beacon-root writes, no burn address) does not balance
    pv_mainnet = dict(pv, burn_addr=None)
    assert balance(traces, pv_mainnet)[0][1].startswith("CTL 6")
    # the same rows under the eth_mainnet Cpu AIR make no sense (columns shifted)
    with pytest.raises(AssertionError):
        _check_air(oairs.make_eval_cpu(*cs.ERIGON_CONSTS), traces[A.CPU][:85])


def test_second_kernel_covers_the_remaining_cpu_modules(oracle):
    """CPU_PROGRAM_2: thirty kernel-mode instructions (PC DUP3 ISZERO SWAP1 SHL NOT PUSH0 SUB MUL DUP1 ADDMOD GT PUSH32
    MSTORE_GENERAL MLOAD_GENERAL GET_CONTEXT OR JUMPI JUMPDEST JUMP POP ...) generated by tests/kernel_run.py from the
    reference's witness conventions.  Every row satisfies the restated Cpu AIR -- this time exercising dup_swap,
    simple_logic, shift, push0, memio, contextops and jumps non-vacuously -- and the run's traffic (five Arithmetic
    operations incl. SHL with its shift-table read and a two-row ADDMOD, an OR, a general store / load pair, stack
    spills through the partial channel and GP channels 1 and 2) balances the CTLs."""
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(5), oracle, KH, cs.CPU_PROGRAM_2,
                                                 cs.CPU_PROGRAM_2_CONSTS[0], 32)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_2_CONSTS)
    _check_air(air, traces[A.CPU])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    cpu = traces[A.CPU]
    flags = {name: int(cpu[6 + i].sum()) for i, name in enumerate(oairs.C_OPS)}
    assert flags == dict(binary_op=3, ternary_op=1, fp254_op=0, eq_iszero=1, logic_op=1, not_pop=2, shift=1,
                         jumpdest_keccak_general=2, jumps=2, push_prover_input=4, dup_swap=4, context_op=1, m_op_32bytes=0,
                         exit_kernel=0, m_op_general=2, pc_push0=6, syscall=0, exception=0)
    assert int(cpu[5, -1]) == 89 and int(cpu[2, -1]) == 162 and int(cpu[3, -1]) == 0        # gas, halt pc, empty stack
    assert [int(traces[A.ARITHMETIC][i].sum()) for i in (1, 2, 5, 12, 14)] == [1, 1, 1, 1, 1]   # MUL SUB ADDMOD GT SHL
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    # each module notices its own kind of mistake
    def violates(col, row, delta=1):
        bad = cpu.copy()
        bad[col, row] = (int(bad[col, row]) + delta) % oseg.P
        try:
            _check_air(air, bad)
        except AssertionError:
            return True
        return False
    rows = {int(cpu[2, r]): r for r in range(cpu.shape[1]) if int(cpu[6:24, r].sum())}       # pc -> row
    assert violates(41 + 13 * 2 + 4, rows[3])           # DUP3 reads another stack slot
    assert violates(41 + 5, rows[4] + 1)                # ISZERO's result
    assert violates(41 + 13 * 2 + 4, rows[6])           # SHL's shift-table address
    assert violates(41 + 5, rows[7] + 1)                # NOT's result
    assert violates(80 + 4, rows[49])                   # MSTORE_GENERAL's target address
    assert violates(2, rows[119] + 1)                   # JUMPI lands elsewhere
    assert violates(41 + 5 + 2, rows[84] + 1)           # GET_CONTEXT pushes another context


def test_third_kernel_leaves_kernel_mode(oracle):
    """CPU_PROGRAM_3: EXIT_KERNEL into user code -- PUSH1 PUSH1 JUMP JUMPDEST ADDRESS -- and back through a syscall.
    Exercises, non-vacuously, the user-mode halves of the Cpu AIR (decode availability, `is_not_kernel`, the stack
    bound `stack_len_bounds_aux`, the JUMPDEST-bit read of jumps.rs, syscalls_exceptions.rs, the exit_kernel part of
    jumps.rs) and the CTLs only user code drives: BytePacking `push` (arguments read from the code) and `jumptable`
    (the 3-byte handler offset), the Arithmetic range check of the pushed kexit_info, the JumpdestBits read in Memory."""
    kw = dict(extra_memory=cs.CPU_PROGRAM_3_MEMORY, syscall_jumptable=cs.CPU_PROGRAM_3_CONSTS[2], syscall_opcodes=(0x30,))
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(6), oracle, KH, cs.CPU_PROGRAM_3,
                                                 cs.CPU_PROGRAM_3_CONSTS[0], 16, **kw)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_3_CONSTS)
    cpu = traces[A.CPU]
    _check_air(air, cpu)
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    _check_air(oairs.eval_byte_packing, traces[A.BYTE_PACKING])
    assert [int(v) for v in cpu[4, :10]] == [1, 1, 0, 0, 0, 0, 0, 1, 1, 1]          # kernel, 5 user rows, kernel again
    assert int(cpu[6 + oairs.C_OPS.index("syscall")].sum()) == 1 and int(cpu[6 + oairs.C_OPS.index("exit_kernel")].sum()) == 1
    assert int(traces[A.ARITHMETIC][16].sum()) == 1                                  # one range-check row
    assert int(traces[A.BYTE_PACKING][1:33].sum()) == 3                              # two PUSH1 arguments + the jump-table entry
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]

    def balance(trs):
        zf = cs.ctl_first_values(trs, ctls, ch)
        extra = [[0, 0] for _ in ctls]
        extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
        return oseg.verify_cross_table_lookups(ctls, zf, extra, 2)
    assert balance(traces) == (True, "")
    bad = [t.copy() for t in traces]
    bad[A.CPU][41 + 5, 3] = 6                          # the value PUSH1 5 left on the stack: BytePacking `push` CTL
    assert balance(bad)[1].startswith("CTL 1")
    bad = [t.copy() for t in traces]
    bad[A.CPU][41 + 13 + 5, 6] = 61                    # the handler offset claimed by the syscall row: `jumptable` CTL
    bad[A.CPU][2, 7] = 61
    assert balance(bad)[1].startswith("CTL")
    for col, row in ((4, 3), (39, 2), (41 + 13 * 2, 4)):   # user row claims kernel mode; stack bound aux; JUMPDEST-bit channel unused
        b2 = cpu.copy()
        b2[col, row] = (int(b2[col, row]) + 1) % oseg.P
        with pytest.raises(AssertionError):
            _check_air(air, b2)


def test_fourth_kernel_switches_and_prunes_a_context(oracle):
    """CPU_PROGRAM_4: SET_CONTEXT to context 1, stack traffic there, SET_CONTEXT back to context 0 with the prune flag.
    contextops.rs is exercised for real (stack-pointer save / restore through ContextMetadata::StackSize, the new top
    fetched through GP channel 2), the Memory generator marks context 1 stale, MemAfter forgets it, and the
    context-pruning CTL -- idle in the other runs -- carries the pruned context from Memory to the Cpu table."""
    from oracle import mem_trace as mt
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(7), oracle, KH, cs.CPU_PROGRAM_4,
                                                 cs.CPU_PROGRAM_4_CONSTS[0], 16)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_4_CONSTS)
    cpu, mem = traces[A.CPU], traces[A.MEMORY]
    _check_air(air, cpu)
    _check_air(oairs.eval_memory, mem)
    assert [int(v) for v in cpu[0, :9]] == [0, 0, 0, 1, 1, 1, 1, 1, 0]               # the context register
    assert int(mem[mt.IS_PRUNED].sum()) == 1 and int(mem[mt.STALE_CONTEXTS].max()) == 2
    after = traces[A.MEM_AFTER]
    assert not any(int(after[1, r]) == 1 for r in range(after.shape[1]) if after[0, r])   # nothing of context 1 survives
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    assert all(zf[A.CPU][-2:]) and zf[A.CPU][-2:] == zf[A.MEMORY][-2:]               # CTL 9 carries the pruned context
    bad = [t.copy() for t in traces]
    bad[A.CPU][32, 7] = 0                                                             # the Cpu row forgets the prune flag
    zf = cs.ctl_first_values(bad, ctls, ch)
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2)[1].startswith("CTL 9")


def test_fifth_kernel_mload_32bytes(oracle):
    """MLOAD_32BYTES: 32 bytes of the kernel image packed big-endian through the BytePacking `pack` looker
    (cpu_stark.rs:150-176) -- the one CTL entry shape the other runs do not touch."""
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(8), oracle, KH, cs.CPU_PROGRAM_5,
                                                 cs.CPU_PROGRAM_5_CONSTS[0], 8)
    _check_air(oairs.make_eval_cpu(*cs.CPU_PROGRAM_5_CONSTS), traces[A.CPU])
    _check_air(oairs.eval_byte_packing, traces[A.BYTE_PACKING])
    assert int(traces[A.BYTE_PACKING][0, 0]) == 1                                     # a read operation
    loaded = sum(int(traces[A.CPU][41 + 5 + i, 3]) << (32 * i) for i in range(8))
    assert loaded == int.from_bytes(code[5:37], "big")
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    bad = [t.copy() for t in traces]
    bad[A.CPU][41 + 5, 3] ^= np.uint64(1)
    zf = cs.ctl_first_values(bad, ctls, ch)
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2)[1].startswith("CTL 1")


def test_sixth_kernel_exception(oracle):
    """User code executes an invalid opcode: the exception half of syscalls_exceptions.rs (exc_code bits, handler from
    exception_jumptable + 3 * code, the pushed info holds pc -- not pc + 1 --, user gas dropped) plus the same two
    CTL shapes a syscall uses (BytePacking `jumptable`, Arithmetic range check)."""
    kw = dict(syscall_jumptable=100, exception_jumptable=300, exception_opcodes={0xfe: 3})
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(9), oracle, KH, cs.CPU_PROGRAM_6,
                                                 cs.CPU_PROGRAM_6_CONSTS[0], 8, **kw)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_6_CONSTS)
    cpu = traces[A.CPU]
    _check_air(air, cpu)
    r = next(i for i in range(8) if cpu[6 + oairs.C_OPS.index("exception"), i])
    assert [int(cpu[32 + i, r]) for i in range(3)] == [1, 1, 0] and int(cpu[2, r + 1]) == 60 and int(cpu[4, r + 1]) == 1
    info = sum(int(cpu[41 + 5 + i, r + 1]) << (32 * i) for i in range(8))
    assert info & 0xFFFFFFFF == int(cpu[2, r]) and (info >> 32) & 1 == 0 and info >> 192 == int(cpu[5, r])
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    bad = cpu.copy()
    bad[32, r] = 0                                                  # exception code 2 instead of 3: another handler slot
    with pytest.raises(AssertionError):
        _check_air(air, bad)


def test_seventh_kernel_fp254(oracle):
    """ADDFP254 MULFP254 SUBFP254: modfp254.rs live (channel 2 shows the BN254 modulus), three two-row modular operations
    in the Arithmetic table through the real CTL (SUBFP254 with a negative difference: 3 - 4 = p - 1)."""
    traces, pv, code = cs.build_with_cpu_program(np.random.default_rng(10), oracle, KH, cs.CPU_PROGRAM_7,
                                                 cs.CPU_PROGRAM_7_CONSTS[0], 16)
    air = oairs.make_eval_cpu(*cs.CPU_PROGRAM_7_CONSTS)
    _check_air(air, traces[A.CPU])
    assert [int(traces[A.ARITHMETIC][i].sum()) for i in (7, 8, 9)] == [1, 1, 1]
    from tests.kernel_run import BN254
    top = sum(int(traces[A.CPU][41 + 5 + i, 7]) << (32 * i) for i in range(8))        # what SUBFP254 left for MULFP254
    assert top == BN254 - 1
    ctls = A.build_ctls()
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, ctls, ch)
    extra = [[0, 0] for _ in ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(ctls, zf, extra, 2) == (True, "")
    bad = traces[A.CPU].copy()
    bad[41 + 13 * 2 + 5, 3] ^= np.uint64(1)                          # channel 2 shows another modulus
    with pytest.raises(AssertionError):
        _check_air(air, bad)


def test_cdk_erigon_poseidon_general(oracle):
    """POSEIDON_GENERAL on the 86-column Cpu table: the Poseidon table's general operation reads its 56 input bytes from
    Memory (the 56 extra Memory lookers of the cdk_erigon wiring), is matched to the Cpu row's (address, len, timestamp)
    by CTL 11 and hands back the digest by CTL 12; the digest equals the reference's `poseidon_hash_padded_byte_vec`."""
    from oracle import poseidon_table as pt
    traces, pv, code = cs.build_cdk_erigon_with_cpu_program(np.random.default_rng(11), oracle, KH, cs.ERIGON_PROGRAM_2,
                                                            cs.ERIGON_CONSTS_2, 8)
    reg = A.Registry(True)
    _check_air(oairs.make_eval_cpu(*cs.ERIGON_CONSTS_2, cdk_erigon=True), traces[A.CPU])
    _check_air(pt.eval_poseidon, traces[9])
    _check_air(oairs.eval_memory, traces[A.MEMORY])
    cap = [0, 0, 0, 0]
    st = [int.from_bytes(code[7 * i:7 * i + 7], "little") for i in range(8)] + cap
    digest = [int(v) for v in oracle.poseidon_permute(st)[:4]]
    pushed = sum(int(traces[A.CPU][42 + 5 + i, 3]) << (32 * i) for i in range(8))
    assert pushed == sum(v << (64 * i) for i, v in enumerate(digest))
    ch = [S.GrandProductChallenge(1234567, 7654321), S.GrandProductChallenge(99, 101)]
    zf = cs.ctl_first_values(traces, reg.ctls, ch)
    extra = [[0, 0] for _ in reg.ctls]
    extra[oseg.MEMORY_CTL_IDX] = [oseg.get_memory_extra_looking_sum(pv, c, KH, len(code)) for c in ch]
    assert oseg.verify_cross_table_lookups(reg.ctls, zf, extra, 2) == (True, "")
    assert all(zf[9][0:2]) and not any(zf[9][2:4]) and all(zf[9][4:8])   # memory reads, no simple op, general in / out
    bad = [t.copy() for t in traces]
    bad[9][pt.INPUT_BYTES + 3, 0] ^= np.uint64(1)                        # one input byte differs from memory
    zf = cs.ctl_first_values(bad, reg.ctls, ch)
    assert oseg.verify_cross_table_lookups(reg.ctls, zf, extra, 2)[1].startswith("CTL 6")
