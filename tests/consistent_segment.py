"""A complete, mutually consistent nine-table segment for the reference's top-level acceptance criterion
(`verify_proof`, evm_arithmetization/src/verifier.rs:184-312), built from the restated reference generators:

* Cpu: a halting-only trace (halt.rs: every operation flag 0, clock 1..n, kernel mode, pc = halt_pc) -- it looks up
  nothing, so its CTL sums are all zero;
* Memory: the MemBefore initialisation writes (timestamp 0) plus exactly the public-value writes the verifier adds
  as the "extra looking sum" (block metadata, trie-root digests, bloom, the 256 previous block hashes, the
  registers before/after; timestamp 2), run through fill_gaps / padding / flags (oracle/mem_trace.py);
* MemBefore: the kernel image + shift table (the `initial_memory_merkle_cap` content, verifier.rs:14-78);
* MemAfter: what the Memory generator reports as the final memory;
* Arithmetic, Keccak, KeccakSponge, Logic: empty-operation tables (padding rows + range-check columns);
  BytePacking: likewise, but only usable with table_in_use[1] = False (its AIR wants an operation on row 0).

Every CTL then balances only because of `get_memory_extra_looking_sum` -- the piece of the verifier no single-table
test exercises.  TEST INFRASTRUCTURE (uses oracle/)."""
import numpy as np

from oracle import arith_trace, keccak_trace, mem_trace
from oracle import segment as oseg
from oracle import tracegen as otg

SEG_CODE, SEG_SHIFT_TABLE = 0, 13


def halting_cpu_trace(n, halt_pc):
    t = np.zeros((85, n), dtype=np.uint64)
    t[2] = halt_pc                      # program_counter
    t[4] = 1                            # is_kernel_mode
    t[40] = np.arange(1, n + 1, dtype=np.uint64)
    return t


def continuation_table(rows):
    """memory_continuation_stark.rs:53-98: filter, ctx, seg, virt, value[8]; padded to max(128, pow2)."""
    n = max(128, 1 << max(len(rows) - 1, 0).bit_length())
    t = np.zeros((12, n), dtype=np.uint64)
    if rows:
        t[:, :len(rows)] = np.array(rows, dtype=np.uint64).T
    return t


def make_public_values(rng):
    return dict(
        roots_before=[rng.bytes(32) for _ in range(3)], roots_after=[rng.bytes(32) for _ in range(3)],
        beneficiary=rng.bytes(20), timestamp=int(rng.integers(1, 1 << 32)), number=int(rng.integers(1, 1 << 32)),
        difficulty=int(rng.integers(0, 1 << 32)), random=rng.bytes(32), gaslimit=int(rng.integers(1, 1 << 32)),
        chain_id=int(rng.integers(1, 1 << 32)), base_fee=int(rng.integers(0, 1 << 63)),
        gas_used=int(rng.integers(0, 1 << 32)), blob_gas_used=int(rng.integers(0, 1 << 63)),
        excess_blob_gas=int(rng.integers(0, 1 << 63)), parent_beacon_root=rng.bytes(32),
        bloom=[int.from_bytes(rng.bytes(32), "big") for _ in range(8)],
        prev_hashes=[rng.bytes(32) for _ in range(256)], cur_hash=rng.bytes(32), checkpoint_root=rng.bytes(32),
        checkpoint_hash=[int(x) for x in rng.integers(0, 1 << 62, 4)], txn_before=3, txn_after=7,
        gas_before=int(rng.integers(0, 1 << 31)), gas_after=int(rng.integers(0, 1 << 32)),
        registers_before=dict(program_counter=4242, is_kernel=1, stack_len=0, stack_top=0, context=0, gas_used=0),
        registers_after=dict(program_counter=31337, is_kernel=1, stack_len=2,
                             stack_top=int.from_bytes(rng.bytes(32), "big"), context=0, gas_used=12345))


def build(rng, halt_pc, kernel_code=b"", kernel_hash=0, cpu_rows=32):
    """-> (traces[9] column-major uint64, pv dict, mem_before rows).  kernel_hash is the value the segment's
    GlobalMetadata::KernelHash write carries (any U256: the kernel itself is out of scope)."""
    pv = make_public_values(rng)
    before = [((0, SEG_CODE, i), b) for i, b in enumerate(kernel_code)]
    before += [((0, SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    ops = [dict(filter=True, timestamp=2, ctx=0, seg=seg, virt=idx, is_read=False, value=val)
           for seg, idx, val in oseg.public_memory_writes(pv, kernel_hash, len(kernel_code))]
    memory, mem_after = mem_trace.generate_trace(ops, before, [])
    before_rows = [[1, c, s, v] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for (c, s, v), val in before]
    keccak_f = None                                        # no sponge operation: never called
    traces = [None] * 9
    traces[0] = arith_trace.generate_trace([])[0]
    traces[1] = otg.byte_packing_generate_trace([], 0)
    traces[2] = halting_cpu_trace(cpu_rows, halt_pc)
    traces[3] = np.ascontiguousarray(keccak_trace.generate_trace_rows([], 32).T)
    traces[4] = otg.keccak_sponge_generate_trace([], 0, keccak_f)
    traces[5] = np.zeros((523, 32), dtype=np.uint64)
    traces[6] = memory
    traces[7] = continuation_table(before_rows)
    traces[8] = continuation_table(mem_after)
    return traces, pv, before_rows


# ---- a segment whose Cpu table really executes instructions ---------------------------------------------------------
# pc 0 PC, 1 PC, 2 PC, 3 ADD, 4 XOR, 5 PC, 6 PC, 7 ADD, 8 KECCAK_GENERAL, 9 PUSH32 <32 bytes at 10..41>,
# 42 MSTORE_32BYTES_32, 43 POP, halt at pc = 44.  The PUSH32 argument is an address word (virt | segment << 32 |
# context << 64) = (context 7, segment 11, virt 5); KECCAK_GENERAL hashes code bytes 11..13, which lie inside it.
STORE_ADDR = (7, 11, 5)
_ADDR_WORD = STORE_ADDR[2] | (STORE_ADDR[1] << 32) | (STORE_ADDR[0] << 64)
CPU_PROGRAM = bytes([0x58, 0x58, 0x58, 0x01, 0x18, 0x58, 0x58, 0x01, 0x21, 0x7f]) + _ADDR_WORD.to_bytes(32, "big") + \
    bytes([0xdf, 0x50, 0x00, 0x00, 0x00, 0x00])
CPU_HALT_PC = 44
CPU_EXECUTED = 12                                                # instructions = Cpu rows before the halting rows
CPU_PROGRAM_CONSTS = (CPU_HALT_PC, 0, 777777, 888888)            # halt_final, init, syscall / exception jumptables


def cpu_program_trace(keccak256, n=16, program=None, halt_pc=None, cdk_erigon=False, poseidon_permute=None,
                      extra_memory=None, return_run=False, **run_kw):
    """The Cpu rows of the kernel-mode run of `program` (default CPU_PROGRAM) and the operation logs it produces, from
    the miniature witness generator tests/kernel_run.py (a restatement of witness/operation.rs for a subset of
    opcodes).  -> (cpu trace, memory ops, arithmetic ops, logic ops, sponge ops, byte-packing ops[, poseidon ops])."""
    from tests.kernel_run import KernelRun
    program = CPU_PROGRAM if program is None else program
    halt_pc = CPU_HALT_PC if halt_pc is None else halt_pc
    memory = {(0, SEG_CODE, i): b for i, b in enumerate(program)}
    memory.update(extra_memory or {})
    run = KernelRun(program, halt_pc, n, keccak256=keccak256, poseidon_permute=poseidon_permute, cdk_erigon=cdk_erigon,
                    memory=memory, **run_kw).run()
    assert not run.stack
    if return_run:
        return run
    out = (run.t, run.mem_ops, run.arith, run.logic, run.sponge, run.packing)
    return out + (run.poseidon,) if cdk_erigon else out


def logic_table(ops, n=32):
    """`LogicStark::generate_trace` rows (logic.rs:165-240): one-hot {and, or, xor}, 2 x 256 input bits, 8 result limbs."""
    t = np.zeros((523, n), dtype=np.uint64)
    for r, (kind, a, b) in enumerate(ops):
        t[kind, r] = 1
        for i in range(256):
            t[3 + i, r], t[259 + i, r] = (a >> i) & 1, (b >> i) & 1
        res = (a & b, a | b, a ^ b)[kind]
        for i in range(8):
            t[515 + i, r] = (res >> (32 * i)) & 0xFFFFFFFF
    return t


def sponge_side_effects(ks):
    """What the KeccakSponge rows ask of the Keccak and Logic tables (keccak_sponge_stark.rs: every absorbing row XORs
    its 136-byte block into the rate -- five 256-bit Logic XORs, `ctl_looking_logic` -- and sends the xored state
    through one Keccak-f permutation tagged with the operation's timestamp)."""
    perms, xors = [], []
    for r in range(ks.shape[1]):
        row = [int(v) for v in ks[:, r]]
        if not (row[0] or row[6 + 135]):                              # is_full_input_block + is_final (padding byte 135 set)
            continue
        words = row[328:362] + row[176:192]                           # xored rate + original capacity: 50 u32
        perms.append(([words[2 * j] | (words[2 * j + 1] << 32) for j in range(25)], row[4]))
        for i in range(5):
            rate = (row[142 + 8 * i:142 + min(8 * i + 8, 34)] + [0] * 8)[:8]
            blk = [sum(row[192 + k + b] << (8 * b) for b in range(min(4, 136 - k))) for k in range(32 * i, min(32 * i + 32, 136), 4)]
            blk = (blk + [0] * 8)[:8]
            word = lambda ls: sum(v << (32 * j) for j, v in enumerate(ls))
            xors.append((2, word(rate), word(blk)))
    return perms, xors


def _push32(v):
    return bytes([0x7f]) + v.to_bytes(32, "big")


# A second kernel: thirty instructions over the rest of the Cpu AIR's modules -- dup_swap, simple_logic (NOT, ISZERO),
# shift (with its shift-table read), push0, memio (MSTORE_GENERAL / MLOAD_GENERAL), contextops (GET_CONTEXT), jumps
# (JUMPI taken, JUMP), ternary and more binary Arithmetic operations, OR through Logic.
_STORE_WORD = 7 | (11 << 32)                                       # (context 0, segment 11, virt 7)
CPU_PROGRAM_2 = (bytes([0x58, 0x58, 0x58, 0x82, 0x15, 0x90, 0x1b, 0x19, 0x5f, 0x03, 0x02, 0x80, 0x58, 0x08, 0x11]) +
                 _push32(_STORE_WORD) + bytes([0x90, 0xfc]) + _push32(_STORE_WORD) + bytes([0xfb, 0xf6, 0x17]) +
                 _push32(123) + bytes([0x57, 0, 0, 0, 0x5b, 0x58]) + _push32(160) + bytes([0x56, 0, 0x5b, 0x50]))
CPU_PROGRAM_2_CONSTS = (162, 0, 777777, 888888)


def _program_3():
    code = bytearray(247)
    code[0:33] = _push32(40)                                        # kexit_info: pc 40, user mode, gas 0
    code[33] = 0xf9                                                 # EXIT_KERNEL
    code[40:50] = bytes([0x60, 0x05, 0x60, 0x30, 0x56, 0, 0, 0, 0x5b, 0x30])   # PUSH1 5, PUSH1 48, JUMP, .., JUMPDEST, ADDRESS
    code[60:62] = bytes([0x50, 0x50])                               # the "handler": POP POP, then halt at 62
    code[100 + 3 * 0x30:100 + 3 * 0x30 + 3] = (60).to_bytes(3, "big")   # syscall_jumptable[0x30] = 60
    return bytes(code)


# A third kernel leaves kernel mode: EXIT_KERNEL to user code that PUSHes (BytePacking-checked arguments), JUMPs (reads
# the JUMPDEST bit of its target) and executes ADDRESS, which is not native: a syscall through the jump table (read
# through BytePacking, kexit_info range-checked through Arithmetic) back into the kernel, which pops and halts.
CPU_PROGRAM_3 = _program_3()
CPU_PROGRAM_3_CONSTS = (62, 0, 100, 888888)                        # halt_final, init, syscall_jumptable, exception_jumptable
CPU_PROGRAM_3_MEMORY = {(0, 14, 48): 1}                            # JumpdestBits of context 0: pc 48 is a JUMPDEST


# A fourth kernel switches contexts: SET_CONTEXT to context 1, some stack traffic there, SET_CONTEXT back with the prune
# flag -- context 1 becomes stale: its Memory rows are flagged, kept out of MemAfter, and the context-pruning CTL (the
# only one the other runs leave idle) carries it.
CPU_PROGRAM_4 = bytes([0x58]) + _push32(1 << 64) + bytes([0xf7, 0x58, 0x58, 0x50]) + _push32(1) + bytes([0xf7, 0x50])
CPU_PROGRAM_4_CONSTS = (73, 0, 777777, 888888)


def _program_6():
    code = bytearray(330)
    code[0:33] = _push32(40)
    code[33] = 0xf9
    code[40:43] = bytes([0x60, 0x09, 0xfe])                        # user code: PUSH1 9, INVALID
    code[60:62] = bytes([0x50, 0x50])
    code[300 + 3 * 3:300 + 3 * 3 + 3] = (60).to_bytes(3, "big")   # exception_jumptable[exc_invalid_opcode = 3] = 60
    return bytes(code)


# A sixth: user code hits an invalid opcode -- an exception (code 3) through the exception jump table.
CPU_PROGRAM_6 = _program_6()
CPU_PROGRAM_6_CONSTS = (62, 0, 100, 300)


# A seventh: the BN254 field operations (modfp254.rs: mem_channels[2] must show the modulus).
CPU_PROGRAM_7 = bytes([0x58, 0x58, 0x58, 0x0c, 0x58, 0x90, 0x0e, 0x0d, 0x50])   # PC PC PC ADDFP254 PC SWAP1 SUBFP254 MULFP254 POP
CPU_PROGRAM_7_CONSTS = (9, 0, 777777, 888888)


def loop_program(n_iterations):
    """A countdown loop: PUSH32 n; L: JUMPDEST PUSH32 1 SWAP1 SUB DUP1 PUSH32 L JUMPI; POP -- seven rows per iteration,
    so that valid tables of a few thousand rows can be produced (one SUB per iteration in Arithmetic, ~13 Memory
    operations per iteration)."""
    loop = 33
    code = _push32(n_iterations) + bytes([0x5b]) + _push32(1) + bytes([0x90, 0x03, 0x80]) + _push32(loop) + bytes([0x57, 0x50])
    return code, len(code)                                          # halt right after the final POP


def hashing_loop_program(n_iterations):
    """A loop whose body hashes three bytes of the kernel image (KECCAK_GENERAL) and stores the digest with
    MSTORE_32BYTES: thirteen rows per iteration that keep every table busy -- one KeccakSponge row, one Keccak
    permutation (24 rows), five Logic XORs, one BytePacking row, one Arithmetic SUB and ~63 Memory operations each."""
    loop = 33
    body = (_push32(3) + _push32(200) + bytes([0x21]) + _push32(_ADDR_WORD) + bytes([0xdf, 0x50]) + _push32(1) +
            bytes([0x90, 0x03, 0x80]) + _push32(loop) + bytes([0x57]))
    code = _push32(n_iterations) + bytes([0x5b]) + body + bytes([0x50])
    assert len(code) == 207 and code[200:203] == code[200:203]
    return code, len(code)


def single_block_sponge_effects(data: bytes, timestamp: int):
    """For an input shorter than one rate block: the Keccak-f input the KeccakSponge row sends to the Keccak table
    (the padded block XORed into the zero state) and its five 256-bit Logic XORs (rate chunk = 0, block chunk)."""
    assert len(data) < 135
    blk = bytearray(136)
    blk[:len(data)] = data
    blk[len(data)] |= 1
    blk[135] |= 0x80
    words = [int.from_bytes(blk[8 * i:8 * i + 8], "little") for i in range(17)] + [0] * 8
    xors = [(2, 0, int.from_bytes(bytes(blk[32 * i:32 * i + 32]).ljust(32, b"\0"), "little")) for i in range(5)]
    return (words, timestamp), xors


# A fifth, tiny one for the last looker shape: MLOAD_32BYTES packs 32 bytes of the kernel image (BytePacking `pack`).
CPU_PROGRAM_5 = _push32(32) + _push32(5) + bytes([0xf8, 0x50])
CPU_PROGRAM_5_CONSTS = (68, 0, 777777, 888888)


def program_logs(rng, oracle, kernel_hash=0, program=None, halt_pc=None, n_rows=16, extra_memory=None, **run_kw):
    """The operation logs of the run (what the reference's interpreter would hand to `generate_traces`): Cpu rows,
    Memory operations (public-value writes + the Cpu's bus traffic), mem_before values, and the Arithmetic / Logic /
    KeccakSponge / Keccak / BytePacking operation lists."""
    from tests.test_oracle_tracegen import _keccak_f
    pv = make_public_values(rng)
    code = CPU_PROGRAM if program is None else program
    run = cpu_program_trace(oracle.keccak256, n=n_rows, program=code, halt_pc=halt_pc, extra_memory=extra_memory,
                            return_run=True, **run_kw)
    cpu, cpu_mem_ops, arith_ops, logic_ops, sponge_ops, packing_ops = run.t, run.mem_ops, run.arith, run.logic, run.sponge, run.packing
    sponge = otg.keccak_sponge_generate_trace(sponge_ops, 0, _keccak_f(oracle))
    perms, sponge_xors = sponge_side_effects(sponge)
    before = [((0, SEG_CODE, i), b) for i, b in enumerate(code)]
    before += [((0, SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    before += sorted((extra_memory or {}).items())                  # not part of the kernel image: is_initial would fail
    mem_ops = [dict(filter=True, timestamp=2, ctx=0, seg=seg, virt=idx, is_read=False, value=val)
               for seg, idx, val in oseg.public_memory_writes(pv, kernel_hash, len(code))] + cpu_mem_ops
    return dict(pv=pv, code=code, cpu=cpu, memory=mem_ops, before=before, arithmetic=arith_ops, stale=list(run.stale_contexts),
                logic=logic_ops + sponge_xors, sponge=sponge_ops, sponge_trace=sponge, keccak=perms, packing=packing_ops)


def build_with_cpu_program(rng, oracle, kernel_hash=0, program=None, halt_pc=None, n_rows=16, extra_memory=None, **run_kw):
    """Like `build`, but the kernel image IS CPU_PROGRAM and the Cpu table executes it: code reads, stack writes /
    reads and the hashed bytes join the Memory table, two ADD rows the Arithmetic table, the XOR and the sponge's
    block XORs the Logic table, one KECCAK_GENERAL the KeccakSponge table and its permutation the Keccak table, one
    MSTORE_32BYTES the BytePacking table (whose 32 byte writes land in Memory and MemAfter): all nine tables live."""
    g = program_logs(rng, oracle, kernel_hash, program, halt_pc, n_rows, extra_memory, **run_kw)
    memory, mem_after = mem_trace.generate_trace(g["memory"], g["before"], g["stale"])
    before_rows = [[1, c, s, v] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for (c, s, v), val in g["before"]]
    traces = [None] * 9
    traces[0] = arith_trace.generate_trace(g["arithmetic"])[0]
    traces[1] = otg.byte_packing_generate_trace(g["packing"], 0)
    traces[2] = g["cpu"]
    traces[3] = np.ascontiguousarray(keccak_trace.generate_trace_rows(g["keccak"], 32).T)
    traces[4] = g["sponge_trace"]
    traces[5] = logic_table(g["logic"])
    traces[6] = memory
    traces[7] = continuation_table(before_rows)
    traces[8] = continuation_table(mem_after)
    return traces, g["pv"], g["code"]


# ---- the cdk_erigon feature set: PC PC PC POSEIDON POP, halt at pc = 5 ----------------------------------------------
ERIGON_PROGRAM = bytes([0x58, 0x58, 0x58, 0x22, 0x50, 0x00, 0x00, 0x00])
ERIGON_CONSTS = (5, 0, 777777, 888888)
# POSEIDON_GENERAL over the first 56 bytes of the kernel image: PUSH32 56, PUSH32 (0, Code, 0), POSEIDON_GENERAL, POP
ERIGON_PROGRAM_2 = bytes([0x7f]) + (56).to_bytes(32, "big") + bytes([0x7f]) + (0).to_bytes(32, "big") + bytes([0x23, 0x50])
ERIGON_CONSTS_2 = (68, 0, 777777, 888888)


def build_cdk_erigon_with_cpu_program(rng, oracle, kernel_hash=0, program=None, consts=None, n_rows=16):
    """Ten tables of a `cdk_erigon` run whose 86-column Cpu table executes POSEIDON on three stack words: the Poseidon
    table gets the matching PoseidonSimpleOp row (CTL 10); public values carry a burn address and no eth_mainnet
    fields.  -> (traces[10], pv, code)."""
    from oracle import poseidon_table as pt
    pv = make_public_values(rng)
    pv.update(burn_addr=int.from_bytes(rng.bytes(20), "big"), blob_gas_used=0, excess_blob_gas=0, parent_beacon_root=bytes(32))
    code = ERIGON_PROGRAM if program is None else program
    consts = ERIGON_CONSTS if consts is None else consts
    cpu, cpu_mem_ops, arith_ops, logic_ops, sponge_ops, packing_ops, poseidon_ops = cpu_program_trace(
        oracle.keccak256, n=n_rows, program=code, halt_pc=consts[0], cdk_erigon=True, poseidon_permute=oracle.poseidon_permute)
    assert not (arith_ops or logic_ops or sponge_ops or packing_ops) and len(poseidon_ops) == 1
    before = [((0, SEG_CODE, i), b) for i, b in enumerate(code)]
    before += [((0, SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    ops = [dict(filter=True, timestamp=2, ctx=0, seg=seg, virt=idx, is_read=False, value=val)
           for seg, idx, val in oseg.public_memory_writes(pv, kernel_hash, len(code))] + cpu_mem_ops
    memory, mem_after = mem_trace.generate_trace(ops, before, [])
    before_rows = [[1, c, s, v] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for (c, s, v), val in before]
    traces = [None] * 10
    traces[0] = arith_trace.generate_trace([])[0]
    traces[1] = otg.byte_packing_generate_trace([], 0)
    traces[2] = cpu
    traces[3] = np.ascontiguousarray(keccak_trace.generate_trace_rows([], 32).T)
    traces[4] = otg.keccak_sponge_generate_trace([], 0, None)
    traces[5] = np.zeros((523, 32), dtype=np.uint64)
    traces[6] = memory
    traces[7] = continuation_table(before_rows)
    traces[8] = continuation_table(mem_after)
    traces[9] = pt.generate_trace(poseidon_ops, 16)
    return traces, pv, code


P_FIELD = 0xFFFFFFFF00000001


def ctl_first_values(traces, ctls, challenges):
    """Z(first row) of every CtlZData straight from the rows: sum_r filter(r) / combine(columns(r)), in the
    z-data order `verify_cross_table_lookups` consumes (one value per looking run / looked table, per challenge)."""
    from oracle import stark as S
    per_table = oseg.cross_table_lookup_data(traces, ctls, challenges, 3)
    out = []
    for t, zds in enumerate(per_table):
        tr = traces[t]
        n = tr.shape[1]
        vals = []
        cache = {}
        for z in zds:
            tot = 0
            for ei, (cols, filt) in enumerate(z.columns_filters):
                key = id(filt)
                if key not in cache:                       # rows this entry's filter selects (same for both challenges)
                    involved = sorted({c for a, b in filt.products for col in (a, b) for c, _ in
                                       col.linear_combination + col.next_row_linear_combination} |
                                      {c for col in filt.constants for c, _ in
                                       col.linear_combination + col.next_row_linear_combination})
                    const_only = not involved
                    if const_only:
                        rows = range(n)
                    else:
                        live = np.zeros(n, dtype=bool)
                        for c in involved:
                            live |= tr[c] != 0
                            live |= np.roll(tr[c], -1) != 0
                        rows = np.nonzero(live)[0]
                        # a filter with a constant term can be nonzero where its columns are all zero
                        probe = [0] * tr.shape[0]
                        if filt.eval_filter(probe, probe) != 0:
                            rows = range(n)
                    cache[key] = rows
                for r in cache[key]:
                    r = int(r)
                    f = filt.eval_table(tr, r)
                    if f:
                        terms = [c.eval_table(tr, r) for c in cols]
                        tot = (tot + f * S.inv(z.challenge.combine(terms))) % S.P
            vals.append(tot)
        out.append(vals)
    return out
