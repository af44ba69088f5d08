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


def cpu_program_trace(keccak256, n=16, program=None, halt_pc=None, cdk_erigon=False, poseidon_permute=None):
    """The Cpu rows of the kernel-mode run of CPU_PROGRAM (cpu/columns/mod.rs:56-97 layout), with the memory-bus,
    Arithmetic, Logic, KeccakSponge and BytePacking operations it performs.  Stack discipline as the reference's
    witness generator keeps it: the top of the stack lives in mem_channels[0].value; a push writes the old top through
    the partial channel (stack.rs:173-282); ADD / XOR / KECCAK_GENERAL / MSTORE_32BYTES read their second operand
    through GP channel 1; KECCAK_GENERAL(addr, len) pushes keccak256(mem[addr .. addr + len]) (the word read
    big-endian); MSTORE_32BYTES_32(addr, value) writes the 32 big-endian bytes of value at addr through the
    BytePacking table and pushes addr + 32 (byte_unpacking.rs); a kernel-mode PUSH is not tied to the code bytes
    (its BytePacking looker is filtered by is_not_kernel, cpu_stark.rs:283-301); a POP that leaves a non-empty stack
    makes the NEXT row read the new top through channel 0 (stack.rs:371-410); timestamps = (clock - 1) * 5 + 1 +
    channel."""
    from oracle import airs
    program = CPU_PROGRAM if program is None else program
    halt_pc = CPU_HALT_PC if halt_pc is None else halt_pc
    ops = airs.C_OPS_ERIGON if cdk_erigon else airs.C_OPS
    x = 1 if cdk_erigon else 0                                        # `poseidon` flag column: later columns move by one
    col = lambda name: 6 + ops.index(name)
    bits, gen, clock, partial = 24 + x, 32 + x, 40 + x, 80 + x
    ch = lambda k: 41 + x + 13 * k
    limbs = lambda v: [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    t = np.zeros((85 + x, n), dtype=np.uint64)
    stack, gas, pc, top_read = [], 0, 0, False
    mem_ops, arith, logic, sponge, packing, poseidon = [], [], [], [], [], []
    CPU_PROGRAM_ = program
    for r in range(n):
        t[clock, r], t[4, r], t[3, r], t[5, r], t[2, r] = r + 1, 1, len(stack), gas, pc
        base = r * 5 + 1
        if pc == halt_pc:
            continue                                                  # halting rows
        op = CPU_PROGRAM_[pc]
        for i in range(8):
            t[bits + i, r] = (op >> i) & 1
        mem_ops.append(dict(filter=True, timestamp=base, ctx=0, seg=0, virt=pc, is_read=True, value=op))   # code read
        sl, top = len(stack), (stack[-1] if stack else 0)
        t[ch(0) + 5:ch(0) + 13, r] = limbs(top)
        if top_read:
            t[ch(0):ch(0) + 5, r] = [1, 1, 0, 1, sl - 1]
            mem_ops.append(dict(filter=True, timestamp=base + 1, ctx=0, seg=1, virt=sl - 1, is_read=True, value=top))
            top_read = False
        next_pc = pc + 1
        if op in (0x58, 0x7f):                                        # PC / PUSH32: push
            t[col("pc_push0" if op == 0x58 else "push_prover_input"), r] = 1
            if sl:
                t[gen + 4, r], t[gen + 5, r] = pow(sl, P_FIELD - 2, P_FIELD), 1      # stack_inv, stack_inv_aux
                t[partial:partial + 5, r] = [1, 0, 0, 1, sl - 1]
                mem_ops.append(dict(filter=True, timestamp=base + 4, ctx=0, seg=1, virt=sl - 1, is_read=False, value=top))
            if op == 0x58:
                stack.append(pc)
                gas += 2
            else:
                stack.append(int.from_bytes(CPU_PROGRAM_[pc + 1:pc + 33], "big"))
                gas += 3
                next_pc = pc + 33
        elif op in (0x01, 0x18, 0x21, 0xdf):      # ADD / XOR / KECCAK_GENERAL / MSTORE_32BYTES_32: two operands
            t[col({0x01: "binary_op", 0x18: "logic_op", 0x21: "jumpdest_keccak_general", 0xdf: "m_op_32bytes"}[op]), r] = 1
            a, b = stack[-1], stack[-2]
            t[ch(1):ch(1) + 5, r] = [1, 1, 0, 1, sl - 2]
            t[ch(1) + 5:ch(1) + 13, r] = limbs(b)
            mem_ops.append(dict(filter=True, timestamp=base + 2, ctx=0, seg=1, virt=sl - 2, is_read=True, value=b))
            if op == 0x01:
                arith.append(("bin", 0, a, b))                        # IS_ADD
                stack[-2:] = [(a + b) % (1 << 256)]
                gas += 3
            elif op == 0x18:
                logic.append((2, a, b))                               # is_xor
                stack[-2:] = [a ^ b]
                gas += 3
            else:                                                     # a = address word
                virt, seg, ctx = a & 0xFFFFFFFF, (a >> 32) & 0xFFFFFFFF, (a >> 64) & 0xFFFFFFFF
                if op == 0x21:
                    assert (ctx, seg) == (0, 0) and virt + b <= len(CPU_PROGRAM_), "this run hashes a slice of the kernel image"
                    data = CPU_PROGRAM_[virt:virt + b]
                    sponge.append(((ctx, seg, virt), base, data))
                    mem_ops += [dict(filter=True, timestamp=base, ctx=ctx, seg=seg, virt=virt + i, is_read=True, value=x)
                                for i, x in enumerate(data)]
                    stack[-2:] = [int.from_bytes(keccak256(data), "big")]
                else:
                    data = b.to_bytes(32, "big")
                    packing.append((False, (ctx, seg, virt), base, data))
                    mem_ops += [dict(filter=True, timestamp=base, ctx=ctx, seg=seg, virt=virt + i, is_read=False, value=x)
                                for i, x in enumerate(data)]
                    stack[-2:] = [a + 32]
        elif op == 0x22:                                              # POSEIDON (cdk_erigon): hash three stack words
            t[col("poseidon"), r] = 1
            words = [stack[-1], stack[-2], stack[-3]]
            for k in (1, 2):
                t[ch(k):ch(k) + 5, r] = [1, 1, 0, 1, sl - 1 - k]
                t[ch(k) + 5:ch(k) + 13, r] = limbs(words[k])
                mem_ops.append(dict(filter=True, timestamp=base + 1 + k, ctx=0, seg=1, virt=sl - 1 - k, is_read=True, value=words[k]))
            inp = [((w >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) % P_FIELD for w in words for i in range(4)]
            poseidon.append(("simple", inp))
            out = [int(v) for v in poseidon_permute(inp)[:4]]
            stack[-3:] = [sum(v << (64 * i) for i, v in enumerate(out))]
        else:                                                         # POP
            assert op == 0x50
            t[col("not_pop"), r] = 1
            if sl - 1:
                t[gen + 4, r], t[gen + 5, r], t[gen + 6, r] = pow(sl - 1, P_FIELD - 2, P_FIELD), 1, 1
                top_read = True
            stack.pop()
            gas += 2
        pc = next_pc
    assert pc == halt_pc and not stack
    if cdk_erigon:
        return t, mem_ops, arith, logic, sponge, packing, poseidon
    return t, mem_ops, arith, logic, sponge, packing


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


def program_logs(rng, oracle, kernel_hash=0):
    """The operation logs of the run (what the reference's interpreter would hand to `generate_traces`): Cpu rows,
    Memory operations (public-value writes + the Cpu's bus traffic), mem_before values, and the Arithmetic / Logic /
    KeccakSponge / Keccak / BytePacking operation lists."""
    from tests.test_oracle_tracegen import _keccak_f
    pv = make_public_values(rng)
    code = CPU_PROGRAM
    cpu, cpu_mem_ops, arith_ops, logic_ops, sponge_ops, packing_ops = cpu_program_trace(oracle.keccak256)
    sponge = otg.keccak_sponge_generate_trace(sponge_ops, 0, _keccak_f(oracle))
    perms, sponge_xors = sponge_side_effects(sponge)
    before = [((0, SEG_CODE, i), b) for i, b in enumerate(code)]
    before += [((0, SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    mem_ops = [dict(filter=True, timestamp=2, ctx=0, seg=seg, virt=idx, is_read=False, value=val)
               for seg, idx, val in oseg.public_memory_writes(pv, kernel_hash, len(code))] + cpu_mem_ops
    return dict(pv=pv, code=code, cpu=cpu, memory=mem_ops, before=before, arithmetic=arith_ops,
                logic=logic_ops + sponge_xors, sponge=sponge_ops, sponge_trace=sponge, keccak=perms, packing=packing_ops)


def build_with_cpu_program(rng, oracle, kernel_hash=0):
    """Like `build`, but the kernel image IS CPU_PROGRAM and the Cpu table executes it: code reads, stack writes /
    reads and the hashed bytes join the Memory table, two ADD rows the Arithmetic table, the XOR and the sponge's
    block XORs the Logic table, one KECCAK_GENERAL the KeccakSponge table and its permutation the Keccak table, one
    MSTORE_32BYTES the BytePacking table (whose 32 byte writes land in Memory and MemAfter): all nine tables live."""
    g = program_logs(rng, oracle, kernel_hash)
    memory, mem_after = mem_trace.generate_trace(g["memory"], g["before"], [])
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


def build_cdk_erigon_with_cpu_program(rng, oracle, kernel_hash=0):
    """Ten tables of a `cdk_erigon` run whose 86-column Cpu table executes POSEIDON on three stack words: the Poseidon
    table gets the matching PoseidonSimpleOp row (CTL 10); public values carry a burn address and no eth_mainnet
    fields.  -> (traces[10], pv, code)."""
    from oracle import poseidon_table as pt
    pv = make_public_values(rng)
    pv.update(burn_addr=int.from_bytes(rng.bytes(20), "big"), blob_gas_used=0, excess_blob_gas=0, parent_beacon_root=bytes(32))
    code = ERIGON_PROGRAM
    cpu, cpu_mem_ops, arith_ops, logic_ops, sponge_ops, packing_ops, poseidon_ops = cpu_program_trace(
        oracle.keccak256, program=code, halt_pc=ERIGON_CONSTS[0], cdk_erigon=True, poseidon_permute=oracle.poseidon_permute)
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
