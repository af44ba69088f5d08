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
CPU_PROGRAM = bytes([0x58, 0x58, 0x58, 0x01, 0x18, 0x50])        # PC PC PC ADD XOR POP, then halt at pc = 6
CPU_PROGRAM_CONSTS = (len(CPU_PROGRAM), 0, 777777, 888888)       # halt_final, init, syscall / exception jumptables


def cpu_program_trace(n=16):
    """The Cpu rows of the kernel-mode run of CPU_PROGRAM (cpu/columns/mod.rs:56-97 layout), with the memory-bus
    operations and Arithmetic operations it performs.  Stack discipline as the reference's witness generator keeps
    it: the top of the stack lives in mem_channels[0].value; a push writes the old top through the partial channel
    (stack.rs:173-282), ADD reads its second operand (as XOR) through GP channel 1, a POP that leaves a non-empty stack makes
    the NEXT row read the new top through channel 0 (stack.rs:371-410); timestamps = (clock - 1) * 5 + 1 + channel."""
    from oracle import airs
    ops = airs.C_OPS
    col = lambda name: 6 + ops.index(name)
    bits, gen, clock, partial = 24, 32, 40, 80
    ch = lambda k: 41 + 13 * k
    limbs = lambda v: [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    t = np.zeros((85, n), dtype=np.uint64)
    stack, gas, mem_ops, arith, logic, top_read = [], 0, [], [], [], False
    for r in range(n):
        t[clock, r], t[4, r], t[3, r], t[5, r] = r + 1, 1, len(stack), gas
        base = r * 5 + 1
        if r >= len(CPU_PROGRAM):
            t[2, r] = len(CPU_PROGRAM)                                # halting rows
            continue
        op = CPU_PROGRAM[r]
        t[2, r] = r
        for i in range(8):
            t[bits + i, r] = (op >> i) & 1
        mem_ops.append(dict(filter=True, timestamp=base, ctx=0, seg=0, virt=r, is_read=True, value=op))   # code read
        sl, top = len(stack), (stack[-1] if stack else 0)
        t[ch(0) + 5:ch(0) + 13, r] = limbs(top)
        if top_read:
            t[ch(0):ch(0) + 5, r] = [1, 1, 0, 1, sl - 1]
            mem_ops.append(dict(filter=True, timestamp=base + 1, ctx=0, seg=1, virt=sl - 1, is_read=True, value=top))
            top_read = False
        if op == 0x58:                                                # PC: push the program counter
            t[col("pc_push0"), r] = 1
            if sl:
                t[gen + 4, r], t[gen + 5, r] = pow(sl, P_FIELD - 2, P_FIELD), 1      # stack_inv, stack_inv_aux
                t[partial:partial + 5, r] = [1, 0, 0, 1, sl - 1]
                mem_ops.append(dict(filter=True, timestamp=base + 4, ctx=0, seg=1, virt=sl - 1, is_read=False, value=top))
            stack.append(r)
            gas += 2
        elif op in (0x01, 0x18):                                      # ADD (Arithmetic CTL) / XOR (Logic CTL)
            t[col("binary_op" if op == 0x01 else "logic_op"), r] = 1
            a, b = stack[-1], stack[-2]
            t[ch(1):ch(1) + 5, r] = [1, 1, 0, 1, sl - 2]
            t[ch(1) + 5:ch(1) + 13, r] = limbs(b)
            mem_ops.append(dict(filter=True, timestamp=base + 2, ctx=0, seg=1, virt=sl - 2, is_read=True, value=b))
            if op == 0x01:
                arith.append(("bin", 0, a, b))                        # IS_ADD
                stack[-2:] = [(a + b) % (1 << 256)]
            else:
                logic.append((2, a, b))                               # is_xor
                stack[-2:] = [a ^ b]
            gas += 3
        else:                                                         # POP
            t[col("not_pop"), r] = 1
            if sl - 1:
                t[gen + 4, r], t[gen + 5, r], t[gen + 6, r] = pow(sl - 1, P_FIELD - 2, P_FIELD), 1, 1
                top_read = True
            stack.pop()
            gas += 2
    return t, mem_ops, arith, logic


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


def build_with_cpu_program(rng, kernel_hash=0):
    """Like `build`, but the kernel image IS CPU_PROGRAM and the Cpu table executes it: six code reads, two stack
    writes and two stack reads join the Memory table, one ADD row the Arithmetic table, one XOR row the Logic table."""
    pv = make_public_values(rng)
    code = CPU_PROGRAM
    cpu, cpu_mem_ops, arith_ops, logic_ops = cpu_program_trace()
    before = [((0, SEG_CODE, i), b) for i, b in enumerate(code)]
    before += [((0, SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    ops = [dict(filter=True, timestamp=2, ctx=0, seg=seg, virt=idx, is_read=False, value=val)
           for seg, idx, val in oseg.public_memory_writes(pv, kernel_hash, len(code))] + cpu_mem_ops
    memory, mem_after = mem_trace.generate_trace(ops, before, [])
    before_rows = [[1, c, s, v] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for (c, s, v), val in before]
    traces = [None] * 9
    traces[0] = arith_trace.generate_trace(arith_ops)[0]
    traces[1] = otg.byte_packing_generate_trace([], 0)
    traces[2] = cpu
    traces[3] = np.ascontiguousarray(keccak_trace.generate_trace_rows([], 32).T)
    traces[4] = otg.keccak_sponge_generate_trace([], 0, None)
    traces[5] = logic_table(logic_ops)
    traces[6] = memory
    traces[7] = continuation_table(before_rows)
    traces[8] = continuation_table(mem_after)
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
