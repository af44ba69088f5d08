"""A miniature kernel-mode witness generator for the Cpu table -- TEST INFRASTRUCTURE.

Restates, for a subset of opcodes, what the reference's interpreter writes into a `CpuColumnsView` row and its
operation logs (evm_arithmetization/src/witness/operation.rs `generate_*`, witness/util.rs `stack_pop_with_log_and_fill`
/ `push_no_write` / `push_with_write` / `mem_*_log_and_fill`), so that the restated Cpu AIR (oracle/airs.py) can be held
against real instruction runs and the CTLs against the traffic those runs create.  Conventions:

* the top of the stack lives in registers: a row shows it in mem_channels[0].value (used = 0); operand k >= 1 of an
  instruction is read from (context, Stack, stack_len - 1 - k) through GP channel k;
* an instruction that pushes onto a non-empty stack spills the old top to (context, Stack, stack_len - 1): through the
  partial channel (push_with_write), through GP channel 1 for DUP, through GP channel 2 for GET_CONTEXT;
* the result of an instruction is the next row's mem_channels[0].value (push_no_write);
* an instruction that shrinks the stack without pushing makes the NEXT row read the new top through GP channel 0
  (is_stack_top_read), unless the stack became empty;
* timestamps: clock * NUM_CHANNELS + channel with channels Code = 0, GP(k) = 1 + k, Partial = 4 (clock starts at 0 for
  the row with clock column 1).

Supported: PUSH0 PC PUSH32 (kernel), DUPn SWAPn POP, ADD MUL SUB LT GT (binary), ADDMOD MULMOD SUBMOD, AND OR XOR, NOT ISZERO EQ,
SHL, KECCAK_GENERAL, MLOAD_GENERAL MSTORE_GENERAL, MSTORE_32BYTES_n MLOAD_32BYTES, JUMP JUMPI JUMPDEST, GET_CONTEXT
SET_CONTEXT, EXIT_KERNEL and,
with cdk_erigon, POSEIDON.  User mode (entered with EXIT_KERNEL): PUSHn reads its argument from the code through the
BytePacking table, JUMP / JUMPI read the JUMPDEST bit of the target, a non-native opcode (`syscall_opcodes`) traps into
the kernel through the syscall jump table (3-byte big-endian handler offsets at syscall_jumptable + 3 * opcode, read
through BytePacking; a range-check row in Arithmetic covers the pushed kexit_info), pushing instructions keep
stack_len_bounds_aux = 1 / (next stack_len - 1025)."""
import numpy as np

P = 0xFFFFFFFF00000001
M256 = (1 << 256) - 1
SEG_CODE, SEG_STACK, SEG_SHIFT_TABLE, SEG_JUMPDEST_BITS = 0, 1, 13, 14
ARITH_CODE = {0x01: 0, 0x02: 1, 0x03: 2, 0x10: 11, 0x11: 12, 0x08: 5, 0x09: 6, 0x1b: 14}   # opcode -> IS_* column
LOGIC_KIND = {0x16: 0, 0x17: 1, 0x18: 2}
BN254 = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def limbs(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def finv(x):
    x %= P
    return pow(x, P - 2, P) if x else 0


class KernelRun:
    def __init__(self, code: bytes, halt_pc: int, n_rows: int, keccak256=None, poseidon_permute=None, cdk_erigon=False,
                 memory=None, syscall_jumptable=0, syscall_opcodes=(), exception_jumptable=0, exception_opcodes=None):
        from oracle import airs
        self.code, self.halt_pc, self.n = code, halt_pc, n_rows
        self.keccak256, self.poseidon_permute = keccak256, poseidon_permute
        self.ops = airs.C_OPS_ERIGON if cdk_erigon else airs.C_OPS
        self.x = x = 1 if cdk_erigon else 0
        self.bits, self.gen, self.clock, self.partial = 24 + x, 32 + x, 40 + x, 80 + x
        self.t = np.zeros((85 + x, n_rows), dtype=np.uint64)
        self.stack, self.gas, self.pc, self.top_read, self.context, self.kernel = [], 0, 0, False, 0, 1
        self.syscall_jumptable, self.syscall_opcodes = syscall_jumptable, set(syscall_opcodes)
        self.exception_jumptable, self.exception_opcodes = exception_jumptable, dict(exception_opcodes or {})
        self.mem = dict(memory or {})                     # (ctx, seg, virt) -> value, for MLOAD / MSTORE consistency
        self.mem_ops, self.arith, self.logic, self.sponge, self.packing, self.poseidon = [], [], [], [], [], []
        self.stacks, self.stale_contexts = {}, []         # other contexts' stacks (their tops are in memory)

    def col(self, name): return 6 + self.ops.index(name)
    def ch(self, k): return 41 + self.x + 13 * k

    # ---- memory bus ----
    def _log(self, ts, addr, is_read, value):
        self.mem_ops.append(dict(filter=True, timestamp=ts, ctx=addr[0], seg=addr[1], virt=addr[2], is_read=is_read, value=value))

    def gp(self, r, k, addr, is_read, value):
        """mem_read_gp_with_log_and_fill / mem_write_gp_log_and_fill on GP channel k"""
        c = self.ch(k)
        self.t[c:c + 5, r] = [1, 1 if is_read else 0, addr[0], addr[1], addr[2]]
        self.t[c + 5:c + 13, r] = limbs(value)
        self._log(r * 5 + 1 + 1 + k, addr, is_read, value)
        if not is_read:
            self.mem[addr] = value

    def partial_write(self, r, addr, value):
        """mem_write_partial_log_and_fill: the value column is mem_channels[0].value"""
        p = self.partial
        self.t[p:p + 5, r] = [1, 0, addr[0], addr[1], addr[2]]
        self._log(r * 5 + 1 + 4, addr, False, value)
        self.mem[addr] = value

    def stack_addr(self, depth):
        return (self.context, SEG_STACK, len(self.stack) - 1 - depth)

    def operand(self, r, k):
        """operand k >= 1 through GP channel k (stack_pop_with_log_and_fill)"""
        v = self.stack[-1 - k]
        self.gp(r, k, self.stack_addr(k), True, v)
        return v

    def stack_inv(self, r, diff, aux2=None):
        g = self.gen
        if diff % P:
            self.t[g + 4, r], self.t[g + 5, r] = finv(diff), 1
            if aux2 is not None:
                self.t[g + 6, r] = aux2
            return True
        return False

    def push_with_write(self, r):
        sl = len(self.stack)
        if sl:
            self.stack_inv(r, sl)
            self.partial_write(r, self.stack_addr(0), self.stack[-1])

    # ---- the run ----
    def run(self):
        t = self.t
        self.bounds_row = None
        for r in range(self.n):
            t[self.clock, r], t[4, r], t[3, r], t[5, r], t[2, r], t[0, r] = r + 1, self.kernel, len(self.stack), self.gas, self.pc, self.context
            t[1, r] = (1 - self.kernel) * self.context                # code_context (membus.rs)
            if r and self.bounds_row is not None and not self.kernel:   # stack.rs MIGHT_OVERFLOW: the row before pushed
                t[self.gen + 7, self.bounds_row] = finv(len(self.stack) - 1025)
            self.bounds_row = None
            if self.pc == self.halt_pc and self.kernel:
                continue
            op = self.code[self.pc]
            for i in range(8):
                t[self.bits + i, r] = (op >> i) & 1
            self._log(r * 5 + 1, ((1 - self.kernel) * self.context, SEG_CODE, self.pc), True, op)
            sl, top = len(self.stack), (self.stack[-1] if self.stack else 0)
            t[self.ch(0) + 5:self.ch(0) + 13, r] = limbs(top)
            if self.top_read:
                c = self.ch(0)
                t[c:c + 5, r] = [1, 1, self.context, SEG_STACK, sl - 1]
                self._log(r * 5 + 2, self.stack_addr(0), True, top)
                self.top_read = False
            self.next_pc = self.pc + 1
            self.step(r, op, sl, top)
            self.pc = self.next_pc
        assert self.pc == self.halt_pc, "the program did not reach halt_pc within the table"
        return self

    def flag(self, r, name): self.t[self.col(name), r] = 1

    def shrink(self, r, new_len):
        """after popping without pushing: the next row must fetch the new top unless the stack is now empty"""
        if new_len:
            self.top_read = True

    def step(self, r, op, sl, top):
        t, S = self.t, self.stack
        trap = op in self.syscall_opcodes or op in self.exception_opcodes
        if trap and not self.kernel:                                  # generate_syscall / generate_exception
            exc = op in self.exception_opcodes
            self.flag(r, "exception" if exc else "syscall")
            if exc:
                code_ = self.exception_opcodes[op]
                for i in range(3):
                    t[self.gen + i, r] = (code_ >> i) & 1             # general.exception().exc_code_bits
            table = self.exception_jumptable + 3 * self.exception_opcodes[op] if exc else self.syscall_jumptable + 3 * op
            handler = int.from_bytes(self.code[table:table + 3], "big")
            c = self.ch(1)                                            # describes the packed read; the channel itself is unused
            t[c:c + 6, r] = [0, 1, 0, SEG_CODE, table, handler]
            self.packing.append((True, (0, SEG_CODE, table), r * 5 + 1, self.code[table:table + 3]))
            for i in range(3):
                self._log(r * 5 + 1, (0, SEG_CODE, table + i), True, self.code[table + i])
            info = (self.pc + (0 if exc else 1)) | (self.kernel << 32) | (self.gas << 192)
            self.arith.append(("range_check", top, handler, 0, op, info))
            self.push_with_write(r)
            S.append(info)
            self.next_pc, self.kernel, self.gas = handler, 1, 0
        elif op in (0x58, 0x5f) or 0x60 <= op <= 0x7f:                # PC / PUSH0 / PUSHn
            self.flag(r, "pc_push0" if op in (0x58, 0x5f) else "push_prover_input")
            self.push_with_write(r)
            self.bounds_row = r
            if op == 0x58:
                S.append(self.pc); self.gas += 2
            elif op == 0x5f:
                S.append(0); self.gas += 2
            else:
                n = op - 0x5f
                data = self.code[self.pc + 1:self.pc + 1 + n]
                S.append(int.from_bytes(data, "big")); self.gas += 3
                self.next_pc = self.pc + 1 + n
                if not self.kernel:                                   # user-mode PUSH: argument checked through BytePacking
                    t[self.gen, r] = 1                                # general.push().is_not_kernel
                    addr = (self.context, SEG_CODE, self.pc + 1)
                    self.packing.append((True, addr, r * 5 + 1, data))
                    for i, v in enumerate(data):
                        self._log(r * 5 + 1, (addr[0], addr[1], addr[2] + i), True, v)
        elif op == 0xf9:                                              # EXIT_KERNEL(kexit_info)
            self.flag(r, "exit_kernel")
            if self.stack_inv(r, sl - 1):
                self.top_read = True
            S.pop()
            self.bounds_row = r
            self.next_pc, self.kernel, self.gas = top & 0xFFFFFFFF, (top >> 32) & 1, (top >> 192) & 0xFFFFFFFF
        elif 0x80 <= op <= 0x8f:                                      # DUPn (n = op & 15, zero-based)
            n = op & 0xF
            self.flag(r, "dup_swap")
            self.gp(r, 1, self.stack_addr(0), False, top)
            val = S[-1 - n]
            self.gp(r, 2, self.stack_addr(n), True, val)
            self.bounds_row = r
            S.append(val); self.gas += 3
        elif 0x90 <= op <= 0x9f:                                      # SWAPn
            n = op & 0xF
            self.flag(r, "dup_swap")
            other = self.stack_addr(n + 1)
            in1 = S[-2 - n]
            self.gp(r, 1, other, True, in1)
            self.gp(r, 2, other, False, top)
            S[-2 - n], S[-1] = top, in1
            self.gas += 3
        elif op == 0x50:                                              # POP
            self.flag(r, "not_pop")
            if self.stack_inv(r, sl - 1, aux2=1):
                self.top_read = True
            S.pop(); self.gas += 2
        elif op == 0x19:                                              # NOT
            self.flag(r, "not_pop")
            self.stack_inv(r, sl - 1)
            S[-1] = top ^ M256; self.gas += 3
        elif op in (0x14, 0x15):                                      # EQ / ISZERO
            self.flag(r, "eq_iszero")
            other = self.operand(r, 1) if op == 0x14 else 0
            a, b = limbs(top), limbs(other)
            ne = sum(1 for x, y in zip(a, b) if x != y)
            for i, (x, y) in enumerate(zip(a, b)):
                t[self.gen + i, r] = finv(x - y) * finv(ne) % P
            res = 1 if top == other else 0
            if op == 0x14:
                S[-2:] = [res]
            else:
                S[-1] = res
            self.gas += 3
        elif op in ARITH_CODE or op in LOGIC_KIND:                    # two / three operand ALU operations
            ternary = op in (0x08, 0x09)
            self.flag(r, "logic_op" if op in LOGIC_KIND else "ternary_op" if ternary else "shift" if op == 0x1b else "binary_op")
            b = self.operand(r, 1)
            if op in LOGIC_KIND:
                res = (top & b, top | b, top ^ b)[LOGIC_KIND[op]]
                self.logic.append((LOGIC_KIND[op], top, b))
                S[-2:] = [res]; self.gas += 3
            elif ternary:
                m = self.operand(r, 2)
                res = 0 if m == 0 else ((top + b) % m if op == 0x08 else (top * b) % m)
                self.arith.append(("ter", ARITH_CODE[op], top, b, m))
                S[-3:] = [res]; self.gas += 8
            elif op == 0x1b:                                          # SHL: shift = top, value = b; 2^shift from the table
                assert top < 256, "large shifts not modelled"
                self.gp(r, 2, (0, SEG_SHIFT_TABLE, top), True, 1 << top)
                self.arith.append(("bin", 14, top, b))
                S[-2:] = [(b << top) & M256]; self.gas += 3
            else:
                res = {0x01: (top + b) & M256, 0x02: (top * b) & M256, 0x03: (top - b) & M256,
                       0x10: int(top < b), 0x11: int(top > b)}[op]
                self.arith.append(("bin", ARITH_CODE[op], top, b))
                S[-2:] = [res]; self.gas += 5 if op == 0x02 else 3
        elif op in (0x0c, 0x0d, 0x0e):                                # ADDFP254 / MULFP254 / SUBFP254 (kernel only)
            self.flag(r, "fp254_op")
            b = self.operand(r, 1)
            t[self.ch(2) + 5:self.ch(2) + 13, r] = limbs(BN254)       # modfp254.rs: channel 2 shows the modulus (no memory op)
            res = ((top + b) % BN254, (top * b) % BN254, (top - b) % BN254)[op - 0x0c]
            self.arith.append(("bin", 7 + op - 0x0c, top, b))
            S[-2:] = [res]
        elif op == 0x21:                                              # KECCAK_GENERAL(addr, len)
            self.flag(r, "jumpdest_keccak_general")
            ln = self.operand(r, 1)
            addr = (top >> 64 & 0xFFFFFFFF, top >> 32 & 0xFFFFFFFF, top & 0xFFFFFFFF)
            data = bytes(self.mem.get((addr[0], addr[1], addr[2] + i), 0) for i in range(ln))
            self.sponge.append((addr, r * 5 + 1, data))
            for i, v in enumerate(data):
                self._log(r * 5 + 1, (addr[0], addr[1], addr[2] + i), True, v)
            S[-2:] = [int.from_bytes(self.keccak256(data), "big")]
        elif op == 0x5b:                                              # JUMPDEST
            self.flag(r, "jumpdest_keccak_general")
            self.gas += 1
        elif op in (0x56, 0x57):                                      # JUMP / JUMPI (kernel mode: no JUMPDEST-bit read)
            self.flag(r, "jumps")
            jumpi = op == 0x57
            cond = self.operand(r, 1) if jumpi else 1
            if not jumpi:
                t[self.ch(1) + 5, r] = 1                              # JUMP: the condition register shows 1, unused channel
            sj = 1 if cond else 0
            t[self.gen, r] = sj                                       # should_jump
            t[self.gen + 1, r] = finv(sum(limbs(cond)))               # cond_sum_pinv
            c2 = self.ch(2)                                           # JUMPDEST-bit channel: a real read only in user mode
            used = sj * (1 - self.kernel)
            t[c2:c2 + 6, r] = [used, 1, self.context, SEG_JUMPDEST_BITS, top & 0xFFFFFFFF, 1]
            if used:
                self._log(r * 5 + 4, (self.context, SEG_JUMPDEST_BITS, top & 0xFFFFFFFF), True, 1)
            new_len = sl - (2 if jumpi else 1)
            if self.stack_inv(r, new_len):
                self.top_read = True
            del S[-(2 if jumpi else 1):]
            if sj:
                self.next_pc = top
            self.gas += 10 if jumpi else 8
        elif op in (0xfb, 0xfc):                                      # MLOAD_GENERAL / MSTORE_GENERAL
            self.flag(r, "m_op_general")
            if op == 0xfb:
                addr = (top >> 64 & 0xFFFFFFFF, top >> 32 & 0xFFFFFFFF, top & 0xFFFFFFFF)
                val = self.mem.get(addr, 0)
                self.gp(r, 1, addr, True, val)
                self.stack_inv(r, sl - 2)
                S[-1] = val
            else:
                a = self.operand(r, 1)
                addr = (a >> 64 & 0xFFFFFFFF, a >> 32 & 0xFFFFFFFF, a & 0xFFFFFFFF)
                self.partial_write(r, addr, top)
                if self.stack_inv(r, sl - 2, aux2=1):
                    self.top_read = True
                del S[-2:]
        elif op == 0xf8:                                              # MLOAD_32BYTES(addr, len)
            self.flag(r, "m_op_32bytes")
            ln = self.operand(r, 1)
            addr = (top >> 64 & 0xFFFFFFFF, top >> 32 & 0xFFFFFFFF, top & 0xFFFFFFFF)
            data = bytes(self.mem.get((addr[0], addr[1], addr[2] + i), 0) for i in range(ln))
            self.packing.append((True, addr, r * 5 + 1, data))
            for i, v in enumerate(data):
                self._log(r * 5 + 1, (addr[0], addr[1], addr[2] + i), True, v)
            S[-2:] = [int.from_bytes(data, "big")]
        elif 0xc0 <= op <= 0xdf:                                      # MSTORE_32BYTES_n(addr, value)
            n = (op & 0x1F) + 1
            self.flag(r, "m_op_32bytes")
            val = self.operand(r, 1)
            addr = (top >> 64 & 0xFFFFFFFF, top >> 32 & 0xFFFFFFFF, top & 0xFFFFFFFF)
            data = (val & ((1 << (8 * n)) - 1)).to_bytes(n, "big")
            self.packing.append((False, addr, r * 5 + 1, data))
            for i, v in enumerate(data):
                self._log(r * 5 + 1, (addr[0], addr[1], addr[2] + i), False, v)
                self.mem[(addr[0], addr[1], addr[2] + i)] = v
            S[-2:] = [top + n]
        elif op == 0xf7:                                              # SET_CONTEXT(new_ctx << 64 | prune_flag)
            self.flag(r, "context_op")
            new_ctx, prune = (top >> 64) & 0xFFFFFFFF, top & 1
            assert new_ctx != self.context, "same-context switch not modelled"
            S.pop()
            # the two stack-pointer operations have no channel of their own: their tuples are built by the CTL from
            # registers (cpu_stark.rs:228-281), with the timestamps of GP channels 1 and 2
            self._log(r * 5 + 3, (self.context, 6, 11), False, len(S))              # ContextMetadata::StackSize
            self.mem[(self.context, 6, 11)] = len(S)
            new_sp = self.mem.get((new_ctx, 6, 11), 0)
            self._log(r * 5 + 4, (new_ctx, 6, 11), True, new_sp)
            self.stacks[self.context] = list(S)
            S[:] = self.stacks.pop(new_ctx, [])
            assert len(S) == new_sp
            if prune:
                t[self.gen, r] = 1                                                  # context_pruning().pruning_flag
                self.stale_contexts.append(self.context)
            old_ctx, self.context = self.context, new_ctx
            if new_sp:
                t[self.gen + 4, r], t[self.gen + 5, r], t[self.gen + 6, r] = finv(new_sp), 1, 1
                self.gp(r, 2, (new_ctx, SEG_STACK, new_sp - 1), True, S[-1])
        elif op == 0xf6:                                              # GET_CONTEXT
            self.flag(r, "context_op")
            if sl:
                self.stack_inv(r, sl)
                self.gp(r, 2, self.stack_addr(0), False, top)
            S.append(self.context << 64)
        elif op == 0x23 and self.x:                                   # POSEIDON_GENERAL(addr, len) (cdk_erigon)
            self.flag(r, "poseidon")
            ln = self.operand(r, 1)
            assert ln and ln % 56 == 0, "the table's generator handles whole 56-byte blocks (see its TODO)"
            addr = (top >> 64 & 0xFFFFFFFF, top >> 32 & 0xFFFFFFFF, top & 0xFFFFFFFF)
            data = bytes(self.mem.get((addr[0], addr[1], addr[2] + i), 0) for i in range(ln))
            ts = (r + 1) * 5                                          # cpu_stark.rs:503: clock * NUM_CHANNELS
            self.poseidon.append(("general", addr, ts, data, ln))
            for i, v in enumerate(data):
                self._log(ts, (addr[0], addr[1], addr[2] + i), True, v)
            cap = [0, 0, 0, 0]
            for off in range(0, ln, 56):                              # poseidon_hash_padded_byte_vec, smt_trie/src/code.rs:16-35
                st = [int.from_bytes(data[off + 7 * i:off + 7 * i + 7], "little") for i in range(8)] + cap
                cap = [int(v) for v in self.poseidon_permute(st)[:4]]
            S[-2:] = [sum(v << (64 * i) for i, v in enumerate(cap))]
        elif op == 0x22 and self.x:                                   # POSEIDON (cdk_erigon)
            self.flag(r, "poseidon")
            words = [top, self.operand(r, 1), self.operand(r, 2)]
            inp = [((w >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) % P for w in words for i in range(4)]
            self.poseidon.append(("simple", inp))
            out = [int(v) for v in self.poseidon_permute(inp)[:4]]
            S[-3:] = [sum(v << (64 * i) for i, v in enumerate(out))]
        else:
            raise ValueError("opcode %#x is not modelled" % op)
