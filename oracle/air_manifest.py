"""oracle/air_manifest.py -- TEST INFRASTRUCTURE ONLY.

A SECOND, structurally different statement of the tables' constraint lists, written by reading each
`eval_packed_generic` for its *shape* only: which ConstraintConsumer method every yield site calls (plain /
transition / first row / last row), in yield order, with the loop trip counts taken from the reference's column
constants.  oracle/airs.py restates the constraints' CONTENT; this file knows nothing about content and was derived
in a separate pass over the Rust sources (line numbers below), so a constraint dropped, duplicated, reordered across a
kind boundary or yielded through the wrong consumer method in airs.py / airs.cuh shows up as a mismatch in
tests/test_oracle_air_manifest.py.  (airs.cuh is tied to airs.py by the quotient parity tests, which are sensitive to
order and kind: the alpha-Horner accumulation multiplies by z_last / L_first / L_last per kind.)

Notation: a manifest is a list of (kind, count) runs; kinds "c" = constraint, "t" = constraint_transition,
"f" = constraint_first_row, "l" = constraint_last_row.
"""
KINDS = {"c": 0, "t": 1, "f": 2, "l": 3}

# memory_continuation/memory_continuation_stark.rs:110-122
MEM_CONTINUATION = [("c", 1)]

# logic.rs:249-303: three flag booleans, their sum, 2 x 256 input bits, 8 result limbs
LOGIC = [("c", 3), ("c", 1), ("c", 512), ("c", 8)]

# memory/memory_stark.rs:474-626
MEMORY = [
    ("c", 1),      # :511 filter boolean
    ("c", 1),      # :522 dummy rows are reads
    ("c", 4),      # :538-541 first-change flags and address_unchanged are boolean
    ("t", 3),      # :545-550 no change before the column corresponding to the flag
    ("t", 3),      # :551-553 address_unchanged: all three address parts stay
    ("t", 1),      # :561 range_check
    ("t", 3),      # :564-585 preinitialized_segments_aux, preinitialized_segments, initialize_aux
    ("t", 16),     # :586-596 VALUE_LIMBS x (read consistency, zero initialisation)
    ("t", 1),      # :599 maybe_in_mem_after
    ("c", 1),      # :604 mem_after_filter boolean
    ("c", 8),      # :609-613 VALUE_LIMBS x mem_after contents
    ("c", 1),      # :617 timestamp * (timestamp * timestamp_inv - 1)
    ("f", 1),      # :623 range counter starts at 0
    ("t", 1),      # :625 and increments
]

# byte_packing/byte_packing_stark.rs:296-352  (NUM_BYTES = 32)
BYTE_PACKING = [("f", 1), ("t", 1), ("l", 1),          # :314-318 range counter
                ("c", 1), ("f", 1),                     # :325, :328 filter
                ("c", 1),                               # :332 is_read boolean
                ("c", 32),                              # :335-338 index_len booleans
                ("t", 1),                               # :342 next_filter
                ("c", 32 * 31 // 2)]                    # :345-351 index_len[i] * value_bytes[j], i < j

# keccak/round_flags.rs:14-60 (NUM_ROUNDS = 24) then keccak/keccak_stark.rs:266-426
KECCAK = [("c", 24), ("f", 1), ("f", 23), ("t", 24), ("t", 1),      # eval_round_flags
          ("c", 1),                                               # :287 timestamp unchanged within a permutation
          ("c", 5 * 64),                                          # :292-301 C'
          ("c", 25 * 2),                                          # :310-328 A recomposed from A' ^ C ^ C' bits
          ("c", 5 * 64),                                          # :334-345 diff * (diff - 2) * (diff - 4)
          ("c", 25 * 2),                                          # :347-372 A''
          ("c", 2),                                               # :376-388 A''[0, 0] bits
          ("c", 2),                                               # :390-411 A'''[0, 0] = A''[0, 0] ^ RC
          ("t", 25 * 2)]                                          # :414-424 next row's A = this row's A'''

# keccak_sponge/keccak_sponge_stark.rs:546-715
# KECCAK_RATE_BYTES 136, KECCAK_RATE_U32S 34, KECCAK_CAPACITY_U32S 16, KECCAK_DIGEST_U32S 8 (keccak_sponge/columns.rs)
KECCAK_SPONGE = [("f", 1), ("t", 1), ("l", 1),                   # :566-570 range counter
                 ("c", 1),                                       # :575 is_full_input_block boolean
                 ("c", 136),                                     # :577-579 is_padding_byte booleans
                 ("c", 135),                                     # :583-588 padding bytes are a suffix
                 ("c", 1),                                       # :591 final block is not a full block
                 ("f", 1 + 34 + 16),                             # :596-602 first row: nothing absorbed, zero state
                 ("t", 1 + 34 + 16),                             # :606-612 after a final block: the same for the next row
                 ("t", 4),                                       # :616-626 full block: same context / segment / virt / timestamp
                 ("t", 8),                                       # :630-642 digest bytes -> next original_rate_u32s[..8]
                 ("t", 34 - 8),                                  # :643-649 partial_updated_state -> original_rate_u32s[8..]
                 ("t", 16),                                      # :650-657 ... -> original_capacity_u32s
                 ("t", 1),                                       # :661-665 already_absorbed_bytes += 136
                 ("t", 1),                                       # :674-678 single padding byte = 0b10000001
                 ("t", 2 * 135),                                 # :680-698 first padding byte 1, other padding bytes 0
                 ("t", 1),                                       # :700-705 last byte 0b10000000
                 ("t", 1)]                                       # :711-713 dummy rows stay dummy

# arithmetic/arithmetic_stark.rs:203-252 and the operation modules it calls, in call order.  N_LIMBS = 16.
_MODULAR_CONSTR_POLY = 3 + (16 + 1 + 15) + 15      # modular.rs:427-508: mod_is_zero boolean, limb_sum * mod_is_zero, out * mod_is_zero
                                                   # = 3 transitions; `check_reduced` = a two-row addcy (16 limb carries, the
                                                   # carry value, 15 zero limbs; no boolean check, addcy.rs:131-142); the 15 high
                                                   # coefficients prod[2 N_LIMBS..] of the 3 N_LIMBS - 1 long product must vanish
ARITHMETIC = [("c", 17), ("c", 1), ("c", 1),                    # :214-223 op-flag booleans, their sum, opcode only on range checks
              ("f", 1), ("t", 1), ("l", 1),                     # :227-231 range counter
              ("c", 16),                                        # mul.rs:123-173: `pol_mul_lo` keeps the low N_LIMBS coefficients of
                                                                # a(x) b(x) - c(x) - (x - 2^16) s(x)
              ("c", 4 * (16 + 1 + 1 + 15)),                     # addcy.rs:98-172: ADD, SUB, LT, GT: 16 limb carries, the carry-out
                                                                # boolean, its value, 15 zero limbs
              # divmod.rs:86-145, DIV then MOD: not on the last row; modular_constr_poly; 2 N_LIMBS coefficients
              ("l", 1), ("t", _MODULAR_CONSTR_POLY + 32), ("l", 1), ("t", _MODULAR_CONSTR_POLY + 32),
              # modular.rs:539-612: not on the last row; the 16 BN254 modulus limbs; SUBMOD's sign boolean + 16 high quotient
              # limbs (the first is the zeroed sign slot); modular_constr_poly twice (sub, add/mul); three 32-coefficient checks
              ("l", 1), ("t", 16), ("c", 1 + 16), ("t", 2 * _MODULAR_CONSTR_POLY + 3 * 32),
              ("c", 5 + 1 + 8 + 4 + 2 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 15),   # byte.rs:201-296
              ("c", 16),                                        # shift.rs:85-94 SHL = the MUL check on (input0, shifted)
              ("l", 1), ("t", _MODULAR_CONSTR_POLY + 32)]       # shift.rs:100-119 SHR = the DIV check
ARITHMETIC_TOTAL = 707

# ---- CpuStark: cpu/cpu_stark.rs:594-626, the 18 modules in call order (eth_mainnet feature set) -------------------
# NUM_GP_CHANNELS = 3; a memory channel carries 8 value limbs; OpsColumnsView field order (cpu/columns/ops.rs:6-47).
CPU_OPS = ["binary_op", "ternary_op", "fp254_op", "eq_iszero", "logic_op", "not_pop", "shift", "jumpdest_keccak_general",
           "jumps", "push_prover_input", "dup_swap", "context_op", "m_op_32bytes", "exit_kernel", "m_op_general", "pc_push0",
           "syscall", "exception"]


def _stack_one(num_pops, pushes, disable):
    """cpu/stack.rs:190-306 `eval_packed_one`"""
    r = []
    if num_pops > 0:
        r += [("c", 5 * (num_pops - 1)), ("c", 1)]           # the popped channels 1.., the partial channel is unused
        if not pushes:
            r += [("t", 5), ("c", 1), ("t", 1)]              # the new top is read next row unless the stack empties
    elif pushes:
        r += [("c", 5 + 1 + 1)]                              # the old top goes out through the partial channel
    else:
        r += [("c", 1 + 8 + 1)]                              # nothing moves: same top, no channel
    if disable:
        r += [("c", len(range(max(1, num_pops), 3 - int(pushes))))]
    return r + [("t", 1)]                                    # stack_len update


_BINARY, _TERNARY, _UNARY = (2, True, True), (3, True, True), (1, True, True)
# cpu/stack.rs:120-171 STACK_BEHAVIORS and :20-41 MIGHT_OVERFLOW
_STACK_BEHAVIORS = dict(binary_op=_BINARY, ternary_op=_TERNARY, fp254_op=_BINARY, logic_op=_BINARY, shift=(2, True, False),
                        push_prover_input=(0, True, True), pc_push0=(0, True, True), m_op_32bytes=(2, True, False),
                        exit_kernel=(1, False, True), syscall=(0, True, False), exception=(0, True, False))
_MIGHT_OVERFLOW = {"push_prover_input", "pc_push0", "dup_swap", "exit_kernel"}
# cpu/contextops.rs:15-38 KEEPS_CONTEXT: every op except context_op; cpu/gas.rs:20-42 SIMPLE_OPCODES: the `Some` entries
_KEEPS_CONTEXT = [op for op in CPU_OPS if op != "context_op"]
_SIMPLE_GAS = ["fp254_op", "eq_iszero", "logic_op", "shift", "pc_push0", "dup_swap", "context_op", "m_op_32bytes", "m_op_general"]

CPU = (
    [("c", 1 + 2 + 5)]                                       # byte_unpacking.rs:11-46
    + [("f", 1), ("t", 1)]                                   # clock.rs:15-24
    # contextops.rs: keep :40-54, get :79-103, set :150-203, top level :277-314
    + [("t", len(_KEEPS_CONTEXT) + 1)]
    + [("c", 1 + 7 + 1 + 1 + 1 + 1)]
    + [("c", 1 + 6 + 1 + 1 + 1 + 8 + 1 + 1)]
    + [("c", 6)]
    + [("t", 5), ("c", 1), ("t", 3)]                         # control_flow.rs:50-102
    # decode.rs:86-221: kernel flag, 8 opcode bits, 5 OPCODES flags, 11 COMBINED_OPCODES flags, their sum, 5 opcode
    # matches, 12 combined-flag decodings
    + [("c", 1 + 8 + 5 + 11 + 1 + 5 + 12)]
    # dup_swap.rs: dup (two channel equalities + two channel descriptions, stack_len, no top read), swap, partial channel
    + [("c", 8 + 5 + 8 + 5), ("t", 1), ("c", 1)] + [("c", 8 + 5 + 8 + 5 + 1 + 1)] + [("c", 1)]
    + [("t", 1 + len(_SIMPLE_GAS) + 4 + 1 + 1 + 1)]          # gas.rs:44-139: accumulate (+ jumps, binary, ternary, not_pop,
                                                             # jumpdest/keccak, push/prover_input), init
    + [("c", 1), ("t", 1), ("c", 1 + 3), ("l", 1), ("c", 1)]  # halt.rs:16-50
    # jumps.rs: exit_kernel :16-37; jump_jumpi :66-189 (the JumpdestBits read only exists in eth_mainnet / polygon_pos)
    + [("t", 3), ("c", 1)]
    + [("t", 5), ("c", 1), ("t", 1), ("c", 1 + 7 + 4 + 6 + 0 + 1 + 1), ("t", 4)]
    + [("c", 1 + 3 + 1)]                                     # membus.rs:42-57
    # memio.rs: load :27-74 (+ MLOAD_GENERAL stack behaviour), store :135-225
    + [("c", 5 + 8 + 1 + 1)] + _stack_one(1, True, False)
    + [("c", 5 + 1 + 5 + 1 + 1), ("t", 5), ("c", 2)]
    + [("c", 8)]                                             # modfp254.rs
    + [("c", 1 + 7)]                                         # pc.rs
    + [("c", 8)]                                             # push0.rs
    + [("c", 6 + 0)]                                         # shift.rs:19-60 (mem_channels[3..NUM_GP_CHANNELS] is empty)
    + [("c", 8)] + _stack_one(*_UNARY)                       # simple_logic/not.rs
    + [("c", 1 + 7 + 8 + 8 + 1)] + _stack_one(2, True, True) + _stack_one(1, True, True)   # simple_logic/eq_iszero.rs
)
# stack.rs:308-388 `eval_packed`: per op its behaviour and the overflow check, then JUMPDEST / KECCAK_GENERAL, then POP / NOT
for _op in CPU_OPS:
    if _op in _STACK_BEHAVIORS:
        CPU += _stack_one(*_STACK_BEHAVIORS[_op])
    if _op in _MIGHT_OVERFLOW:
        CPU += [("t", 1)]
CPU += _stack_one(0, False, True) + _stack_one(2, True, True)
CPU += [("c", 2), ("t", 5), ("c", 1 + 2 + 1), ("t", 1)]
# syscalls_exceptions.rs:29-135
CPU += [("c", 2 + 1 + 3 + 4 + 2 + 7 + 1), ("t", 3), ("c", 6 + 4)]
CPU_TOTAL = 514               # 531 with the cdk_erigon column set (not itemised)

TABLES = {1: MEM_CONTINUATION, 2: LOGIC, 3: MEMORY, 4: BYTE_PACKING, 5: ARITHMETIC, 6: KECCAK, 7: KECCAK_SPONGE, 8: CPU}


def expand(runs):
    return [KINDS[k] for k, n in runs for _ in range(n)]
