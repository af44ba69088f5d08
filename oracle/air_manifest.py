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

CPU_TOTAL = 514               # cpu_stark.rs:594-626, 18 modules (eth_mainnet); 531 with the cdk_erigon column set

TABLES = {1: MEM_CONTINUATION, 2: LOGIC, 3: MEMORY, 4: BYTE_PACKING, 5: ARITHMETIC, 6: KECCAK, 7: KECCAK_SPONGE}


def expand(runs):
    return [KINDS[k] for k, n in runs for _ in range(n)]
