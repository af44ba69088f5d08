"""oracle/poseidon_table.py -- TEST INFRASTRUCTURE ONLY.

The `cdk_erigon` Poseidon table: columns evm_arithmetization/src/poseidon/columns.rs:14-94,
constraints poseidon_stark.rs:445-690 (`eval_packed_generic`), witness generator :183-405 (`generate_trace_rows`,
`generate_row_for_simple_op`, `generate_rows_for_general_op`, `generate_perm`), CTL descriptors :35-137.

The reference evaluates the 22 partial rounds through plonky2's *fast* sparse factorisation
(`mds_partial_layer_init/_fast`, constants not present in this tree: [EXT] poseidon_goldilocks.rs).  This restatement
uses the *plain* round function (S-box on word 0, full MDS, full constant vector) instead.  That is the same
constraint *polynomial*, not just the same zero set: between two S-boxes both forms are affine maps of (state after the
first full rounds, the S-box outputs so far); the fast form is an algebraic rewriting of the plain one that holds for
an arbitrary function on word 0, hence also with the S-box outputs as free symbols, so the affine maps -- and with
them every `state[0] - partial_sbox[r]` constraint and the state entering the second full rounds -- coincide.  The
S-box input columns are therefore also what the plain permutation sees, which is how the generator fills them."""
import os
import re

import numpy as np

P = 0xFFFFFFFF00000001
WIDTH, RATE, DIGEST = 12, 8, 4
HALF_FULL, N_PARTIAL = 4, 22
FELT_MAX_BYTES = 7
BLOCK_BYTES = FELT_MAX_BYTES * RATE                       # 56
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11

# ---- columns (columns.rs:14-94) ----
CONTEXT, SEGMENT, VIRT, TIMESTAMP, LEN, ALREADY_ABSORBED = range(6)
IS_FINAL_INPUT_LEN = 6                                    # [8]
IS_FULL_INPUT_BLOCK = 14
INPUT = 15                                                # [12]
CUBED_FULL = 27                                           # [96]
CUBED_PARTIAL = 123                                       # [22]
FULL_SBOX_0 = 145                                         # [36]
PARTIAL_SBOX = 181                                        # [22]
FULL_SBOX_1 = 203                                         # [48]
DIGEST_COL = 251                                          # [8]
OUTPUT_PARTIAL = 259                                      # [8]
PINV = 267                                                # [4]
INPUT_BYTES = 271                                         # [8][6]
IS_SIMPLE_OP, IS_FIRST_ROW_GENERAL_OP, NOT_PADDING = 319, 320, 321
NUM_COLUMNS = 322


def _round_constants():
    """The 360 constants from the generated header the device code is built from (tools/gen_poseidon_constants.py:
    ChaCha8Rng::seed_from_u64(0)); the C oracle carries its own copy and the KAT tests pin both."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "poseidon_constants.h")
    text = open(path).read()
    body = text[text.index("ZK_POSEIDON_RC_INIT"):text.index("ZK_POSEIDON_RCS_INIT")]
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{16})ULL", body)]
    assert len(vals) == 360 and vals[0] == 0xb585f766f2144405
    return vals


RC = _round_constants()


def mds(state):
    """`mds_layer`: out[r] = sum_i circ[i] * s[(i + r) % 12] + diag[r] * s[r]; works on ints and on Ext."""
    out = []
    for r in range(WIDTH):
        acc = state[r] * MDS_DIAG[r]
        for i in range(WIDTH):
            acc = acc + state[(i + r) % WIDTH] * MDS_CIRC[i]
        out.append(acc)
    return out


def eval_poseidon(lv, nv, c):
    """poseidon_stark.rs:445-690, plain-round form (see the module docstring)."""
    is_full = lv[IS_FULL_INPUT_BLOCK]
    c.constraint(is_full * (is_full - 1))
    finals = [lv[IS_FINAL_INPUT_LEN + i] for i in range(RATE)]
    is_final = sum(finals[1:], finals[0])
    c.constraint(is_final * (is_final - 1))
    for f in finals:
        c.constraint(f * (f - 1))
    first_general = lv[IS_FIRST_ROW_GENERAL_OP]
    c.constraint(first_general * (first_general - 1))
    c.constraint(is_final * is_full)
    absorbed = lv[ALREADY_ABSORBED]
    c.constraint_first_row(absorbed)
    for i in range(RATE, WIDTH):
        c.constraint_first_row(lv[LEN] * lv[INPUT + i])
    c.constraint_transition(is_final * nv[ALREADY_ABSORBED])
    for i in range(RATE, WIDTH):
        c.constraint_transition(nv[LEN] * is_final * nv[INPUT + i])
    for col in (CONTEXT, SEGMENT, VIRT, TIMESTAMP):
        c.constraint_transition(is_full * (lv[col] - nv[col]))
    c.constraint_transition(is_full * (absorbed + BLOCK_BYTES - nv[ALREADY_ABSORBED]))
    for i in range(WIDTH - RATE):
        c.constraint_transition(is_full * (lv[DIGEST_COL + 2 * i] + lv[DIGEST_COL + 2 * i + 1] * (1 << 32)
                                           - nv[INPUT + RATE + i]))
    is_dummy = 1 - is_full - is_final
    nfinals = [nv[IS_FINAL_INPUT_LEN + i] for i in range(RATE)]
    next_is_final = sum(nfinals[1:], nfinals[0])
    c.constraint_transition(is_dummy * (nv[IS_FULL_INPUT_BLOCK] + next_is_final))
    offset = lv[LEN] - absorbed
    for i, f in enumerate(finals):
        c.constraint(lv[LEN] * f * (offset - (BLOCK_BYTES - i)))
    # ---- the permutation ----
    state = [lv[INPUT + i] for i in range(WIDTH)]
    rnd = 0
    for r in range(HALF_FULL):
        state = [state[i] + RC[rnd * WIDTH + i] for i in range(WIDTH)]
        for i in range(WIDTH):
            if r != 0:
                sbox_in = lv[FULL_SBOX_0 + WIDTH * (r - 1) + i]
                c.constraint(state[i] - sbox_in)
                state[i] = sbox_in
            cube = lv[CUBED_FULL + WIDTH * r + i]
            c.constraint(state[i] * state[i] * state[i] - cube)
            state[i] = state[i] * (cube * cube)
        state = mds(state)
        rnd += 1
    for r in range(N_PARTIAL):
        state = [state[i] + RC[rnd * WIDTH + i] for i in range(WIDTH)]
        sbox_in = lv[PARTIAL_SBOX + r]
        c.constraint(state[0] - sbox_in)
        cube = lv[CUBED_PARTIAL + r]
        c.constraint(sbox_in * sbox_in * sbox_in - cube)
        state[0] = cube * cube * sbox_in
        state = mds(state)
        rnd += 1
    for r in range(HALF_FULL):
        state = [state[i] + RC[rnd * WIDTH + i] for i in range(WIDTH)]
        for i in range(WIDTH):
            sbox_in = lv[FULL_SBOX_1 + WIDTH * r + i]
            c.constraint(state[i] - sbox_in)
            cube = lv[CUBED_FULL + WIDTH * (HALF_FULL + r) + i]
            c.constraint(sbox_in * sbox_in * sbox_in - cube)
            state[i] = sbox_in * (cube * cube)
        state = mds(state)
        rnd += 1
    for i in range(DIGEST):
        c.constraint(state[i] - (lv[DIGEST_COL + 2 * i] + lv[DIGEST_COL + 2 * i + 1] * (1 << 32)))
    for i in range(DIGEST, WIDTH):
        c.constraint(state[i] - lv[OUTPUT_PARTIAL + i - DIGEST])
    for i in range(DIGEST):
        c.constraint(((lv[DIGEST_COL + 2 * i + 1] - 0xFFFFFFFF) * lv[PINV + i] - 1) * lv[DIGEST_COL + 2 * i])


# ---- witness generation (poseidon_stark.rs:183-405) ----
def generate_perm(row, inp):
    """`generate_perm`: fills input, the S-box input / cube columns, digest limbs, output_partial, pinv."""
    state = [int(x) % P for x in inp]
    row[INPUT:INPUT + WIDTH] = state
    rnd = 0
    for r in range(HALF_FULL):
        state = [(s + RC[rnd * WIDTH + i]) % P for i, s in enumerate(state)]
        for i in range(WIDTH):
            if r != 0:
                row[FULL_SBOX_0 + WIDTH * (r - 1) + i] = state[i]
            cube = pow(state[i], 3, P)
            row[CUBED_FULL + WIDTH * r + i] = cube
            state[i] = state[i] * cube * cube % P
        state = [x % P for x in mds(state)]
        rnd += 1
    for r in range(N_PARTIAL):
        state = [(s + RC[rnd * WIDTH + i]) % P for i, s in enumerate(state)]
        row[PARTIAL_SBOX + r] = state[0]
        cube = pow(state[0], 3, P)
        row[CUBED_PARTIAL + r] = cube
        state[0] = state[0] * cube * cube % P
        state = [x % P for x in mds(state)]
        rnd += 1
    for r in range(HALF_FULL):
        state = [(s + RC[rnd * WIDTH + i]) % P for i, s in enumerate(state)]
        for i in range(WIDTH):
            row[FULL_SBOX_1 + WIDTH * r + i] = state[i]
            cube = pow(state[i], 3, P)
            row[CUBED_FULL + WIDTH * (HALF_FULL + r) + i] = cube
            state[i] = state[i] * cube * cube % P
        state = [x % P for x in mds(state)]
        rnd += 1
    for i in range(DIGEST):
        lo, hi = state[i] & 0xFFFFFFFF, state[i] >> 32
        d = (hi - 0xFFFFFFFF) % P
        row[PINV + i] = pow(d, P - 2, P) if d else 0
        row[DIGEST_COL + 2 * i], row[DIGEST_COL + 2 * i + 1] = lo, hi
    row[OUTPUT_PARTIAL:OUTPUT_PARTIAL + WIDTH - DIGEST] = state[DIGEST:]
    return state


def _rows_for_general_op(addr, timestamp, data, length):
    assert len(data) % BLOCK_BYTES == 0 and len(data) > 0, "the input is padded to a multiple of 56 bytes"
    blocks = [data[o:o + BLOCK_BYTES] for o in range(0, len(data), BLOCK_BYTES)]
    last_non_padding = length % BLOCK_BYTES
    assert last_non_padding < RATE, "is_final_input_len has 8 entries (the reference indexes it with len % 56)"
    rows, absorbed, cap = [], 0, [0] * (WIDTH - RATE)
    for k, block in enumerate(blocks):
        state = [int.from_bytes(block[7 * i:7 * i + 7], "little") for i in range(RATE)] + cap
        row = [0] * NUM_COLUMNS
        final = k == len(blocks) - 1
        if final:
            row[IS_FINAL_INPUT_LEN + last_non_padding] = 1
        else:
            row[IS_FULL_INPUT_BLOCK] = 1
        row[CONTEXT], row[SEGMENT], row[VIRT] = addr
        row[TIMESTAMP], row[LEN], row[ALREADY_ABSORBED] = timestamp, length, absorbed
        generate_perm(row, state)
        absorbed += last_non_padding if final else BLOCK_BYTES
        row[NOT_PADDING] = 1
        for i in range(RATE):
            for j in range(FELT_MAX_BYTES - 1):
                row[INPUT_BYTES + 6 * i + j] = block[7 * i + 1 + j]
        cap = [row[DIGEST_COL + 2 * i] + (row[DIGEST_COL + 2 * i + 1] << 32) for i in range(WIDTH - RATE)]
        rows.append(row)
    rows[0][IS_FIRST_ROW_GENERAL_OP] = 1
    return rows


def generate_trace(operations, min_rows):
    """operations: ("simple", [12 field elements]) | ("general", (context, segment, virt), timestamp, padded input
    bytes, len).  -> (322, n) uint64 column-major, n = max(rows, min_rows).next_power_of_two(); padding rows are the
    permutation of the all-zero state with every flag 0."""
    rows = []
    for op in operations:
        if op[0] == "simple":
            row = [0] * NUM_COLUMNS
            generate_perm(row, op[1])
            row[IS_FINAL_INPUT_LEN + RATE - 1] = 1
            row[NOT_PADDING] = 1
            row[IS_SIMPLE_OP] = 1
            rows.append(row)
        else:
            rows += _rows_for_general_op(op[1], op[2], bytes(op[3]), op[4])
    n = max(len(rows), min_rows, 1)
    n = 1 << (n - 1).bit_length()
    pad = [0] * NUM_COLUMNS
    generate_perm(pad, [0] * WIDTH)
    rows += [pad] * (n - len(rows))
    return np.array(rows, dtype=np.uint64).T.copy()


# ---- CTL roles (poseidon_stark.rs:35-137); Table::Poseidon = 9 ----
POSEIDON_TABLE = 9


def _h():
    from . import all_stark as A
    return A


def ctl_looked_simple_op():
    A = _h()
    cols = [A.S(INPUT + i) for i in range(WIDTH)] + [A.S(DIGEST_COL + i) for i in range(2 * DIGEST)]
    return A.TWC(POSEIDON_TABLE, cols, A.simple(A.S(IS_SIMPLE_OP)))


def ctl_looked_general_output():
    A = _h()
    cols = [A.S(DIGEST_COL + i) for i in range(2 * DIGEST)] + [A.S(TIMESTAMP)]
    filt = A.prod(A.LC([(IS_FINAL_INPUT_LEN + i, 1) for i in range(RATE)]), A.LC([(IS_SIMPLE_OP, -1)], 1))
    return A.TWC(POSEIDON_TABLE, cols, filt)


def ctl_looked_general_input():
    A = _h()
    return A.TWC(POSEIDON_TABLE, [A.S(CONTEXT), A.S(SEGMENT), A.S(VIRT), A.S(LEN), A.S(TIMESTAMP)],
                 A.simple(A.S(IS_FIRST_ROW_GENERAL_OP)))


def ctl_looking_memory(i):
    """Byte i of the 56-byte block: (is_read = 1, context, segment, virt + already_absorbed + i, byte, 0 x 7,
    timestamp); the first byte of each 7-byte element is input - sum_j input_bytes[j] * 256^(j+1)."""
    A = _h()
    from .stark import Column
    e, j = divmod(i, FELT_MAX_BYTES)
    if j == 0:
        byte = A.LC([(INPUT + e, 1)] + [(INPUT_BYTES + 6 * e + k, -(1 << (8 * (k + 1)))) for k in range(FELT_MAX_BYTES - 1)])
    else:
        byte = A.S(INPUT_BYTES + 6 * e + j - 1)
    cols = [A.K(1), A.S(CONTEXT), A.S(SEGMENT), A.LC([(VIRT, 1), (ALREADY_ABSORBED, 1)], i), byte]
    cols += [Column() for _ in range(7)] + [A.S(TIMESTAMP)]
    return A.TWC(POSEIDON_TABLE, cols, A.prod(A.S(NOT_PADDING), A.LC([(IS_SIMPLE_OP, -1)], 1)))


def ctl_roles():
    """z-data of the table in `cross_table_lookup_data` order under cdk_erigon: the 56 Memory lookers (CTL 6), then
    looked in poseidon_simple (10), general_input (11), general_output (12) -- all_stark.rs:153-172."""
    return [[ctl_looking_memory(i) for i in range(BLOCK_BYTES)], [ctl_looked_simple_op()], [ctl_looked_general_input()],
            [ctl_looked_general_output()]]
