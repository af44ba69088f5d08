"""oracle/keccak_trace.py -- TEST INFRASTRUCTURE ONLY.
Restatement of the reference's Keccak table witness generator, evm_arithmetization/src/keccak/keccak_stark.rs:65-234
(`generate_trace_rows`, `generate_trace_rows_for_perm`, `copy_output_to_input`, `generate_trace_row_for_round`),
bit by bit as the reference fills the row (columns: keccak/columns.rs:7-134).  Pinned the way the reference pins it
(keccak_stark.rs:657-690): the A''' registers of the last round equal keccak-f[1600] of the input, checked against
the C oracle's permutation, and every generated row pair satisfies the restated AIR (oracle/airs.py eval_keccak)."""
import numpy as np

from .airs import (K_R, K_RC, K_ROUNDS, K_TIMESTAMP, k_reg_a, k_reg_a_pp, k_reg_a_pp_00_bit, k_reg_a_ppp, k_reg_a_prime,
                   k_reg_b, k_reg_c, k_reg_c_prime)

NUM_COLUMNS = 2431


def _row_for_round(row, rnd):
    row[rnd] = 1                                               # reg_step(round)
    for x in range(5):                                         # C[x] = xor of the column
        for z in range(64):
            hi, bit = z // 32, z % 32
            v = 0
            for i in range(5):
                v ^= (int(row[k_reg_a(x, i) + hi]) >> bit) & 1
            row[k_reg_c(x, z)] = v
    for x in range(5):                                         # C'[x, z] = C[x, z] ^ C[x-1, z] ^ C[x+1, z-1]
        for z in range(64):
            row[k_reg_c_prime(x, z)] = int(row[k_reg_c(x, z)]) ^ int(row[k_reg_c((x + 4) % 5, z)]) ^ \
                int(row[k_reg_c((x + 1) % 5, (z + 63) % 64)])
    for x in range(5):                                         # A'[x, y, z] = A ^ C ^ C'
        for y in range(5):
            for z in range(64):
                a_bit = (int(row[k_reg_a(x, y) + z // 32]) >> (z % 32)) & 1
                row[k_reg_a_prime(x, y, z)] = a_bit ^ int(row[k_reg_c(x, z)]) ^ int(row[k_reg_c_prime(x, z)])
    for x in range(5):                                         # A''[x, y] = B ^ (~B[x+1] & B[x+2])
        for y in range(5):
            def get_bit(z):
                b0 = int(row[k_reg_b(x, y, z)])
                b1 = int(row[k_reg_b((x + 1) % 5, y, z)])
                b2 = int(row[k_reg_b((x + 2) % 5, y, z)])
                return b0 ^ ((1 - b1) & b2)
            lo = hi = 0
            for z in range(31, -1, -1):
                lo = 2 * lo + get_bit(z)
            for z in range(63, 31, -1):
                hi = 2 * hi + get_bit(z)
            row[k_reg_a_pp(x, y)] = lo
            row[k_reg_a_pp(x, y) + 1] = hi
    val = int(row[k_reg_a_pp(0, 0)]) | (int(row[k_reg_a_pp(0, 0) + 1]) << 32)
    for i in range(64):
        row[k_reg_a_pp_00_bit(i)] = (val >> i) & 1
    row[k_reg_a_ppp(0, 0)] = int(row[k_reg_a_pp(0, 0)]) ^ (K_RC[rnd] & 0xFFFFFFFF)
    row[k_reg_a_ppp(0, 0) + 1] = int(row[k_reg_a_pp(0, 0) + 1]) ^ (K_RC[rnd] >> 32)


def generate_trace_rows(inputs_and_timestamps, min_rows):
    """-> (num_rows, 2431) uint64, row-major like the reference's Vec<[F; NUM_COLUMNS]>."""
    n = max(len(inputs_and_timestamps) * K_ROUNDS, min_rows, 1)
    num_rows = 1 << (n - 1).bit_length()
    rows = np.zeros((num_rows, NUM_COLUMNS), dtype=np.uint64)
    for p, (inp, ts) in enumerate(inputs_and_timestamps):
        base = p * K_ROUNDS
        for rnd in range(K_ROUNDS):
            rows[base + rnd][K_TIMESTAMP] = ts
        for x in range(5):
            for y in range(5):
                v = int(inp[y * 5 + x])
                rows[base][k_reg_a(x, y)] = v & 0xFFFFFFFF
                rows[base][k_reg_a(x, y) + 1] = v >> 32
        _row_for_round(rows[base], 0)
        for rnd in range(1, K_ROUNDS):
            for x in range(5):                                 # copy_output_to_input
                for y in range(5):
                    rows[base + rnd][k_reg_a(x, y)] = rows[base + rnd - 1][k_reg_a_ppp(x, y)]
                    rows[base + rnd][k_reg_a(x, y) + 1] = rows[base + rnd - 1][k_reg_a_ppp(x, y) + 1]
            _row_for_round(rows[base + rnd], rnd)
    return rows
