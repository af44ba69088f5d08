"""oracle/tracegen.py -- TEST INFRASTRUCTURE ONLY.
Restatements of two more witness generators of the reference, row by row:
  * BytePacking: evm_arithmetization/src/byte_packing/byte_packing_stark.rs:174-283 (`generate_trace`,
    `generate_trace_rows`, `generate_row_for_op`, `generate_range_checks`), columns byte_packing/columns.rs:12-40;
  * KeccakSponge: evm_arithmetization/src/keccak_sponge/keccak_sponge_stark.rs:252-533 (`generate_trace`,
    `generate_rows_for_op`, `generate_full_input_row`, `generate_final_row`, `generate_common_fields`,
    `generate_range_checks`), columns keccak_sponge/columns.rs:31-95.
Both are checked against the restated AIRs (every constraint vanishes on every generated row) in
tests/test_oracle_tracegen.py; the sponge's digest is additionally pinned by keccak256."""
import numpy as np

BYTE_RANGE_MAX = 256
KECCAK_RATE_BYTES = 136


def _next_pow2(n):
    return 1 << max(n - 1, 0).bit_length()


def _range_checks(cols, first, count, counter_col, freq_col, range_max):
    n = cols.shape[1]
    cols[counter_col] = np.minimum(np.arange(n, dtype=np.uint64), np.uint64(range_max - 1))
    cols[freq_col] = 0
    cols[freq_col, :range_max] = np.bincount(cols[first:first + count].astype(np.int64).reshape(-1),
                                             minlength=range_max).astype(np.uint64)


def byte_packing_generate_trace(ops, min_rows):
    """ops: (is_read, (context, segment, virt), timestamp, bytes).  -> (71, n) uint64, column-major."""
    live = [op for op in ops if len(op[3])]
    n = _next_pow2(max(len(ops), BYTE_RANGE_MAX, min_rows))
    t = np.zeros((71, n), dtype=np.uint64)
    for r, (is_read, (ctx, seg, virt), ts, data) in enumerate(live):
        t[0, r] = 1 if is_read else 0
        t[1 + len(data) - 1, r] = 1                      # index_len[len - 1]
        t[33, r], t[34, r], t[35, r], t[36, r] = ctx, seg, virt, ts
        for i, b in enumerate(reversed(bytes(data))):
            t[37 + i, r] = b                             # value_bytes: most significant byte last read first
    _range_checks(t, 37, 32, 69, 70, BYTE_RANGE_MAX)
    return t


def keccak_sponge_generate_trace(ops, min_rows, keccak_f):
    """ops: ((context, segment, virt), timestamp, input bytes); keccak_f(list of 25 u64) -> list of 25 u64.
    -> (438, n) uint64, column-major."""
    rows = []

    def common(row, op, absorbed, state):
        (ctx, seg, virt), ts, _ = op
        row[1], row[2], row[3], row[4], row[5] = ctx, seg, virt, ts, absorbed
        row[142:176] = state[:34]                        # original_rate_u32s
        row[176:192] = state[34:]                        # original_capacity_u32s
        st = list(state)
        for i in range(34):
            blk = sum(int(row[192 + 4 * i + j]) << (8 * j) for j in range(4))
            st[i] ^= blk
        row[328:362] = st[:34]                           # xored_rate_u32s
        w = keccak_f([st[2 * i] | (st[2 * i + 1] << 32) for i in range(25)])
        st = [(w[i // 2] >> (32 * (i % 2))) & 0xFFFFFFFF for i in range(50)]
        row[362:404] = st[8:]                            # partial_updated_state_u32s
        for l in range(8):
            for i in range(4):
                row[404 + 4 * l + i] = (st[l] >> (8 * i)) & 0xFF      # updated_digest_state_bytes
        return st

    for op in ops:
        data = bytes(op[2])
        state = [0] * 50
        absorbed = 0
        while len(data) - absorbed >= KECCAK_RATE_BYTES:
            row = np.zeros(438, dtype=np.uint64)
            row[0] = 1                                   # is_full_input_block
            row[192:328] = list(data[absorbed:absorbed + KECCAK_RATE_BYTES])
            state = common(row, op, absorbed, state)
            rows.append(row)
            absorbed += KECCAK_RATE_BYTES
        rest = data[absorbed:]
        row = np.zeros(438, dtype=np.uint64)
        row[192:192 + len(rest)] = list(rest)
        if len(rest) == KECCAK_RATE_BYTES - 1:           # pad10*1, both bits in one byte
            row[192 + len(rest)] = 0b10000001
        else:
            row[192 + len(rest)] = 1
            row[192 + KECCAK_RATE_BYTES - 1] = 0b10000000
        row[6 + len(rest):6 + KECCAK_RATE_BYTES] = 1     # is_padding_byte
        common(row, op, absorbed, state)
        rows.append(row)
    n = _next_pow2(max(len(rows), min_rows, BYTE_RANGE_MAX))
    t = np.zeros((438, n), dtype=np.uint64)
    if rows:
        t[:, :len(rows)] = np.array(rows, dtype=np.uint64).T
    _range_checks(t, 192, 136, 436, 437, BYTE_RANGE_MAX)
    return t
