"""Host mirror of the reference's table registry and cross-table-lookup wiring
(evm_arithmetization/src/all_stark.rs:34-417, eth_mainnet feature set: 9 tables, 10 CTLs) and of each
table's ``lookups()`` / ``ctl_*`` column definitions.  These are *data*: `Column` / `Filter` programs
that the HIP library interprets (include/zkstark.h "program encoding"); no arithmetic happens here.

Function names follow the reference so a maintainer can diff them:
  arithmetic/arithmetic_stark.rs:33-117,320-327     byte_packing/byte_packing_stark.rs:55-149,426-437
  cpu/cpu_stark.rs:33-466                            keccak/keccak_stark.rs:38-59, keccak/columns.rs:15-41
  keccak_sponge/keccak_sponge_stark.rs:34-229,946-953  logic.rs:84-113
  memory/memory_stark.rs:30-93,858-885               memory_continuation/memory_continuation_stark.rs:30-51
"""
from itertools import chain, islice, repeat
from typing import List

from .stark import Column, CrossTableLookup, Filter, Lookup, TableWithColumns

P = 0xFFFFFFFF00000001


class Table:
    """all_stark.rs:84-99"""
    Arithmetic, BytePacking, Cpu, Keccak, KeccakSponge, Logic, Memory, MemBefore, MemAfter = range(9)

    @staticmethod
    def all(): return list(range(9))


NUM_TABLES = 9
TABLE_NAMES = ["Arithmetic", "BytePacking", "Cpu", "Keccak", "KeccakSponge", "Logic", "Memory", "MemBefore", "MemAfter"]
# the `AllStark` field names: `stringify!($stark)` in prove_table! (prover.rs:227-249) keys the TimingTree scopes
# "prove arithmetic_stark STARK", ...
STARK_FIELD_NAMES = ["arithmetic_stark", "byte_packing_stark", "cpu_stark", "keccak_stark", "keccak_sponge_stark",
                     "logic_stark", "memory_stark", "mem_before_stark", "mem_after_stark"]
# all_stark.rs:124-131
OPTIONAL_TABLE_INDICES = [Table.BytePacking, Table.Keccak, Table.KeccakSponge, Table.Logic, Table.MemAfter]
NUM_CTLS = 10          # all_stark.rs:148
MEMORY_CTL_IDX = 6     # all_stark.rs:149
# trace widths (arithmetic/columns.rs:120, byte_packing/columns.rs:40, cpu/columns/mod.rs:97, keccak/columns.rs:134,
# keccak_sponge/columns.rs:95, logic.rs:71, memory/columns.rs:94, memory_continuation/columns.rs:23)
TABLE_COLUMNS = [116, 71, 85, 2431, 438, 523, 30, 12, 12]
# include/zkstark.h zk_air ids per table (MemBefore and MemAfter share MemoryContinuationStark)
TABLE_AIR = [5, 4, 8, 6, 7, 2, 3, 1, 1]

# ---- memory / segments -----------------------------------------------------------------------
VALUE_LIMBS = 8
NUM_GP_CHANNELS = 3                      # cpu/membus.rs:10
NUM_CHANNELS = 1 + NUM_GP_CHANNELS + 1   # cpu/membus.rs:32 (code, GP, partial)
SEGMENT_CODE, SEGMENT_CONTEXT_METADATA = 0, 6   # memory/segments.rs:16-24 (unscaled ids)
CTX_METADATA_STACK_SIZE = 11             # cpu/kernel/constants/context_metadata.rs:35 (unscaled)


# ============================ Arithmetic =======================================================
class _A:
    IS_ADD, IS_MUL, IS_SUB, IS_DIV, IS_MOD, IS_ADDMOD, IS_MULMOD, IS_ADDFP254, IS_MULFP254, IS_SUBFP254, \
        IS_SUBMOD, IS_LT, IS_GT, IS_BYTE, IS_SHL, IS_SHR, IS_RANGE_CHECK, OPCODE_COL = range(18)
    N_LIMBS = 16
    START_SHARED_COLS = 18
    NUM_SHARED_COLS = 6 * N_LIMBS
    INPUT_REGISTER_0 = range(18, 34)
    INPUT_REGISTER_1 = range(34, 50)
    INPUT_REGISTER_2 = range(50, 66)
    OUTPUT_REGISTER = range(66, 82)
    RANGE_COUNTER = START_SHARED_COLS + NUM_SHARED_COLS
    RC_FREQUENCIES = RANGE_COUNTER + 1


def _cpu_arith_data_link(combined_ops, regs) -> List[Column]:
    res = [Column.linear_combination([(col, code) for col, code in combined_ops])]
    for reg in regs:
        for i in range(_A.N_LIMBS // 2):
            res.append(Column.linear_combination([(reg.start + 2 * i, 1), (reg.start + 2 * i + 1, 1 << 16)]))
    return res


def ctl_arithmetic_rows() -> TableWithColumns:
    combined = [(_A.IS_ADD, 0x01), (_A.IS_MUL, 0x02), (_A.IS_SUB, 0x03), (_A.IS_DIV, 0x04), (_A.IS_MOD, 0x06),
                (_A.IS_ADDMOD, 0x08), (_A.IS_MULMOD, 0x09), (_A.IS_ADDFP254, 0x0c), (_A.IS_MULFP254, 0x0d),
                (_A.IS_SUBFP254, 0x0e), (_A.IS_SUBMOD, 0x0f), (_A.IS_LT, 0x10), (_A.IS_GT, 0x11), (_A.IS_BYTE, 0x1a),
                (_A.IS_SHL, 0x1b), (_A.IS_SHR, 0x1c)]
    regs = [_A.INPUT_REGISTER_0, _A.INPUT_REGISTER_1, _A.INPUT_REGISTER_2, _A.OUTPUT_REGISTER]
    filt = Filter.new_simple(Column.sum([c for c, _ in combined] + [_A.IS_RANGE_CHECK]))
    return TableWithColumns(Table.Arithmetic, _cpu_arith_data_link(combined + [(_A.OPCODE_COL, 0x01)], regs), filt)


def arithmetic_lookups() -> List[Lookup]:
    shared = range(_A.START_SHARED_COLS, _A.START_SHARED_COLS + _A.NUM_SHARED_COLS)
    return [Lookup(Column.singles(shared), Column.single(_A.RANGE_COUNTER), Column.single(_A.RC_FREQUENCIES),
                   [Filter() for _ in shared])]


# ============================ BytePacking ======================================================
class _B:
    NUM_BYTES = 32
    is_read = 0
    index_len = list(range(1, 33))
    addr_context, addr_segment, addr_virtual, timestamp = 33, 34, 35, 36
    value_bytes = list(range(37, 69))
    range_counter, rc_frequencies = 69, 70


def byte_packing_ctl_looked_data() -> List[Column]:
    outputs = [Column.linear_combination([(_B.value_bytes[i * 4] + j, 1 << (8 * j)) for j in range(4)])
               for i in range(8)]
    sequence_len = Column.linear_combination([(_B.index_len[i], i + 1) for i in range(_B.NUM_BYTES)])
    return Column.singles([_B.is_read, _B.addr_context, _B.addr_segment, _B.addr_virtual]) + [sequence_len] + \
        Column.singles([_B.timestamp]) + outputs


def byte_packing_ctl_looked_filter() -> Filter:
    return Filter.new_simple(Column.sum(_B.index_len))


def byte_packing_ctl_looking_memory(i: int) -> List[Column]:
    res = Column.singles([_B.is_read, _B.addr_context, _B.addr_segment])
    cols = [(_B.addr_virtual, 1)] + [(_B.index_len[j], j) for j in range(_B.NUM_BYTES)]
    res.append(Column.linear_combination_with_constant(cols, -i))
    res.append(Column.single(_B.value_bytes[i]))
    res += [Column.zero() for _ in range(1, 8)]
    res.append(Column.single(_B.timestamp))
    return res


def byte_packing_ctl_looking_memory_filter(i: int) -> Filter:
    return Filter.new_simple(Column.sum(_B.index_len[i:]))


def byte_packing_lookups() -> List[Lookup]:
    return [Lookup(Column.singles(_B.value_bytes), Column.single(_B.range_counter), Column.single(_B.rc_frequencies),
                   [Filter() for _ in _B.value_bytes])]


# ============================ Cpu ==============================================================
class _MemChannel:
    def __init__(self, base):
        self.used, self.is_read, self.addr_context, self.addr_segment, self.addr_virtual = range(base, base + 5)
        self.value = list(range(base + 5, base + 13))


class _CpuColumns:
    """cpu/columns/mod.rs:56-97, ops.rs:6-47, general.rs (8-wide union after the opcode bits).  `cdk_erigon` adds the
    `poseidon` flag after jumpdest_keccak_general (ops.rs:22-25): every later column moves by one."""

    def __init__(self, cdk_erigon=False):
        x = 1 if cdk_erigon else 0
        self.context, self.code_context, self.program_counter, self.stack_len, self.is_kernel_mode, self.gas = range(6)
        self.OPS = ["binary_op", "ternary_op", "fp254_op", "eq_iszero", "logic_op", "not_pop", "shift",
                    "jumpdest_keccak_general"] + (["poseidon"] if cdk_erigon else []) + \
                   ["jumps", "push_prover_input", "dup_swap", "context_op", "m_op_32bytes",
                    "exit_kernel", "m_op_general", "pc_push0", "syscall", "exception"]
        self.op = {name: 6 + i for i, name in enumerate(self.OPS)}
        self.opcode_bits = list(range(24 + x, 32 + x))
        self.general = list(range(32 + x, 40 + x))
        self.push_is_not_kernel = 32 + x           # general.push().is_not_kernel
        self.context_pruning_flag = 32 + x         # general.context_pruning().pruning_flag
        self.clock = 40 + x
        self.mem_channels = [_MemChannel(41 + x + 13 * k) for k in range(NUM_GP_CHANNELS)]
        self.partial_channel = _MemChannel(80 + x)   # used, is_read, addr_* only (5 columns)
        self.num_columns = 85 + x


# the column map the cpu_ctl_* functions below read; `all_cross_table_lookups(cdk_erigon=True)` swaps it while it runs
_C = _CpuColumns(False)


MEM_CODE_CHANNEL_IDX = 0
MEM_GP_CHANNELS_IDX_START = 1


def _get_addr(mem_channel: int):
    v = _C.mem_channels[mem_channel].value
    return v[2], v[1], v[0]   # (context, segment, virt)


def _timestamp_col() -> Column:
    # timestamp = (clock - 1) * num_channels + 1
    return Column.linear_combination_with_constant([(_C.clock, NUM_CHANNELS)], 1 - NUM_CHANNELS)


def cpu_ctl_data_keccak_sponge() -> List[Column]:
    context, segment, virt = _get_addr(0)
    cols = [Column.single(context), Column.single(segment), Column.single(virt),
            Column.single(_C.mem_channels[1].value[0]), _timestamp_col()]
    return cols + Column.singles_next_row(_C.mem_channels[0].value)


def cpu_ctl_filter_keccak_sponge() -> Filter:
    return Filter.new([(Column.single(_C.op["jumpdest_keccak_general"]),
                        Column.linear_combination_with_constant([(_C.opcode_bits[1], -1)], 1))], [])


def _ctl_data_binops() -> List[Column]:
    return Column.singles(_C.mem_channels[0].value) + Column.singles(_C.mem_channels[1].value) + \
        Column.singles_next_row(_C.mem_channels[0].value)


def _ctl_data_ternops() -> List[Column]:
    return Column.singles(_C.mem_channels[0].value) + Column.singles(_C.mem_channels[1].value) + \
        Column.singles(_C.mem_channels[2].value) + Column.singles_next_row(_C.mem_channels[0].value)


def cpu_ctl_data_logic() -> List[Column]:
    return [Column.le_bits(_C.opcode_bits)] + _ctl_data_binops()


def cpu_ctl_filter_logic() -> Filter:
    return Filter.new_simple(Column.single(_C.op["logic_op"]))


def cpu_ctl_arithmetic_base_rows() -> TableWithColumns:
    columns = [Column.le_bits(_C.opcode_bits)] + _ctl_data_ternops()
    col_bit = Column.single(_C.opcode_bits[7])
    filt = Filter.new([(Column.single(_C.op["push_prover_input"]), col_bit)],
                      [Column.sum([_C.op[k] for k in ("binary_op", "fp254_op", "ternary_op", "shift", "syscall",
                                                      "exception")])])
    return TableWithColumns(Table.Cpu, columns, filt)


def cpu_ctl_context_pruning_looked() -> TableWithColumns:
    return TableWithColumns(Table.Cpu, [Column.single(_C.context)],
                            Filter.new([(Column.single(_C.op["context_op"]),
                                         Column.single(_C.context_pruning_flag))], []))


def cpu_ctl_data_byte_packing() -> List[Column]:
    return [Column.constant_col(1)] + cpu_ctl_data_keccak_sponge()


def cpu_ctl_filter_byte_packing() -> Filter:
    return Filter.new([(Column.single(_C.op["m_op_32bytes"]), Column.single(_C.opcode_bits[5]))], [])


def cpu_ctl_data_byte_unpacking() -> List[Column]:
    context, segment, virt = _get_addr(0)
    v0 = _C.mem_channels[0].value[0]
    res = [Column.constant_col(0), Column.single(context), Column.single(segment), Column.single(virt),
           Column.linear_combination_and_next_row_with_constant([(v0, -1)], [(v0, 1)], 0), _timestamp_col()]
    return res + Column.singles(_C.mem_channels[1].value)


def cpu_ctl_filter_byte_unpacking() -> Filter:
    return Filter.new([(Column.single(_C.op["m_op_32bytes"]),
                        Column.linear_combination_with_constant([(_C.opcode_bits[5], -1)], 1))], [])


def cpu_ctl_data_jumptable_read() -> List[Column]:
    ch = _C.mem_channels[1]
    res = [Column.constant_col(1)] + Column.singles([ch.addr_context, ch.addr_segment, ch.addr_virtual])
    res.append(Column.constant_col(3))      # len is always 3
    res.append(_timestamp_col())
    return res + Column.singles(ch.value)


def cpu_ctl_filter_syscall_exceptions() -> Filter:
    return Filter.new_simple(Column.sum([_C.op["syscall"], _C.op["exception"]]))


def cpu_ctl_data_byte_packing_push() -> List[Column]:
    res = [Column.constant_col(1), Column.single(_C.code_context), Column.constant_col(SEGMENT_CODE),
           Column.linear_combination_with_constant([(_C.program_counter, 1)], 1),
           Column.le_bits_with_constant(_C.opcode_bits[0:5], 1), _timestamp_col()]
    return res + Column.singles_next_row(_C.mem_channels[0].value)


def cpu_ctl_filter_byte_packing_push() -> Filter:
    return Filter.new([(Column.single(_C.push_is_not_kernel), Column.single(_C.op["push_prover_input"]))], [])


def _mem_time_and_channel(channel: int) -> Column:
    return Column.linear_combination_with_constant([(_C.clock, NUM_CHANNELS)], channel - NUM_CHANNELS + 1)


def cpu_ctl_data_code_memory() -> List[Column]:
    cols = [Column.constant_col(1), Column.single(_C.code_context), Column.constant_col(SEGMENT_CODE),
            Column.single(_C.program_counter), Column.le_bits(_C.opcode_bits)]
    cols += [Column.constant_col(0) for _ in range(VALUE_LIMBS - 1)]
    cols.append(_mem_time_and_channel(MEM_CODE_CHANNEL_IDX))
    return cols


def cpu_ctl_data_gp_memory(channel: int) -> List[Column]:
    ch = _C.mem_channels[channel]
    cols = Column.singles([ch.is_read, ch.addr_context, ch.addr_segment, ch.addr_virtual]) + Column.singles(ch.value)
    cols.append(_mem_time_and_channel(MEM_GP_CHANNELS_IDX_START + channel))
    return cols


def cpu_ctl_data_partial_memory() -> List[Column]:
    ch = _C.partial_channel
    cols = Column.singles([ch.is_read, ch.addr_context, ch.addr_segment, ch.addr_virtual]) + \
        Column.singles(_C.mem_channels[0].value)
    cols.append(_mem_time_and_channel(MEM_GP_CHANNELS_IDX_START + NUM_GP_CHANNELS))
    return cols


def cpu_ctl_data_memory_old_sp_write_set_context() -> List[Column]:
    cols = [Column.constant_col(0), Column.single(_C.context), Column.constant_col(SEGMENT_CONTEXT_METADATA),
            Column.constant_col(CTX_METADATA_STACK_SIZE),
            Column.linear_combination_with_constant([(_C.stack_len, 1)], -1)]
    cols += [Column.constant_col(0) for _ in range(VALUE_LIMBS - 1)]
    cols.append(_mem_time_and_channel(MEM_GP_CHANNELS_IDX_START + 1))
    return cols


def cpu_ctl_data_memory_new_sp_read_set_context() -> List[Column]:
    cols = [Column.constant_col(1), Column.single(_C.mem_channels[0].value[2]),
            Column.constant_col(SEGMENT_CONTEXT_METADATA), Column.constant_col(CTX_METADATA_STACK_SIZE),
            Column.single_next_row(_C.stack_len)]
    cols += [Column.constant_col(0) for _ in range(VALUE_LIMBS - 1)]
    cols.append(_mem_time_and_channel(MEM_GP_CHANNELS_IDX_START + 2))
    return cols


def cpu_ctl_filter_code_memory() -> Filter:
    return Filter.new_simple(Column.sum([_C.op[k] for k in _C.OPS]))


def cpu_ctl_filter_gp_memory(channel: int) -> Filter:
    return Filter.new_simple(Column.single(_C.mem_channels[channel].used))


def cpu_ctl_filter_partial_memory() -> Filter:
    return Filter.new_simple(Column.single(_C.partial_channel.used))


def cpu_ctl_filter_set_context() -> Filter:
    return Filter.new([(Column.single(_C.op["context_op"]), Column.single(_C.opcode_bits[0]))], [])


# ============================ Keccak ===========================================================
class _K:
    NUM_ROUNDS, NUM_INPUTS, TIMESTAMP = 24, 25, 24

    @staticmethod
    def reg_step(i): return i
    @staticmethod
    def reg_a(x, y): return 25 + (x * 5 + y) * 2
    @staticmethod
    def reg_a_prime_prime(x, y): return 2315 + x * 10 + y * 2
    @staticmethod
    def reg_a_prime_prime_prime(x, y): return 2429 if (x == 0 and y == 0) else _K.reg_a_prime_prime(x, y)

    @staticmethod
    def reg_input_limb(i):
        i64 = i // 2
        return Column.single(_K.reg_a(i64 % 5, i64 // 5) + i % 2)

    @staticmethod
    def reg_output_limb(i):
        i64 = i // 2
        return _K.reg_a_prime_prime_prime(i64 % 5, i64 // 5) + i % 2


def keccak_ctl_data_inputs() -> List[Column]:
    return [_K.reg_input_limb(i) for i in range(2 * _K.NUM_INPUTS)] + [Column.single(_K.TIMESTAMP)]


def keccak_ctl_data_outputs() -> List[Column]:
    return Column.singles([_K.reg_output_limb(i) for i in range(2 * _K.NUM_INPUTS)]) + [Column.single(_K.TIMESTAMP)]


def keccak_ctl_filter_inputs() -> Filter:
    return Filter.new_simple(Column.single(_K.reg_step(0)))


def keccak_ctl_filter_outputs() -> Filter:
    return Filter.new_simple(Column.single(_K.reg_step(_K.NUM_ROUNDS - 1)))


# ============================ KeccakSponge =====================================================
class _S:
    KECCAK_RATE_BYTES, KECCAK_RATE_U32S, KECCAK_CAPACITY_U32S = 136, 34, 16
    is_full_input_block, context, segment, virt, timestamp, already_absorbed_bytes = range(6)
    is_padding_byte = list(range(6, 142))
    original_rate_u32s = list(range(142, 176))
    original_capacity_u32s = list(range(176, 192))
    block_bytes = list(range(192, 328))
    xored_rate_u32s = list(range(328, 362))
    partial_updated_state_u32s = list(range(362, 404))
    updated_digest_state_bytes = list(range(404, 436))
    range_counter, rc_frequencies = 436, 437


def keccak_sponge_ctl_looked_data() -> List[Column]:
    outputs = []
    for i in reversed(range(8)):
        outputs.append(Column.linear_combination(
            [(c, 1 << (24 - 8 * j)) for j, c in enumerate(_S.updated_digest_state_bytes[i * 4:(i + 1) * 4])]))
    len_col = Column.linear_combination_with_constant(
        [(_S.already_absorbed_bytes, 1)] + [(_S.is_padding_byte[i], -1) for i in range(_S.KECCAK_RATE_BYTES)],
        _S.KECCAK_RATE_BYTES)
    return Column.singles([_S.context, _S.segment, _S.virt]) + [len_col, Column.single(_S.timestamp)] + outputs


def keccak_sponge_ctl_looking_keccak_inputs() -> List[Column]:
    return Column.singles(_S.xored_rate_u32s + _S.original_capacity_u32s) + [Column.single(_S.timestamp)]


def keccak_sponge_ctl_looking_keccak_outputs() -> List[Column]:
    b = _S.updated_digest_state_bytes
    digest_u32s = [Column.linear_combination([(c, 1 << (8 * i)) for i, c in enumerate(b[k:k + 4])])
                   for k in range(0, 32, 4)]
    return digest_u32s + Column.singles(_S.partial_updated_state_u32s) + [Column.single(_S.timestamp)]


def keccak_sponge_ctl_looking_memory(i: int) -> List[Column]:
    res = [Column.constant_col(1)] + Column.singles([_S.context, _S.segment])
    res.append(Column.linear_combination_with_constant([(_S.virt, 1), (_S.already_absorbed_bytes, 1)], i))
    res.append(Column.single(_S.block_bytes[i]))
    res += [Column.zero() for _ in range(1, 8)]
    res.append(Column.single(_S.timestamp))
    return res


def keccak_sponge_num_logic_ctls() -> int:
    return -(-_S.KECCAK_RATE_BYTES // 32)


def keccak_sponge_ctl_looking_logic(i: int) -> List[Column]:
    U32S_PER_CTL, U8S_PER_CTL = 8, 32
    res = [Column.constant_col(0x18)]    # is_xor
    res += list(islice(chain(Column.singles(_S.original_rate_u32s[i * U32S_PER_CTL:]), repeat(Column.zero())),
                       U32S_PER_CTL))
    bb = _S.block_bytes[i * U8S_PER_CTL:]
    res += list(islice(chain((Column.le_bytes(bb[k:k + 4]) for k in range(0, len(bb), 4)), repeat(Column.zero())),
                       U32S_PER_CTL))
    res += list(islice(chain(Column.singles(_S.xored_rate_u32s[i * U32S_PER_CTL:]), repeat(Column.zero())),
                       U32S_PER_CTL))
    return res


def keccak_sponge_ctl_looked_filter() -> Filter:
    return Filter.new_simple(Column.single(_S.is_padding_byte[_S.KECCAK_RATE_BYTES - 1]))


def keccak_sponge_ctl_looking_memory_filter(i: int) -> Filter:
    if i == _S.KECCAK_RATE_BYTES - 1:
        return Filter.new_simple(Column.single(_S.is_full_input_block))
    return Filter.new_simple(Column.linear_combination(
        [(_S.is_full_input_block, 1), (_S.is_padding_byte[_S.KECCAK_RATE_BYTES - 1], 1), (_S.is_padding_byte[i], -1)]))


def keccak_sponge_ctl_looking_logic_filter() -> Filter:
    return Filter.new_simple(Column.sum([_S.is_full_input_block, _S.is_padding_byte[_S.KECCAK_RATE_BYTES - 1]]))


keccak_sponge_ctl_looking_keccak_filter = keccak_sponge_ctl_looking_logic_filter


def keccak_sponge_lookups() -> List[Lookup]:
    return [Lookup(Column.singles(_S.block_bytes), Column.single(_S.range_counter), Column.single(_S.rc_frequencies),
                   [Filter() for _ in _S.block_bytes])]


# ============================ Logic ============================================================
class _L:
    is_and, is_or, is_xor = 0, 1, 2
    input0 = list(range(3, 259))
    input1 = list(range(259, 515))
    result = list(range(515, 523))
    PACKED_LIMB_BITS = 32


def logic_ctl_data() -> List[Column]:
    res = [Column.linear_combination([(_L.is_and, 0x16), (_L.is_or, 0x17), (_L.is_xor, 0x18)])]
    res += [Column.le_bits(_L.input0[k:k + 32]) for k in range(0, 256, 32)]
    res += [Column.le_bits(_L.input1[k:k + 32]) for k in range(0, 256, 32)]
    return res + Column.singles(_L.result)


def logic_ctl_filter() -> Filter:
    return Filter.new_simple(Column.sum([_L.is_and, _L.is_or, _L.is_xor]))


# ============================ Memory ===========================================================
class _M:
    filter, timestamp, timestamp_inv, is_read, addr_context, addr_segment, addr_virtual = range(7)
    value_limbs = list(range(7, 15))
    context_first_change, segment_first_change, virtual_first_change, initialize_aux, preinitialized_segments, \
        preinitialized_segments_aux, stale_contexts, is_pruned, stale_context_frequencies, is_stale, \
        maybe_in_mem_after, mem_after_filter, range_check, counter, frequencies = range(15, 30)


def memory_ctl_data() -> List[Column]:
    return Column.singles([_M.is_read, _M.addr_context, _M.addr_segment, _M.addr_virtual]) + \
        Column.singles(_M.value_limbs) + [Column.single(_M.timestamp)]


def memory_ctl_filter() -> Filter:
    return Filter.new_simple(Column.single(_M.filter))


def memory_ctl_looking_mem() -> List[Column]:
    return Column.singles([_M.addr_context, _M.addr_segment, _M.addr_virtual]) + Column.singles(_M.value_limbs)


def memory_ctl_context_pruning_looking() -> TableWithColumns:
    return TableWithColumns(Table.Memory, [Column.linear_combination_with_constant([(_M.stale_contexts, 1)], -1)],
                            Filter.new([], [Column.single(_M.is_pruned)]))


def memory_ctl_filter_mem_before() -> Filter:
    return Filter.new([(Column.single(_M.timestamp), Column.linear_combination([(_M.timestamp_inv, -1)]))],
                      [Column.constant_col(1)])


def memory_ctl_filter_mem_after() -> Filter:
    return Filter.new_simple(Column.single(_M.mem_after_filter))


def memory_lookups() -> List[Lookup]:
    return [
        Lookup([Column.single(_M.range_check), Column.single_next_row(_M.addr_virtual)], Column.single(_M.counter),
               Column.single(_M.frequencies),
               [Filter(), Filter.new_simple(Column.sum([_M.context_first_change, _M.segment_first_change]))]),
        Lookup([Column.linear_combination_with_constant([(_M.addr_context, 1)], 1)], Column.single(_M.stale_contexts),
               Column.single(_M.stale_context_frequencies), [Filter.new_simple(Column.single(_M.is_stale))]),
    ]


# ============================ MemBefore / MemAfter =============================================
class _MC:
    FILTER, ADDR_CONTEXT, ADDR_SEGMENT, ADDR_VIRTUAL = range(4)
    @staticmethod
    def value_limb(i): return 4 + i


def memory_continuation_ctl_data() -> List[Column]:
    return Column.singles([_MC.ADDR_CONTEXT, _MC.ADDR_SEGMENT, _MC.ADDR_VIRTUAL]) + \
        Column.singles([_MC.value_limb(i) for i in range(8)])


def memory_continuation_ctl_filter() -> Filter:
    return Filter.new_simple(Column.single(_MC.FILTER))


def memory_continuation_ctl_data_memory() -> List[Column]:
    return [Column.constant_col(0)] + Column.singles([_MC.ADDR_CONTEXT, _MC.ADDR_SEGMENT, _MC.ADDR_VIRTUAL]) + \
        Column.singles([_MC.value_limb(i) for i in range(8)]) + [Column.constant_col(0)]


# ============================ the ten CTLs (all_stark.rs:153-417) =============================
def ctl_arithmetic() -> CrossTableLookup:
    return CrossTableLookup([cpu_ctl_arithmetic_base_rows()], ctl_arithmetic_rows())


def ctl_byte_packing() -> CrossTableLookup:
    looking = [
        TableWithColumns(Table.Cpu, cpu_ctl_data_byte_packing(), cpu_ctl_filter_byte_packing()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_byte_unpacking(), cpu_ctl_filter_byte_unpacking()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_byte_packing_push(), cpu_ctl_filter_byte_packing_push()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_jumptable_read(), cpu_ctl_filter_syscall_exceptions()),
    ]
    return CrossTableLookup(looking, TableWithColumns(Table.BytePacking, byte_packing_ctl_looked_data(),
                                                      byte_packing_ctl_looked_filter()))


def ctl_keccak_sponge() -> CrossTableLookup:
    return CrossTableLookup(
        [TableWithColumns(Table.Cpu, cpu_ctl_data_keccak_sponge(), cpu_ctl_filter_keccak_sponge())],
        TableWithColumns(Table.KeccakSponge, keccak_sponge_ctl_looked_data(), keccak_sponge_ctl_looked_filter()))


def ctl_keccak_inputs() -> CrossTableLookup:
    return CrossTableLookup(
        [TableWithColumns(Table.KeccakSponge, keccak_sponge_ctl_looking_keccak_inputs(),
                          keccak_sponge_ctl_looking_keccak_filter())],
        TableWithColumns(Table.Keccak, keccak_ctl_data_inputs(), keccak_ctl_filter_inputs()))


def ctl_keccak_outputs() -> CrossTableLookup:
    return CrossTableLookup(
        [TableWithColumns(Table.KeccakSponge, keccak_sponge_ctl_looking_keccak_outputs(),
                          keccak_sponge_ctl_looking_keccak_filter())],
        TableWithColumns(Table.Keccak, keccak_ctl_data_outputs(), keccak_ctl_filter_outputs()))


def ctl_logic() -> CrossTableLookup:
    lookers = [TableWithColumns(Table.Cpu, cpu_ctl_data_logic(), cpu_ctl_filter_logic())]
    for i in range(keccak_sponge_num_logic_ctls()):
        lookers.append(TableWithColumns(Table.KeccakSponge, keccak_sponge_ctl_looking_logic(i),
                                        keccak_sponge_ctl_looking_logic_filter()))
    return CrossTableLookup(lookers, TableWithColumns(Table.Logic, logic_ctl_data(), logic_ctl_filter()))


def ctl_memory() -> CrossTableLookup:
    lookers = [
        TableWithColumns(Table.Cpu, cpu_ctl_data_code_memory(), cpu_ctl_filter_code_memory()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_partial_memory(), cpu_ctl_filter_partial_memory()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_memory_old_sp_write_set_context(), cpu_ctl_filter_set_context()),
        TableWithColumns(Table.Cpu, cpu_ctl_data_memory_new_sp_read_set_context(), cpu_ctl_filter_set_context()),
    ]
    lookers += [TableWithColumns(Table.Cpu, cpu_ctl_data_gp_memory(ch), cpu_ctl_filter_gp_memory(ch))
                for ch in range(NUM_GP_CHANNELS)]
    lookers += [TableWithColumns(Table.KeccakSponge, keccak_sponge_ctl_looking_memory(i),
                                 keccak_sponge_ctl_looking_memory_filter(i)) for i in range(_S.KECCAK_RATE_BYTES)]
    lookers += [TableWithColumns(Table.BytePacking, byte_packing_ctl_looking_memory(i),
                                 byte_packing_ctl_looking_memory_filter(i)) for i in range(32)]
    lookers.append(TableWithColumns(Table.MemBefore, memory_continuation_ctl_data_memory(),
                                    memory_continuation_ctl_filter()))
    return CrossTableLookup(lookers, TableWithColumns(Table.Memory, memory_ctl_data(), memory_ctl_filter()))


def ctl_context_pruning() -> CrossTableLookup:
    return CrossTableLookup([memory_ctl_context_pruning_looking()], cpu_ctl_context_pruning_looked())


def ctl_mem_before() -> CrossTableLookup:
    return CrossTableLookup(
        [TableWithColumns(Table.Memory, memory_ctl_looking_mem(), memory_ctl_filter_mem_before())],
        TableWithColumns(Table.MemBefore, memory_continuation_ctl_data(), memory_continuation_ctl_filter()))


def ctl_mem_after() -> CrossTableLookup:
    return CrossTableLookup(
        [TableWithColumns(Table.Memory, memory_ctl_looking_mem(), memory_ctl_filter_mem_after())],
        TableWithColumns(Table.MemAfter, memory_continuation_ctl_data(), memory_continuation_ctl_filter()))


# ============================ Poseidon (cdk_erigon) ===========================================
class _P:
    """poseidon/columns.rs:14-94"""
    context, segment, virt, timestamp, len, already_absorbed_elements = range(6)
    is_final_input_len = list(range(6, 14))
    is_full_input_block = 14
    input = list(range(15, 27))
    digest = list(range(251, 259))
    input_bytes = [[271 + 6 * i + j for j in range(6)] for i in range(8)]
    is_simple_op, is_first_row_general_op, not_padding = 319, 320, 321
    NUM_COLUMNS = 322
    FELT_MAX_BYTES, SPONGE_RATE = 7, 8


POSEIDON_TABLE = 9     # Table::Poseidon (all_stark.rs:96-97)


def poseidon_ctl_looked_simple_op() -> TableWithColumns:                    # poseidon_stark.rs:35-43
    return TableWithColumns(POSEIDON_TABLE, Column.singles(_P.input) + Column.singles(_P.digest),
                            Filter.new_simple(Column.single(_P.is_simple_op)))


def poseidon_ctl_looked_general_output() -> TableWithColumns:               # poseidon_stark.rs:45-63
    cols = Column.singles(_P.digest) + [Column.single(_P.timestamp)]
    filt = Filter.new([(Column.sum(_P.is_final_input_len),
                        Column.linear_combination_with_constant([(_P.is_simple_op, -1)], 1))], [])
    return TableWithColumns(POSEIDON_TABLE, cols, filt)


def poseidon_ctl_looked_general_input() -> TableWithColumns:                # poseidon_stark.rs:65-79
    return TableWithColumns(POSEIDON_TABLE, Column.singles([_P.context, _P.segment, _P.virt, _P.len, _P.timestamp]),
                            Filter.new_simple(Column.single(_P.is_first_row_general_op)))


def poseidon_ctl_looking_memory(i: int) -> List[Column]:                    # poseidon_stark.rs:81-124
    res = [Column.constant_col(1)] + Column.singles([_P.context, _P.segment])
    res.append(Column.linear_combination_with_constant([(_P.virt, 1), (_P.already_absorbed_elements, 1)], i))
    e, j = divmod(i, _P.FELT_MAX_BYTES)
    if j == 0:
        res.append(Column.linear_combination([(_P.input[e], 1)] +
                                             [(_P.input_bytes[e][k], -(1 << (8 * (k + 1)))) for k in range(_P.FELT_MAX_BYTES - 1)]))
    else:
        res.append(Column.single(_P.input_bytes[e][j - 1]))
    res += [Column.zero() for _ in range(1, 8)]
    res.append(Column.single(_P.timestamp))
    return res


def poseidon_ctl_looking_memory_filter() -> Filter:                         # poseidon_stark.rs:126-137
    return Filter.new([(Column.single(_P.not_padding),
                        Column.linear_combination_with_constant([(_P.is_simple_op, -1)], 1))], [])


def cpu_ctl_poseidon_simple_filter() -> Filter:                             # cpu_stark.rs:510-521
    return Filter.new([(Column.single(_C.op["poseidon"]),
                        Column.linear_combination_with_constant([(_C.opcode_bits[0], -1)], 1))], [])


def cpu_ctl_poseidon_general_filter() -> Filter:                            # cpu_stark.rs:523-534
    return Filter.new([(Column.single(_C.op["poseidon"]), Column.single(_C.opcode_bits[0]))], [])


def cpu_ctl_poseidon_simple_op() -> TableWithColumns:                       # cpu_stark.rs:465-487
    cols = []
    for channel in range(3):
        v = _C.mem_channels[channel].value
        cols += [Column.linear_combination([(v[2 * i], 1), (v[2 * i + 1], 1 << 32)]) for i in range(VALUE_LIMBS // 2)]
    cols += Column.singles_next_row(_C.mem_channels[0].value)
    return TableWithColumns(Table.Cpu, cols, cpu_ctl_poseidon_simple_filter())


def cpu_ctl_poseidon_general_input() -> TableWithColumns:                   # cpu_stark.rs:489-508
    context, segment, virt = _get_addr(0)
    cols = Column.singles([context, segment, virt, _C.mem_channels[1].value[0]])
    cols.append(Column.linear_combination([(_C.clock, NUM_CHANNELS)]))
    return TableWithColumns(Table.Cpu, cols, cpu_ctl_poseidon_general_filter())


def cpu_ctl_poseidon_general_output() -> TableWithColumns:                  # cpu_stark.rs:536-544
    cols = Column.singles_next_row(_C.mem_channels[0].value) + [Column.linear_combination([(_C.clock, NUM_CHANNELS)])]
    return TableWithColumns(Table.Cpu, cols, cpu_ctl_poseidon_general_filter())


def all_cross_table_lookups(cdk_erigon: bool = False) -> List[CrossTableLookup]:
    """all_stark.rs:153-172 (order is part of the protocol: CTL z-data are appended per table in it).  With
    `cdk_erigon`: the Cpu columns of that build, 56 more Memory lookers (the Poseidon table's byte reads,
    all_stark.rs:344-366) and the three Poseidon CTLs (:419-441)."""
    global _C
    saved = _C
    _C = _CpuColumns(cdk_erigon)
    try:
        memory = ctl_memory()
        if cdk_erigon:
            memory.looking_tables += [TableWithColumns(POSEIDON_TABLE, poseidon_ctl_looking_memory(i),
                                                       poseidon_ctl_looking_memory_filter())
                                      for i in range(_P.FELT_MAX_BYTES * _P.SPONGE_RATE)]
        ctls = [ctl_arithmetic(), ctl_byte_packing(), ctl_keccak_sponge(), ctl_keccak_inputs(), ctl_keccak_outputs(),
                ctl_logic(), memory, ctl_mem_before(), ctl_mem_after(), ctl_context_pruning()]
        if cdk_erigon:
            ctls += [CrossTableLookup([cpu_ctl_poseidon_simple_op()], poseidon_ctl_looked_simple_op()),
                     CrossTableLookup([cpu_ctl_poseidon_general_input()], poseidon_ctl_looked_general_input()),
                     CrossTableLookup([cpu_ctl_poseidon_general_output()], poseidon_ctl_looked_general_output())]
        return ctls
    finally:
        _C = saved


def table_lookups(table: int) -> List[Lookup]:
    """`Stark::lookups()` of each table (tables without range checks return none)."""
    return {Table.Arithmetic: arithmetic_lookups, Table.BytePacking: byte_packing_lookups,
            Table.KeccakSponge: keccak_sponge_lookups, Table.Memory: memory_lookups}.get(table, lambda: [])()


class AllStark:
    """all_stark.rs:34-76: the tables (by AIR id and width) plus the CTL list; `air_consts` carries the four kernel
    labels the Cpu AIR needs (include/zkstark.h ZK_AIR_CPU).  cdk_erigon=True is the ten-table feature set: the
    86-column Cpu table (ZK_AIR_CPU_ERIGON), the Poseidon table (ZK_AIR_POSEIDON, optional) and 13 CTLs."""

    def __init__(self, cpu_air_consts=(0, 0, 0, 0), cdk_erigon: bool = False):
        self.cdk_erigon = bool(cdk_erigon)
        self.cross_table_lookups = all_cross_table_lookups(cdk_erigon)
        self.table_air = list(TABLE_AIR)
        self.table_columns = list(TABLE_COLUMNS)
        self.table_names = list(TABLE_NAMES)
        self.stark_field_names = list(STARK_FIELD_NAMES)
        self.optional_table_indices = list(OPTIONAL_TABLE_INDICES)
        if cdk_erigon:
            self.table_air[Table.Cpu] = 10
            self.table_columns[Table.Cpu] = 86
            self.table_air.append(9)
            self.table_columns.append(_P.NUM_COLUMNS)
            self.table_names.append("Poseidon")
            self.stark_field_names.append("poseidon_stark")
            self.optional_table_indices.append(POSEIDON_TABLE)
        self.num_tables = len(self.table_air)
        self.lookups = [table_lookups(t) for t in range(self.num_tables)]
        self.air_consts = [tuple(cpu_air_consts) if t == Table.Cpu else () for t in range(self.num_tables)]
        self.constraint_degree = 3
