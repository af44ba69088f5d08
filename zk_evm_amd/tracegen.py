"""Device trace generators.  `keccak_generate_trace` binds zk_keccak_generate_trace, the
replacement of `KeccakStark::generate_trace` (evm_arithmetization/src/keccak/keccak_stark.rs:65-259)."""
import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .context import Context, default_context

NUM_ROUNDS = 24
NUM_INPUTS = 25
KECCAK_COLUMNS = 2431


def keccak_generate_trace(inputs_and_timestamps: Sequence[Tuple[Sequence[int], int]], min_rows: int, device=0,
                          ctx: Context = None):
    """-> CUDA int64 tensor (2431, num_rows): the column-major `Vec<PolynomialValues<F>>` of the Keccak table;
    num_rows = max(24 * len(inputs), min_rows).next_power_of_two() as in the reference."""
    import torch
    n_perms = len(inputs_and_timestamps)
    n = max(n_perms * NUM_ROUNDS, min_rows, 1)
    log_n = (n - 1).bit_length()
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    if isinstance(inputs_and_timestamps, tuple) and len(inputs_and_timestamps) == 2 and isinstance(inputs_and_timestamps[0], np.ndarray):
        inp, ts = (np.ascontiguousarray(a, dtype=np.uint64) for a in inputs_and_timestamps)   # packed: ((n, 25), (n,))
        n_perms = inp.shape[0]
        n = max(n_perms * NUM_ROUNDS, min_rows, 1)
        log_n = (n - 1).bit_length()
    else:
        inp = np.array([[int(w) for w in i] for i, _ in inputs_and_timestamps], dtype=np.uint64).reshape(n_perms, NUM_INPUTS)
        ts = np.array([int(t) for _, t in inputs_and_timestamps], dtype=np.uint64)
    if inp.shape != (n_perms, NUM_INPUTS):
        raise ZkStarkError(-1, "every Keccak input is 25 words")
    out = torch.empty((KECCAK_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_keccak_generate_trace(ctx.handle, inp.ctypes.data if n_perms else None,
                                               ts.ctypes.data if n_perms else None, n_perms, log_n,
                                               C.c_void_p(out.data_ptr()), 1 << log_n))
    return out


def range_check_columns(trace, first_col: int, n_cols: int, counter_col: int, freq_col: int, range_max: int,
                        ctx: Context = None) -> None:
    """In place on a CUDA (columns, 2^k) tensor: the `generate_range_checks` step of the Arithmetic (range 2^16,
    columns 18..113 -> 114 / 115), BytePacking (256, 37..68 -> 69 / 70) and KeccakSponge (256, 192..327 -> 436 / 437)
    tables."""
    from .stark import _trace_args
    n_trace_cols, n, log_n, stride = _trace_args(trace)
    ctx = ctx or default_context(trace.device.index or 0)
    ctx.use_torch_current_stream()
    ctx.check(ctx.lib.zk_range_check_columns(ctx.handle, C.c_void_p(trace.data_ptr()), stride, n_trace_cols, log_n,
                                             first_col, n_cols, counter_col, freq_col, range_max))


LOGIC_COLUMNS = 523
OP_AND, OP_OR, OP_XOR = 0, 1, 2


def logic_generate_trace(operations: Sequence[Tuple[int, int, int]], min_rows: int, device=0, ctx: Context = None):
    """`LogicStark::generate_trace` (logic.rs:189-240).  operations: (operator, input0, input1) with 256-bit ints;
    -> CUDA int64 tensor (523, max(len, min_rows).next_power_of_two())."""
    import torch
    n_ops = len(operations)
    n = max(n_ops, min_rows, 1)
    log_n = (n - 1).bit_length()
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    m64 = (1 << 64) - 1
    packed = isinstance(operations, np.ndarray)              # the C ABI's record layout: (n, 9) uint64
    if packed and (operations.dtype != np.uint64 or operations.ndim != 2 or operations.shape[1] != 9):
        raise ZkStarkError(-1, "packed logic operations are a (n, 9) uint64 array")
    flat = np.ascontiguousarray(operations) if packed else np.zeros((n_ops, 9), dtype=np.uint64)
    for r, (op, a, b) in enumerate(() if packed else operations):
        if not (0 <= a < (1 << 256) and 0 <= b < (1 << 256)):
            raise ZkStarkError(-1, "logic inputs are U256")
        flat[r] = [op] + [(a >> (64 * l)) & m64 for l in range(4)] + [(b >> (64 * l)) & m64 for l in range(4)]
    out = torch.empty((LOGIC_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_logic_generate_trace(ctx.handle, flat.ctypes.data if n_ops else None, n_ops, log_n,
                                              C.c_void_p(out.data_ptr()), 1 << log_n))
    return out


MEM_CONTINUATION_COLUMNS = 12


def memory_continuation_generate_trace(mem_values: Sequence[Tuple[Tuple[int, int, int], int]], device=0, ctx: Context = None):
    """`mem_before_values_to_rows` + `MemoryContinuationStark::generate_trace`
    (memory_continuation_stark.rs:53-98).  mem_values: ((context, segment, virt), value U256);
    -> CUDA int64 tensor (12, max(128, len.next_power_of_two()))."""
    import torch
    n = len(mem_values)
    rows = max(128, 1 << max(n - 1, 0).bit_length()) if n else 128
    log_n = rows.bit_length() - 1
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    m64 = (1 << 64) - 1
    packed = isinstance(mem_values, np.ndarray)              # the C ABI's record layout: (n, 7) uint64
    if packed and (mem_values.dtype != np.uint64 or mem_values.ndim != 2 or mem_values.shape[1] != 7):
        raise ZkStarkError(-1, "packed mem values are a (n, 7) uint64 array")
    flat = np.ascontiguousarray(mem_values) if packed else np.zeros((n, 7), dtype=np.uint64)
    for r, ((c, s, v), val) in enumerate(() if packed else mem_values):
        flat[r] = [c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)]
    out = torch.empty((MEM_CONTINUATION_COLUMNS, rows), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_memory_continuation_generate_trace(ctx.handle, flat.ctypes.data if n else None, n, log_n,
                                                            C.c_void_p(out.data_ptr()), rows))
    return out


POSEIDON_COLUMNS = 322
GOLDILOCKS_P = 0xFFFFFFFF00000001


def poseidon_generate_trace(operations, min_rows: int, device=0, ctx: Context = None):
    """`PoseidonStark::generate_trace(operations, min_rows)` (poseidon/poseidon_stark.rs:407-425; `cdk_erigon`).
    operations: ("simple", [12 field elements]) for PoseidonSimpleOp, ("general", (context, segment, virt), timestamp,
    padded input bytes, len) for PoseidonGeneralOp.  -> CUDA int64 (322, max(rows, min_rows).next_power_of_two())."""
    import torch
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    n_ops = len(operations)
    flat = np.zeros((n_ops, 13), dtype=np.uint64)
    data = bytearray()
    rows = 0
    for r, op in enumerate(operations):
        if op[0] == "simple":
            if len(op[1]) != 12 or any(not 0 <= int(x) < GOLDILOCKS_P for x in op[1]):
                raise ZkStarkError(-1, "a simple operation takes 12 canonical field elements")
            flat[r, 1:13] = [int(x) for x in op[1]]
            rows += 1
        else:
            _, (c, s, v), ts, inp, length = op
            inp = bytes(inp)
            flat[r, 0:7] = [1, c, s, v, ts, length, len(inp)]
            data += inp
            rows += len(inp) // 56
    log_n = _pow2_log(max(rows, min_rows, 1))
    out = torch.empty((POSEIDON_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if data else np.zeros(0, dtype=np.uint8)
    ctx.check(ctx.lib.zk_poseidon_generate_trace(ctx.handle, flat.ctypes.data if n_ops else None, n_ops,
                                                 buf.ctypes.data if buf.size else None, buf.size, log_n,
                                                 C.c_void_p(out.data_ptr()), 1 << log_n))
    return out


ARITHMETIC_COLUMNS = 116
# the operation's flag column (arithmetic/columns.rs:25-45); 16 = RangeCheckOperation
(ARITH_ADD, ARITH_MUL, ARITH_SUB, ARITH_DIV, ARITH_MOD, ARITH_ADDMOD, ARITH_MULMOD, ARITH_ADDFP254, ARITH_MULFP254,
 ARITH_SUBFP254, ARITH_SUBMOD, ARITH_LT, ARITH_GT, ARITH_BYTE, ARITH_SHL, ARITH_SHR, ARITH_RANGE_CHECK) = range(17)
_ARITH_TWO_ROWS = (ARITH_DIV, ARITH_MOD, ARITH_SHR, ARITH_ADDMOD, ARITH_MULMOD, ARITH_SUBMOD, ARITH_ADDFP254,
                   ARITH_MULFP254, ARITH_SUBFP254)


def arithmetic_generate_trace(operations, device=0, ctx: Context = None):
    """`ArithmeticStark::generate_trace(operations)` (arithmetic_stark.rs:158-190) on the device.  operations:
    (code, input0, input1) for binary operations (BYTE: index, value; SHL / SHR: shift, value; FP254: two inputs),
    (code, input0, input1, input2) for ADDMOD / MULMOD / SUBMOD, (ARITH_RANGE_CHECK, input0, input1, input2, opcode,
    result) for range-check rows.  -> (CUDA int64 (116, max(2^16, pow2(rows))), rows used)."""
    import torch
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    m64 = (1 << 64) - 1
    n_ops = len(operations)
    packed = isinstance(operations, np.ndarray)          # the C ABI's own record layout: (n_ops, 18) uint64
    if packed:
        if operations.dtype != np.uint64 or operations.ndim != 2 or operations.shape[1] != 18:
            raise ZkStarkError(-1, "packed operations are a (n, 18) uint64 array (include/zkstark.h)")
        flat = np.ascontiguousarray(operations)
        rows = int(np.isin(flat[:, 0], _ARITH_TWO_ROWS).sum()) + n_ops
    else:
        flat = np.zeros((n_ops, 18), dtype=np.uint64)
        rows = 0
    for r, op in enumerate(() if packed else operations):
        code = int(op[0])
        if code == ARITH_RANGE_CHECK:
            _, a, b, c, opcode, res = op
        else:
            a, b = op[1], op[2]
            c = op[3] if len(op) > 3 else 0
            opcode, res = 0, 0
        flat[r, 0], flat[r, 1] = code, opcode
        for k, v in enumerate((a, b, c, res)):
            if not 0 <= v < 1 << 256:
                raise ZkStarkError(-1, "operands are U256")
            flat[r, 2 + 4 * k:6 + 4 * k] = [(v >> (64 * l)) & m64 for l in range(4)]
        rows += 2 if code in _ARITH_TWO_ROWS else 1
    n = max(1 << max(rows - 1, 0).bit_length(), 1 << 16)
    out = torch.empty((ARITHMETIC_COLUMNS, n), dtype=torch.int64, device=f"cuda:{device}")
    used = C.c_size_t()
    ctx.check(ctx.lib.zk_arithmetic_generate_trace(ctx.handle, flat.ctypes.data if n_ops else None, n_ops,
                                                   n.bit_length() - 1, C.c_void_p(out.data_ptr()), n, C.byref(used)))
    return out, used.value


MEMORY_COLUMNS = 30


def memory_generate_trace(memory_ops, mem_before_values=(), stale_contexts=(), device=0, ctx: Context = None,
                          packed_final: bool = False):
    """`MemoryStark::generate_trace(memory_ops, mem_before_values, stale_contexts)` (memory/memory_stark.rs:405-455)
    on the device: sort, fill_gaps, padding, flags, range-check / frequency / stale-context columns, final memory.
    memory_ops: (filter, timestamp, (context, segment, virt), is_read, value U256); mem_before_values: ((context,
    segment, virt), value).  -> (trace: CUDA int64 (30, 2^k), mem_after: CUDA int64 (12, max(128, pow2)) MemAfter
    table, final_values: [((context, segment, virt), value)] in row order -- or, with packed_final, the C ABI's own (k, 7)
    uint64 array [context, segment, virt, value limbs], which costs no per-entry Python objects --, unpadded_length)."""
    import torch
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    m64 = (1 << 64) - 1
    n_ops, n_before = len(memory_ops), len(mem_before_values)
    # numpy arrays in the C ABI's record layout ((n, 9) / (n, 7) uint64, include/zkstark.h) are passed through
    if isinstance(memory_ops, np.ndarray):
        if memory_ops.dtype != np.uint64 or memory_ops.ndim != 2 or memory_ops.shape[1] != 9:
            raise ZkStarkError(-1, "packed memory operations are a (n, 9) uint64 array")
        ops = np.ascontiguousarray(memory_ops)
    else:
        ops = np.zeros((n_ops, 9), dtype=np.uint64)
        for r, (filt, ts, (c, s, v), is_read, val) in enumerate(memory_ops):
            ops[r] = [(1 if is_read else 0) | (2 if filt else 0), ts, c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)]
    if isinstance(mem_before_values, np.ndarray):
        if mem_before_values.dtype != np.uint64 or mem_before_values.ndim != 2 or mem_before_values.shape[1] != 7:
            raise ZkStarkError(-1, "packed mem_before values are a (n, 7) uint64 array")
        before = np.ascontiguousarray(mem_before_values)
    else:
        before = np.zeros((n_before, 7), dtype=np.uint64)
        for r, ((c, s, v), val) in enumerate(mem_before_values):
            before[r] = [c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)]
    stale = np.array([int(x) for x in stale_contexts], dtype=np.uint64)
    gen = C.c_void_p()
    ctx.check(ctx.lib.zk_memory_trace_begin(ctx.handle, ops.ctypes.data if n_ops else None, n_ops,
                                            before.ctypes.data if n_before else None, n_before, C.byref(gen)))
    try:
        log_n = ctx.lib.zk_memory_gen_log_n(gen)
        unpadded = ctx.lib.zk_memory_gen_unpadded_length(gen)
        trace = torch.empty((MEMORY_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
        n_after = C.c_size_t()
        ctx.check(ctx.lib.zk_memory_trace_finish(ctx.handle, gen, stale.ctypes.data if stale.size else None, stale.size,
                                                 C.c_void_p(trace.data_ptr()), 1 << log_n, C.byref(n_after)))
        k = n_after.value
        flat = np.zeros((k, 7), dtype=np.uint64)
        ctx.check(ctx.lib.zk_memory_gen_final_values(ctx.handle, gen, flat.ctypes.data if k else None))
        rows = max(128, 1 << max(k - 1, 0).bit_length())
        after = torch.empty((MEM_CONTINUATION_COLUMNS, rows), dtype=torch.int64, device=f"cuda:{device}")
        ctx.check(ctx.lib.zk_memory_gen_mem_after_trace(ctx.handle, gen, rows.bit_length() - 1,
                                                        C.c_void_p(after.data_ptr()), rows))
    finally:
        ctx.lib.zk_memory_gen_free(gen)
    final = flat if packed_final else [((int(e[0]), int(e[1]), int(e[2])), sum(int(e[3 + l]) << (64 * l) for l in range(4)))
                                       for e in flat]
    return trace, after, final, unpadded


def initial_memory_merkle_cap(kernel_code: bytes, rate_bits: int, cap_height: int, hasher: int = 0, device=0,
                              ctx: Context = None) -> np.ndarray:
    """`initial_memory_merkle_cap::<F, C, D>(rate_bits, cap_height)` (verifier.rs:14-78) for a given kernel image
    (the reference reads the global KERNEL.code).  -> (2^cap_height, 4) uint64."""
    from .config import StarkConfig
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    cfg = StarkConfig(hasher=hasher).to_c(rate_bits=rate_bits, cap_height=cap_height)
    code = np.frombuffer(bytes(kernel_code), dtype=np.uint8)
    out = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    ctx.check(ctx.lib.zk_initial_memory_merkle_cap(ctx.handle, C.byref(cfg), code.ctypes.data if code.size else None,
                                                   code.size, out.ctypes.data))
    return out


BYTE_PACKING_COLUMNS, KECCAK_SPONGE_COLUMNS, BYTE_RANGE_MAX, KECCAK_RATE_BYTES = 71, 438, 256, 136


def _pow2_log(n: int) -> int:
    return max(n - 1, 0).bit_length()


def byte_packing_generate_trace(ops, min_rows: int, device=0, ctx: Context = None):
    """`BytePackingStark::generate_trace` (byte_packing_stark.rs:174-283).  ops: (is_read, (context, segment, virt),
    timestamp, bytes); -> CUDA int64 tensor (71, max(len(ops), 256, min_rows).next_power_of_two())."""
    import torch
    log_n = _pow2_log(max(len(ops), BYTE_RANGE_MAX, min_rows))
    packed = isinstance(ops, np.ndarray)                     # the C ABI's record layout: (n, 10) uint64, no empty operations
    if packed and (ops.dtype != np.uint64 or ops.ndim != 2 or ops.shape[1] != 10):
        raise ZkStarkError(-1, "packed byte-packing operations are a (n, 10) uint64 array")
    live = ops if packed else [op for op in ops if len(op[3])]
    flat = np.ascontiguousarray(ops) if packed else np.zeros((len(live), 10), dtype=np.uint64)
    for r, (is_read, (c, s, v), ts, data) in enumerate(() if packed else live):
        data = bytes(data)
        if len(data) > 32:
            raise ZkStarkError(-1, "byte sequences are at most 32 bytes")
        words = [int.from_bytes(data[8 * k:8 * k + 8].ljust(8, b"\0"), "little") for k in range(4)]
        flat[r] = [1 if is_read else 0, c, s, v, ts, len(data)] + words
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    out = torch.empty((BYTE_PACKING_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_byte_packing_generate_trace(ctx.handle, flat.ctypes.data if len(live) else None, len(live), log_n,
                                                     C.c_void_p(out.data_ptr()), 1 << log_n))
    return out


def keccak_sponge_generate_trace(ops, min_rows: int, device=0, ctx: Context = None):
    """`KeccakSpongeStark::generate_trace` (keccak_sponge_stark.rs:252-533).  ops: ((context, segment, virt), timestamp,
    input bytes); -> CUDA int64 tensor (438, max(rows, 256, min_rows).next_power_of_two())."""
    import torch
    rows = sum(len(bytes(d)) // KECCAK_RATE_BYTES + 1 for _, _, d in ops)
    log_n = _pow2_log(max(rows, BYTE_RANGE_MAX, min_rows))
    flat = np.array([[c, s, v, ts, len(bytes(d))] for (c, s, v), ts, d in ops], dtype=np.uint64).reshape(len(ops), 5)
    blob = np.frombuffer(b"".join(bytes(d) for _, _, d in ops), dtype=np.uint8)
    ctx = ctx or default_context(device)
    ctx.use_torch_current_stream()
    out = torch.empty((KECCAK_SPONGE_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_keccak_sponge_generate_trace(ctx.handle, flat.ctypes.data if len(ops) else None, len(ops),
                                                      blob.ctypes.data if blob.size else None, blob.size, log_n,
                                                      C.c_void_p(out.data_ptr()), 1 << log_n))
    return out


class Traces:
    """`witness::traces::Traces` (witness/traces.rs:36-48): the operation logs the interpreter hands over, and
    `into_tables` (:135-262), which turns them into the per-table `Vec<PolynomialValues>` -- here column-major CUDA
    tensors built by the device generators, ready for `segment.prove_with_traces`.

    arithmetic_ops / byte_packing_ops / logic_ops / memory_ops / keccak_inputs / keccak_sponge_ops / poseidon_ops are
    lists in the formats of the corresponding `*_generate_trace` functions of this module; `cpu` is the (rows, 85 or
    86) array of `CpuColumnsView` rows, already padded to a power of two by the caller (the interpreter appends the
    halting rows)."""

    def __init__(self):
        self.arithmetic_ops, self.byte_packing_ops, self.cpu, self.logic_ops = [], [], None, []
        self.memory_ops, self.keccak_inputs, self.keccak_sponge_ops, self.poseidon_ops = [], [], [], []

    def into_tables(self, all_stark, mem_before_values, stale_contexts, config, device=0, ctx: Context = None,
                    packed_final: bool = False):
        """-> (tables in `Table` order, final_values): `Traces::into_tables(all_stark, mem_before_values,
        stale_contexts, trace_lengths, config, timing)`.  min_rows of the optional tables = the number of cap elements,
        as in the reference (:148)."""
        import torch
        cap_elements = 1 << config.fri_config.cap_height
        cpu = self.cpu
        if not torch.is_tensor(cpu):
            cpu = torch.from_numpy(np.ascontiguousarray(np.asarray(cpu, dtype=np.uint64)).view(np.int64))
        cpu = cpu.to(f"cuda:{device}")
        n_cpu_cols = all_stark.table_columns[2]
        if cpu.dim() != 2 or cpu.shape[1] != n_cpu_cols or cpu.shape[0] & (cpu.shape[0] - 1) or cpu.shape[0] == 0:
            raise ZkStarkError(-1, "cpu: expected (2^k, %d) rows" % n_cpu_cols)
        arithmetic, _ = arithmetic_generate_trace(self.arithmetic_ops, device=device, ctx=ctx)
        byte_packing = byte_packing_generate_trace(self.byte_packing_ops, cap_elements, device=device, ctx=ctx)
        cpu_trace = cpu.t().contiguous()                                   # trace_rows_to_poly_values
        keccak = keccak_generate_trace(self.keccak_inputs, cap_elements, device=device, ctx=ctx)
        keccak_sponge = keccak_sponge_generate_trace(self.keccak_sponge_ops, cap_elements, device=device, ctx=ctx)
        logic = logic_generate_trace(self.logic_ops, cap_elements, device=device, ctx=ctx)
        memory, mem_after, final_values, self.unpadded_memory_length = memory_generate_trace(
            self.memory_ops, mem_before_values, stale_contexts, device=device, ctx=ctx, packed_final=packed_final)
        mem_before = memory_continuation_generate_trace(mem_before_values, device=device, ctx=ctx)
        tables = [arithmetic, byte_packing, cpu_trace, keccak, keccak_sponge, logic, memory, mem_before, mem_after]
        if all_stark.cdk_erigon:
            tables.append(poseidon_generate_trace(self.poseidon_ops, cap_elements, device=device, ctx=ctx))
        elif self.poseidon_ops:
            raise ZkStarkError(-1, "Poseidon operations in an eth_mainnet run")
        return tables, final_values
