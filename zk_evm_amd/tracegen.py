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
    flat = np.zeros((n_ops, 9), dtype=np.uint64)
    for r, (op, a, b) in enumerate(operations):
        if not (0 <= a < (1 << 256) and 0 <= b < (1 << 256)):
            raise ZkStarkError(-1, "logic inputs are U256")
        flat[r] = [op] + [(a >> (64 * l)) & m64 for l in range(4)] + [(b >> (64 * l)) & m64 for l in range(4)]
    out = torch.empty((LOGIC_COLUMNS, 1 << log_n), dtype=torch.int64, device=f"cuda:{device}")
    ctx.check(ctx.lib.zk_logic_generate_trace(ctx.handle, flat.ctypes.data if n_ops else None, n_ops, log_n,
                                              C.c_void_p(out.data_ptr()), 1 << log_n))
    return out
