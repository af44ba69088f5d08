"""Host mirror of starky's lookup / cross-table-lookup description types and of the two
auxiliary-column builders the reference drives (marshalling only):

  * ``Column`` / ``Filter`` / ``Lookup`` / ``TableWithColumns`` mirror starky ``lookup.rs`` /
    ``cross_table_lookup.rs`` ([EXT]); the reference instantiates them in
    evm_arithmetization/src/all_stark.rs:153-417 and each table's ``lookups()`` / ``ctl_*``.
  * ``lookup_helper_columns`` = starky ``lookup_helper_columns`` (reached from prover.rs:322),
    ``ctl_partial_sums`` = starky ``cross_table_lookup::partial_sums`` (reached from prover.rs:137).

The descriptions are encoded into the flat u64 "program" of include/zkstark.h and evaluated on
the GPU; there is no Python arithmetic here.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .context import Context, default_context

P = 0xFFFFFFFF00000001


class Column:
    """starky `Column`: sum_i c_i * local[i] + sum_j d_j * next[j] + constant."""
    # (a plain class: the field `linear_combination` shares its name with the reference's constructor below)
    def __init__(self, linear_combination=None, next_row_linear_combination=None, constant=0):
        self.linear_combination = list(linear_combination) if linear_combination is not None else []
        self.next_row_linear_combination = list(next_row_linear_combination) if next_row_linear_combination is not None else []
        self.constant = constant

    def __repr__(self):
        return "Column(%r, %r, %r)" % (self.linear_combination, self.next_row_linear_combination, self.constant)

    @staticmethod
    def single(c): return Column([(c, 1)])
    @staticmethod
    def singles(cs): return [Column.single(c) for c in cs]
    @staticmethod
    def single_next_row(c): return Column([], [(c, 1)])
    @staticmethod
    def singles_next_row(cs): return [Column.single_next_row(c) for c in cs]
    @staticmethod
    def constant_col(k): return Column([], [], k % P)
    @staticmethod
    def zero(): return Column()
    @staticmethod
    def one(): return Column([], [], 1)
    @staticmethod
    def linear_combination_with_constant(it, constant):
        return Column([(c, f % P) for c, f in it], [], constant % P)
    @staticmethod
    def linear_combination_and_next_row_with_constant(it, nit, constant):
        return Column([(c, f % P) for c, f in it], [(c, f % P) for c, f in nit], constant % P)
    @staticmethod
    def linear_combination(it): return Column.linear_combination_with_constant(it, 0)
    @staticmethod
    def le_bits(cs): return Column.linear_combination([(c, 1 << i) for i, c in enumerate(cs)])
    @staticmethod
    def le_bits_with_constant(cs, k):
        return Column.linear_combination_with_constant([(c, 1 << i) for i, c in enumerate(cs)], k)
    @staticmethod
    def le_bytes(cs): return Column.linear_combination([(c, 1 << (8 * i)) for i, c in enumerate(cs)])
    @staticmethod
    def sum(cs): return Column.linear_combination([(c, 1) for c in cs])

    def encode(self) -> List[int]:
        w = [len(self.linear_combination), len(self.next_row_linear_combination), self.constant % P]
        for c, f in self.linear_combination:
            w += [int(c), int(f) % P]
        for c, f in self.next_row_linear_combination:
            w += [int(c), int(f) % P]
        return w


@dataclass
class Filter:
    products: List[Tuple[Column, Column]] = field(default_factory=list)
    constants: List[Column] = field(default_factory=lambda: [Column.one()])   # Default = always on

    @staticmethod
    def new(products, constants): return Filter(list(products), list(constants))
    @staticmethod
    def new_simple(col): return Filter([], [col])

    def encode(self) -> List[int]:
        w = [len(self.products), len(self.constants)]
        for a, b in self.products:
            w += a.encode() + b.encode()
        for c in self.constants:
            w += c.encode()
        return w


@dataclass
class Lookup:
    columns: List[Column]
    table_column: Column
    frequencies_column: Column
    filter_columns: List[Filter]

    def num_helper_columns(self, constraint_degree: int) -> int:
        return -(-len(self.columns) // (constraint_degree - 1)) + 1


@dataclass
class TableWithColumns:
    table: int
    columns: List[Column]
    filter: Filter


@dataclass
class CrossTableLookup:
    looking_tables: List[TableWithColumns]
    looked_table: TableWithColumns


def encode_program(entries: Sequence[Tuple[Sequence[Column], Filter]], table_column: Column = None,
                   frequencies_column: Column = None) -> np.ndarray:
    """program := n_entries, entry_offset[n], table_off, freq_off, payload (include/zkstark.h)."""
    n = len(entries)
    head = 3 + n
    payload: List[int] = []
    offs = []
    for cols, filt in entries:
        offs.append(head + len(payload))
        payload.append(len(cols))
        for c in cols:
            payload += c.encode()
        payload += filt.encode()
    extra = [0, 0]
    if table_column is not None:
        extra[0] = head + len(payload)
        payload += table_column.encode()
        extra[1] = head + len(payload)
        payload += frequencies_column.encode()
    return np.array([n] + offs + extra + payload, dtype=np.uint64)


def _trace_args(trace):
    import torch
    if not (type(trace).__module__.split(".")[0] == "torch" and trace.is_cuda and trace.dim() == 2):
        raise ZkStarkError(-1, "trace must be a 2-D CUDA tensor (n_cols, n)")
    if trace.dtype not in (torch.int64, torch.uint64) or trace.stride(1) != 1:
        raise ZkStarkError(-1, "trace must be int64/uint64 with contiguous columns")
    n_cols, n = trace.shape
    if n & (n - 1):
        raise ZkStarkError(-1, "trace length must be a power of two")
    return n_cols, n, n.bit_length() - 1, (trace.stride(0) if n_cols > 1 else n)


def _run(fn_name, trace, prog, n_max_out, ctx, *scalars):
    import torch
    n_cols, n, log_n, stride = _trace_args(trace)
    ctx = ctx or default_context(trace.device.index or 0)
    ctx.use_torch_current_stream()
    out = torch.empty((n_max_out, n), dtype=torch.int64, device=trace.device)
    n_out = C.c_size_t(0)
    fn = getattr(ctx.lib, fn_name)
    rc = fn(ctx.handle, C.c_void_p(trace.data_ptr()), stride, n_cols, log_n, prog.ctypes.data, prog.size,
            *scalars, C.c_void_p(out.data_ptr()), n, C.byref(n_out))
    ctx.check(rc)
    return out[: n_out.value]


def lookup_helper_columns(lookup: Lookup, trace, challenge: int, constraint_degree: int, ctx: Context = None):
    """-> CUDA tensor (num_helper_columns, n): helper columns then Z."""
    if len(lookup.columns) != len(lookup.filter_columns):
        raise ZkStarkError(-1, "columns / filter_columns length mismatch")
    prog = encode_program([([c], f) for c, f in zip(lookup.columns, lookup.filter_columns)],
                          lookup.table_column, lookup.frequencies_column)
    return _run("zk_lookup_helper_columns", trace, prog, lookup.num_helper_columns(constraint_degree), ctx,
                C.c_uint64(challenge % (1 << 64)), constraint_degree)


def ctl_partial_sums(trace, columns_filters: Sequence[Tuple[Sequence[Column], Filter]], beta: int, gamma: int,
                     constraint_degree: int, ctx: Context = None):
    """-> CUDA tensor: helper columns (only when more than one entry) then Z."""
    prog = encode_program(columns_filters)
    n_help = -(-len(columns_filters) // (constraint_degree - 1))
    return _run("zk_ctl_partial_sums", trace, prog, n_help + 1, ctx, C.c_uint64(beta % (1 << 64)),
                C.c_uint64(gamma % (1 << 64)), constraint_degree)
