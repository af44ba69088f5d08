"""oracle/stark.py -- TEST INFRASTRUCTURE ONLY (CPU oracle; never imported by the product).

Pure-Python (big-int) restatement of the starky 1.0.0 layer that evm_arithmetization drives
([EXT] starky/src/{lookup.rs, cross_table_lookup.rs, constraint_consumer.rs, prover.rs,
vanishing_poly.rs, proof.rs, get_challenges.rs}; crate pinned at Cargo.lock:4740-4743, not
vendored).  Reference call sites: evm_arithmetization/src/prover.rs:137 (`get_ctl_data`) and
prover.rs:322 (`prove_with_commitment`); the column / filter / lookup / CTL *definitions* it is
fed with live in the reference tree (all_stark.rs:153-417, each table's `lookups()` and `ctl_*`).

Python loops are only for small cases (n <= 2^10); NTT / Merkle / FRI go through the C oracle.
Nothing here is pinned by a reference golden vector ("parity unpinned"); the module
also carries a verifier restatement so that prover <-> verifier consistency is checked.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

P = 0xFFFFFFFF00000001


def inv(a: int) -> int:
    return pow(a % P, P - 2, P)


# ---- Column / Filter ([EXT] starky lookup.rs) -------------------------------------------------
@dataclass
class Column:
    """sum_i c_i * local[i] + sum_j d_j * next[j] + constant"""
    linear_combination: List[Tuple[int, int]] = field(default_factory=list)
    next_row_linear_combination: List[Tuple[int, int]] = field(default_factory=list)
    constant: int = 0

    # constructors used by the reference (SURVEY 8(a'))
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
    def linear_combination_with_constant(it, constant): return Column([(c, f % P) for c, f in it], [], constant % P)
    @staticmethod
    def linear_combination_and_next_row_with_constant(it, nit, constant):
        return Column([(c, f % P) for c, f in it], [(c, f % P) for c, f in nit], constant % P)
    @staticmethod
    def linear_combination(it): return Column.linear_combination_with_constant(it, 0)
    @staticmethod
    def le_bits(cs): return Column.linear_combination([(c, 1 << i) for i, c in enumerate(cs)])
    @staticmethod
    def le_bits_with_constant(cs, k): return Column.linear_combination_with_constant([(c, 1 << i) for i, c in enumerate(cs)], k)
    @staticmethod
    def le_bytes(cs): return Column.linear_combination([(c, 1 << (8 * i)) for i, c in enumerate(cs)])
    @staticmethod
    def sum(cs): return Column.linear_combination([(c, 1) for c in cs])

    def eval_with_next(self, v, nv) -> int:
        r = self.constant
        for c, f in self.linear_combination:
            r += v[c] * f
        for c, f in self.next_row_linear_combination:
            r += nv[c] * f
        return r % P

    def eval(self, v) -> int:
        return self.eval_with_next(v, None) if not self.next_row_linear_combination else \
            (self.constant + sum(v[c] * f for c, f in self.linear_combination)) % P

    def eval_table(self, table, row) -> int:
        """table[c][row]; at the last row the next-row part is dropped (not wrapped)."""
        n = len(table[0])
        r = self.constant + sum(int(table[c][row]) * f for c, f in self.linear_combination)
        if self.next_row_linear_combination and row < n - 1:
            r += sum(int(table[c][row + 1]) * f for c, f in self.next_row_linear_combination)
        return r % P


@dataclass
class Filter:
    """sum (a*b) over products + sum over constants; Default = constant 1 (always on)."""
    products: List[Tuple[Column, Column]] = field(default_factory=list)
    constants: List[Column] = field(default_factory=lambda: [Column.one()])

    @staticmethod
    def new(products, constants): return Filter(list(products), list(constants))
    @staticmethod
    def new_simple(col): return Filter([], [col])

    def eval_filter(self, v, nv) -> int:
        r = sum(a.eval_with_next(v, nv) * b.eval_with_next(v, nv) for a, b in self.products)
        r += sum(c.eval_with_next(v, nv) for c in self.constants)
        return r % P

    def eval_table(self, table, row) -> int:
        r = sum(a.eval_table(table, row) * b.eval_table(table, row) for a, b in self.products)
        r += sum(c.eval_table(table, row) for c in self.constants)
        return r % P


@dataclass
class Lookup:
    columns: List[Column]
    table_column: Column
    frequencies_column: Column
    filter_columns: List[Filter]

    def num_helper_columns(self, constraint_degree: int) -> int:
        return -(-len(self.columns) // (constraint_degree - 1)) + 1


@dataclass
class GrandProductChallenge:
    beta: int
    gamma: int

    def combine(self, terms: Sequence[int]) -> int:
        acc = 0
        for t in reversed(list(terms)):
            acc = (acc * self.beta + t) % P
        return (acc + self.gamma) % P


# ---- helper columns ([EXT] lookup.rs get_helper_cols / lookup_helper_columns) -----------------
def get_helper_cols(trace, degree, columns_filters, challenge: GrandProductChallenge, constraint_degree):
    """columns_filters: list of (list[Column], Filter).  Returns list of helper columns (lists)."""
    chunk = constraint_degree - 1
    helpers = []
    for s in range(0, len(columns_filters), chunk):
        acc = [0] * degree
        for cols, filt in columns_filters[s:s + chunk]:
            for d in range(degree):
                f = filt.eval_table(trace, d)
                if f == 1:
                    v = challenge.combine([c.eval_table(trace, d) for c in cols])
                    acc[d] = (acc[d] + inv(v)) % P
                else:
                    assert f == 0, "Non-binary filter?"
        helpers.append(acc)
    return helpers


def lookup_helper_columns(lookup: Lookup, trace, challenge: int, constraint_degree: int):
    assert constraint_degree in (2, 3)
    degree = len(trace[0])
    cf = [([c], f) for c, f in zip(lookup.columns, lookup.filter_columns)]
    helpers = get_helper_cols(trace, degree, cf, GrandProductChallenge(1, challenge), constraint_degree)
    table_inv = [inv(challenge + lookup.table_column.eval_table(trace, d)) for d in range(degree)]
    freq = [lookup.frequencies_column.eval_table(trace, d) for d in range(degree)]
    z = [0]
    for i in range(degree - 1):
        x = (sum(h[i] for h in helpers) - freq[i] * table_inv[i]) % P
        z.append((z[i] + x) % P)
    return helpers + [z]


# ---- CTL partial sums ([EXT] cross_table_lookup.rs partial_sums) -------------------------------
def partial_sums(trace, columns_filters, challenge: GrandProductChallenge, constraint_degree):
    degree = len(trace[0])
    helpers = get_helper_cols(trace, degree, columns_filters, challenge, constraint_degree)
    z = [0] * degree
    run = 0
    for i in range(degree - 1, -1, -1):
        run = (run + sum(h[i] for h in helpers)) % P
        z[i] = run
    return helpers + [z] if len(columns_filters) > 1 else [z]
