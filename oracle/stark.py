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
class Column:
    """sum_i c_i * local[i] + sum_j d_j * next[j] + constant"""
    # (a plain class: the field `linear_combination` shares its name with the reference's constructor below)
    def __init__(self, linear_combination=None, next_row_linear_combination=None, constant=0):
        self.linear_combination = list(linear_combination) if linear_combination is not None else []
        self.next_row_linear_combination = list(next_row_linear_combination) if next_row_linear_combination is not None else []
        self.constant = constant

    def __repr__(self):
        return "Column(%r, %r, %r)" % (self.linear_combination, self.next_row_linear_combination, self.constant)

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


# ---- constraint consumer + checks ([EXT] constraint_consumer.rs, lookup.rs, cross_table_lookup.rs)
class ConstraintConsumer:
    def __init__(self, alphas, z_last, lagrange_first, lagrange_last):
        self.alphas = list(alphas)
        self.z_last, self.lf, self.ll = z_last, lagrange_first, lagrange_last
        self.accs = [0] * len(self.alphas)

    def constraint(self, c):
        c %= P
        self.accs = [(a * al + c) % P for a, al in zip(self.accs, self.alphas)]

    def constraint_transition(self, c): self.constraint(c * self.z_last)
    def constraint_first_row(self, c): self.constraint(c * self.lf)
    def constraint_last_row(self, c): self.constraint(c * self.ll)


def eval_helper_columns(filters, columns_evals, lv, nv, helper_values, constraint_degree, challenge, consumer):
    if not helper_values:
        return
    chunk = constraint_degree - 1
    h = 0
    for s in range(0, len(columns_evals), chunk):
        cols = columns_evals[s:s + chunk]
        fs = filters[s:s + chunk]
        hv = helper_values[h]
        h += 1
        if len(cols) == 2:
            c0, c1 = challenge.combine(cols[0]), challenge.combine(cols[1])
            f0, f1 = fs[0].eval_filter(lv, nv), fs[1].eval_filter(lv, nv)
            consumer.constraint(c1 * c0 * hv - f0 * c1 - f1 * c0)
        elif len(cols) == 1:
            consumer.constraint(challenge.combine(cols[0]) * hv - fs[0].eval_filter(lv, nv))
        else:
            raise NotImplementedError


def eval_packed_lookups(lookups, challenges, lv, nv, aux_lv, aux_nv, consumer, degree):
    start = 0
    for lookup in lookups:
        nh = lookup.num_helper_columns(degree)
        for ch in challenges:
            gc = GrandProductChallenge(1, ch)
            col_evals = [[c.eval_with_next(lv, nv)] for c in lookup.columns]
            eval_helper_columns(lookup.filter_columns, col_evals, lv, nv, aux_lv[start:start + nh - 1], degree, gc, consumer)
            z, next_z = aux_lv[start + nh - 1], aux_nv[start + nh - 1]
            table = (lookup.table_column.eval(lv) + ch) % P
            y = (sum(aux_lv[start:start + nh - 1]) * table - lookup.frequencies_column.eval(lv)) % P
            consumer.constraint_first_row(z)
            consumer.constraint((next_z - z) * table - y)
            start += nh


@dataclass
class CtlZData:
    """One `CtlZData`: challenge, the looking (columns, filter) entries of THIS table, helper count."""
    challenge: GrandProductChallenge
    columns_filters: list
    n_helpers: int


def eval_cross_table_lookup_checks(zdatas, lv, nv, aux_lv, aux_nv, num_lookup_columns, consumer, degree):
    total_helpers = sum(z.n_helpers for z in zdatas)
    start = 0
    for i, zd in enumerate(zdatas):
        helpers = aux_lv[num_lookup_columns + start: num_lookup_columns + start + zd.n_helpers]
        local_z = aux_lv[num_lookup_columns + total_helpers + i]
        next_z = aux_nv[num_lookup_columns + total_helpers + i]
        evals = [[c.eval_with_next(lv, nv) for c in cols] for cols, _ in zd.columns_filters]
        filters = [f for _, f in zd.columns_filters]
        eval_helper_columns(filters, evals, lv, nv, helpers, degree, zd.challenge, consumer)
        if helpers:
            hs = sum(helpers) % P
            consumer.constraint_last_row(local_z - hs)
            consumer.constraint_transition(local_z - next_z - hs)
        elif len(evals) > 1:
            c0, c1 = zd.challenge.combine(evals[0]), zd.challenge.combine(evals[1])
            f0, f1 = filters[0].eval_filter(lv, nv), filters[1].eval_filter(lv, nv)
            consumer.constraint_last_row(c0 * c1 * local_z - f0 * c1 - f1 * c0)
            consumer.constraint_transition(c0 * c1 * (local_z - next_z) - f0 * c1 - f1 * c0)
        else:
            c0 = zd.challenge.combine(evals[0])
            f0 = filters[0].eval_filter(lv, nv)
            consumer.constraint_last_row(c0 * local_z - f0)
            consumer.constraint_transition(c0 * (local_z - next_z) - f0)
        start += zd.n_helpers


G = 14293326489335486720
POW2_GEN = 7277203076849721926


def root_of_unity(log_n):
    return pow(POW2_GEN, 1 << (32 - log_n), P)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def compute_quotient_values(air_eval, lookups, lookup_challenges, zdatas, alphas, degree_bits, rate_bits,
                            constraint_degree, trace_leaves, aux_leaves):
    """[EXT] starky prover.rs `compute_quotient_polys`, up to (excluding) the coset_ifft.
    trace_leaves / aux_leaves: committed leaves (bit-reversed rows, as MerkleTree stores them).
    Returns [num_challenges][size] quotient VALUES on the coset of size n * 2^quotient_degree_bits."""
    n = 1 << degree_bits
    qdf = max(1, constraint_degree - 1)
    qdb = (qdf - 1).bit_length()
    assert qdb <= rate_bits
    step = 1 << (rate_bits - qdb)
    next_step = 1 << qdb
    size = n << qdb
    log_lde = degree_bits + rate_bits
    w_size = root_of_unity(degree_bits + qdb)
    last = inv(root_of_unity(degree_bits))
    num_lookup_columns = sum(l.num_helper_columns(constraint_degree) for l in lookups) * len(lookup_challenges)
    out = [[0] * size for _ in alphas]

    def row(leaves, i):
        return [int(v) for v in leaves[bitrev(i * step, log_lde)]]   # get_lde_values(i, step)

    n_inv = inv(n)
    for i in range(size):
        x = G * pow(w_size, i, P) % P
        zh = (pow(x, n, P) - 1) % P
        lf = zh * n_inv % P * inv(x - 1) % P
        ll = zh * n_inv % P * last % P * inv(x - last) % P
        cons = ConstraintConsumer(alphas, (x - last) % P, lf, ll)
        i_next = (i + next_step) % size
        lv, nv = row(trace_leaves, i), row(trace_leaves, i_next)
        air_eval(lv, nv, cons)
        if aux_leaves is not None:
            alv, anv = row(aux_leaves, i), row(aux_leaves, i_next)
            if lookups:
                eval_packed_lookups(lookups, lookup_challenges, lv, nv, alv, anv, cons, constraint_degree)
            if zdatas:
                eval_cross_table_lookup_checks(zdatas, lv, nv, alv, anv, num_lookup_columns, cons, constraint_degree)
        zinv = inv(zh)
        for k, a in enumerate(cons.accs):
            out[k][i] = a * zinv % P
    return out
