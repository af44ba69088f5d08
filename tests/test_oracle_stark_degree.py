"""CPU: the reference's `test_stark_degree` / `test_stark_low_degree` for every table (e.g. keccak_stark.rs:633-646,
logic.rs:405-417, memory_stark.rs tests, cpu_stark.rs tests; [EXT] starky stark_testing.rs), run on the oracle's
restated constraint systems: random low-degree trace (and auxiliary) polynomials, the alpha-combined constraint value at
every point of a 4x larger coset, interpolated -- its degree must be below constraint_degree * n = 3n, which is what
makes the quotient fit `quotient_degree_factor = 2` chunks.  Covers the table AIR, its range-check lookups and its
cross-table-lookup checks with the real `all_stark.rs` wiring, so a restatement slip that raises a degree (a product
of one column too many) fails here."""
import numpy as np
import pytest

from oracle import airs as oairs
from oracle import all_stark as A
from oracle import segment as oseg
from oracle import stark as S

P = S.P
LOG_N, BLOWUP_BITS, DEGREE = 2, 2, 3


def _interp_eval(vals, w_n, xs):
    """vals[i] = f(w_n^i), deg f < n  ->  [f(x) for x in xs] (naive Lagrange through coefficients)."""
    n = len(vals)
    ninv = pow(n, P - 2, P)
    winv = pow(w_n, P - 2, P)
    coeffs = [sum(v * pow(winv, i * k, P) for i, v in enumerate(vals)) * ninv % P for k in range(n)]
    out = []
    for x in xs:
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % P
        out.append(acc)
    return out


def _degree_of_values(vals, xs):
    """Degree of the unique polynomial of degree < len(xs) through (xs, vals); xs = g * <w_N>."""
    N = len(xs)
    w = xs[1] * pow(xs[0], P - 2, P) % P
    g = xs[0]
    ninv = pow(N, P - 2, P)
    winv = pow(w, P - 2, P)
    coeffs = [sum(v * pow(winv, i * k, P) for i, v in enumerate(vals)) * ninv % P for k in range(N)]
    ginv = pow(g, P - 2, P)
    coeffs = [c * pow(ginv, k, P) % P for k, c in enumerate(coeffs)]
    return max((k for k, c in enumerate(coeffs) if c), default=-1)


@pytest.mark.parametrize("table", range(A.NUM_TABLES))
def test_constraint_degree_is_three(table):
    rng = np.random.default_rng(100 + table)
    n, N = 1 << LOG_N, 1 << (LOG_N + BLOWUP_BITS)
    w_n, w_N = S.root_of_unity(LOG_N), S.root_of_unity(LOG_N + BLOWUP_BITS)
    g = 7
    xs = [g * pow(w_N, j, P) % P for j in range(N)]
    step = N // n
    n_cols = A.TABLE_COLUMNS[table]
    air = oairs.AIRS[A.TABLE_AIR[table]][0]
    lookups = A.build_lookups()[table]
    chal = [S.GrandProductChallenge(int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62))) for _ in range(2)]
    zdatas = oseg.cross_table_lookup_data([None] * A.NUM_TABLES, A.build_ctls(), chal, DEGREE)[table]
    for z in zdatas:
        k = len(z.columns_filters)
        z.n_helpers = -(-k // 2) if k > 1 else 0
    lookup_challenges = [c.beta for c in chal] if lookups else []
    n_lookup = sum(l.num_helper_columns(DEGREE) for l in lookups) * len(lookup_challenges)
    n_aux = n_lookup + sum(z.n_helpers for z in zdatas) + len(zdatas)

    def rand_polys(count):
        return [_interp_eval([int(v) for v in rng.integers(0, P, n, dtype=np.uint64)], w_n, xs) for _ in range(count)]
    trace, aux = rand_polys(n_cols), rand_polys(n_aux)
    alpha = int(rng.integers(1, P, dtype=np.uint64))
    last = pow(w_n, P - 2, P)
    ninv = pow(n, P - 2, P)
    vals = []
    for j, x in enumerate(xs):
        zh = (pow(x, n, P) - 1) % P
        l_first = zh * ninv % P * pow((x - 1) % P, P - 2, P) % P
        l_last = zh * ninv % P * pow((x * w_n - 1) % P, P - 2, P) % P
        cons = S.ConstraintConsumer([alpha], (x - last) % P, l_first, l_last)
        k = (j + step) % N
        lv, nv = [c[j] for c in trace], [c[k] for c in trace]
        alv, anv = [c[j] for c in aux], [c[k] for c in aux]
        air(lv, nv, cons)
        if lookups:
            S.eval_packed_lookups(lookups, lookup_challenges, lv, nv, alv, anv, cons, DEGREE)
        if zdatas:
            S.eval_cross_table_lookup_checks(zdatas, lv, nv, alv, anv, n_lookup, cons, DEGREE)
        vals.append(cons.accs[0])
    deg = _degree_of_values(vals, xs)
    assert 0 <= deg < DEGREE * n, (A.TABLE_NAMES[table] if hasattr(A, "TABLE_NAMES") else table, deg)
    # and the bound is tight for tables with cubic constraints (a vacuous pass -- all-zero values -- is excluded above)
