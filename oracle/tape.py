"""oracle/tape.py -- TEST INFRASTRUCTURE ONLY (CPU oracle; never imported by the product).

Symbolic tracing of the Python restatements into straight-line field programs ("tapes") that oracle/stark.c runs
over every row / coset point with OpenMP.  The big-int Python code in oracle/airs.py (each table's
`eval_packed_generic`, constraint for constraint in the reference's order) and oracle/stark.py (starky's
`eval_packed_lookups_generic`, `eval_cross_table_lookup_checks`, `Column` / `Filter` evaluation) stays the single
statement of the algorithm: it is *executed once on symbols* instead of once per row on integers.  A constraint's
value is a field element, so any evaluation order of the same expression gives the same word; the yield ORDER and
kind (plain / transition / first row / last row) of the constraints is recorded as is.

Tape := inputs [0, n_in), constants [n_in, n_in + n_consts), then one node per op (op, a, b): 0 ADD, 1 SUB, 2 MUL.
"""
import numpy as np

P = 0xFFFFFFFF00000001
ADD, SUB, MUL = 0, 1, 2
KIND_PLAIN, KIND_TRANSITION, KIND_FIRST, KIND_LAST = 0, 1, 2, 3


class Sym:
    """A node of the tape under construction.  Supports exactly the arithmetic the restatements use on row values:
    + - * with Syms and Python ints, unary minus, `% P` (a no-op: nodes are field elements)."""
    __slots__ = ("b", "i")

    def __init__(self, builder, idx):
        self.b, self.i = builder, idx

    def _o(self, other):
        if isinstance(other, Sym):
            return other
        if isinstance(other, (int, np.integer)):
            return self.b.const(int(other))
        return None

    def __add__(self, o):
        o = self._o(o)
        return NotImplemented if o is None else self.b.op(ADD, self, o)
    __radd__ = __add__

    def __sub__(self, o):
        o = self._o(o)
        return NotImplemented if o is None else self.b.op(SUB, self, o)

    def __rsub__(self, o):
        o = self._o(o)
        return NotImplemented if o is None else self.b.op(SUB, o, self)

    def __mul__(self, o):
        o = self._o(o)
        return NotImplemented if o is None else self.b.op(MUL, self, o)
    __rmul__ = __mul__

    def __neg__(self):
        return self.b.op(SUB, self.b.const(0), self)

    def __mod__(self, m):
        assert m == P
        return self

    def __bool__(self):
        raise TypeError("control flow on a row value: the restatement is not a polynomial program")

    def __index__(self):
        raise TypeError("row value used as an integer")
    __int__ = __index__
    __eq__ = None            # comparisons on row values would silently pick a branch
    __hash__ = None


class TapeBuilder:
    def __init__(self, n_in):
        self.n_in = n_in
        self.consts = []            # values
        self.const_ix = {}          # value -> node
        self.ops = []               # (op, a, b) on provisional node ids
        self.cse = {}
        self.inputs = [Sym(self, i) for i in range(n_in)]
        # provisional ids: inputs 0..n_in-1, constants -(k+1), ops n_in + k  (constants are renumbered in finish())

    def const(self, v):
        v %= P
        s = self.const_ix.get(v)
        if s is None:
            self.consts.append(v)
            s = self.const_ix[v] = Sym(self, -len(self.consts))
        return s

    def _cval(self, s):
        return self.consts[-s.i - 1] if s.i < 0 else None

    def op(self, code, a, b):
        ca, cb = self._cval(a), self._cval(b)
        if ca is not None and cb is not None:        # constant folding
            return self.const(ca + cb if code == ADD else ca - cb if code == SUB else ca * cb)
        if code == ADD:
            if ca == 0:
                return b
            if cb == 0:
                return a
        elif code == SUB:
            if cb == 0:
                return a
        else:
            if ca == 0 or cb == 0:
                return self.const(0)
            if ca == 1:
                return b
            if cb == 1:
                return a
        x, y = a.i, b.i
        if code != SUB and x > y:
            x, y = y, x
        key = (code, x, y)
        s = self.cse.get(key)
        if s is None:
            self.ops.append(key)
            s = self.cse[key] = Sym(self, self.n_in + len(self.ops) - 1)
        return s

    def finish(self, outputs):
        """-> Tape with the given output Syms (ints are allowed: constant outputs)."""
        outs = [o if isinstance(o, Sym) else self.const(int(o)) for o in outputs]
        nc = len(self.consts)

        def fix(i):
            return self.n_in + (-i - 1) if i < 0 else (i if i < self.n_in else i + nc)
        ops = np.array([(c, fix(a), fix(b)) for c, a, b in self.ops], dtype=np.uint32).reshape(-1, 3)
        return Tape(self.n_in, np.array(self.consts, dtype=np.uint64), ops,
                    np.array([fix(o.i) for o in outs], dtype=np.uint32))


class Tape:
    def __init__(self, n_in, consts, ops, outputs):
        self.n_in, self.consts, self.ops, self.outputs = n_in, consts, np.ascontiguousarray(ops), outputs
        self.kinds = None

    @property
    def n_nodes(self):
        return self.n_in + len(self.consts) + len(self.ops)

    def eval_py(self, inputs):
        """reference interpreter (Python ints) -- used by the tests to pin the C executor"""
        v = [int(x) % P for x in inputs] + [int(c) for c in self.consts]
        for c, a, b in self.ops.tolist():
            x, y = v[a], v[b]
            v.append((x + y) % P if c == ADD else (x - y) % P if c == SUB else x * y % P)
        return [v[o] for o in self.outputs.tolist()]


class RecordingConsumer:
    """Stands in for oracle/stark.py ConstraintConsumer while tracing: records (kind, expression) in yield order."""

    def __init__(self):
        self.items = []

    def constraint(self, c): self.items.append((KIND_PLAIN, c))
    def constraint_transition(self, c): self.items.append((KIND_TRANSITION, c))
    def constraint_first_row(self, c): self.items.append((KIND_FIRST, c))
    def constraint_last_row(self, c): self.items.append((KIND_LAST, c))


def trace_constraints(air_eval, n_cols, lookups=(), lookup_challenges=(), zdatas=(), n_aux=0, constraint_degree=3):
    """The whole vanishing-polynomial evaluation of one table at one point ([EXT] starky vanishing_poly.rs
    `eval_vanishing_poly`): table AIR, then lookup checks, then CTL checks.  Inputs: lv[n_cols], nv[n_cols],
    aux_lv[n_aux], aux_nv[n_aux].  -> Tape whose outputs are the constraint values in yield order, with .kinds."""
    from . import stark as S
    tb = TapeBuilder(2 * n_cols + 2 * n_aux)
    lv, nv = tb.inputs[:n_cols], tb.inputs[n_cols:2 * n_cols]
    alv, anv = tb.inputs[2 * n_cols:2 * n_cols + n_aux], tb.inputs[2 * n_cols + n_aux:]
    cons = RecordingConsumer()
    air_eval(lv, nv, cons)
    if n_aux:
        if lookups:
            S.eval_packed_lookups(lookups, lookup_challenges, lv, nv, alv, anv, cons, constraint_degree)
        if zdatas:
            nlc = sum(l.num_helper_columns(constraint_degree) for l in lookups) * len(lookup_challenges)
            S.eval_cross_table_lookup_checks(zdatas, lv, nv, alv, anv, nlc, cons, constraint_degree)
    t = tb.finish([c for _, c in cons.items])
    t.kinds = np.array([k for k, _ in cons.items], dtype=np.uint32)
    return t


def trace_entries(n_cols, columns_filters, challenge):
    """Per-row (filter value, combined value) of each looking entry ([EXT] lookup.rs `get_helper_cols`): inputs
    lv[n_cols], nv[n_cols]; outputs f_0, v_0, f_1, v_1, ...  At the last row the caller feeds nv = 0 (the next-row
    part of a Column is dropped there, not wrapped: `Column::eval_table`)."""
    tb = TapeBuilder(2 * n_cols)
    lv, nv = tb.inputs[:n_cols], tb.inputs[n_cols:]
    outs = []
    for cols, filt in columns_filters:
        outs.append(filt.eval_filter(lv, nv))
        outs.append(challenge.combine([c.eval_with_next(lv, nv) for c in cols]))
    return tb.finish(outs)


def trace_columns(n_cols, columns):
    """values of `Column`s per row (lookup table / frequency columns)."""
    tb = TapeBuilder(2 * n_cols)
    lv, nv = tb.inputs[:n_cols], tb.inputs[n_cols:]
    return tb.finish([c.eval_with_next(lv, nv) for c in columns])
