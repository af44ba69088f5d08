"""CPU: the constraint restatements (oracle/airs.py, and through the quotient parity tests csrc/airs.cuh) against the
independently written shape manifest (oracle/air_manifest.py): number of constraints, their consumer kinds, their order."""
import pytest

from oracle import air_manifest as M
from oracle import airs as A
from oracle import tape as T


def _kinds(air_id):
    ev, n_cols = A.AIRS[air_id]
    tb = T.TapeBuilder(2 * n_cols)
    rec = T.RecordingConsumer()
    ev(tb.inputs[:n_cols], tb.inputs[n_cols:], rec)
    return [k for k, _ in rec.items]


@pytest.mark.parametrize("air_id", sorted(M.TABLES))
def test_kind_sequence_matches_manifest(air_id):
    got, exp = _kinds(air_id), M.expand(M.TABLES[air_id])
    assert len(got) == len(exp), (air_id, len(got), len(exp))
    first = next((i for i, (g, e) in enumerate(zip(got, exp)) if g != e), None)
    assert first is None, "table %d: constraint %d is kind %d, the manifest says %d" % (air_id, first, got[first], exp[first])


def test_totals():
    assert len(M.expand(M.ARITHMETIC)) == M.ARITHMETIC_TOTAL and len(M.expand(M.CPU)) == M.CPU_TOTAL
    assert len(_kinds(8)) == M.CPU_TOTAL and len(_kinds(10)) == 531


def test_degrees_stay_within_the_quotient_degree_factor():
    """constraint_degree() = 3 for every table: no traced constraint exceeds it (the tape tracks polynomial degree)."""
    for air_id, (ev, n_cols) in A.AIRS.items():
        if n_cols is None:
            continue
        tp = T.trace_constraints(ev, n_cols)
        deg = [1] * tp.n_in + [0] * len(tp.consts)
        for c, a, b in tp.ops.tolist():
            deg.append(deg[a] + deg[b] if c == T.MUL else max(deg[a], deg[b]))
        # transition / first / last constraints are multiplied by a degree-1 selector: starky's bound is on the
        # constraint itself
        worst = max([deg[o] for o in tp.outputs.tolist()] or [0])
        assert worst <= 3, (air_id, worst)
        if air_id in (2, 3, 5, 6, 7, 8):            # BytePacking (4) tops out at degree 2
            assert worst == 3, (air_id, worst)
