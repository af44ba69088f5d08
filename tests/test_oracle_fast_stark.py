"""CPU: the C/OpenMP row loops of the oracle (oracle/stark.c over tapes traced by oracle/tape.py) against the pure-Python
big-int restatements they are traced from (oracle/stark.py, oracle/airs.py), table by table: helper columns, Z columns,
quotient values, and the complete `prove_with_commitment` transcript -- so the fast path, which is what the -m gpu parity
tests compare the HIP library with at 2^12..2^16 rows, is itself pinned to the literal restatement at small sizes."""
import ctypes as C

import numpy as np
import pytest

import tests.oracle_lib as ol
from oracle import airs as oairs
from oracle import all_stark as oas
from oracle import fast_stark as FS
from oracle import segment as oseg
from oracle import stark as S
from oracle import stark_prover as SP
from oracle import tape as T

P = S.P


def test_tape_tracer_matches_python_on_every_air():
    rng = np.random.default_rng(5)
    for air_id, (ev, n_cols) in sorted(oairs.AIRS.items()):
        if n_cols is None:
            continue
        tp = T.trace_constraints(ev, n_cols)
        lv = [int(x) for x in rng.integers(0, P, size=n_cols, dtype=np.uint64)]
        nv = [int(x) for x in rng.integers(0, P, size=n_cols, dtype=np.uint64)]
        rec = T.RecordingConsumer()
        ev(lv, nv, rec)
        assert [k for k, _ in rec.items] == tp.kinds.tolist(), air_id
        assert [c % P for _, c in rec.items] == tp.eval_py(lv + nv), air_id


def test_symbols_refuse_control_flow():
    tb = T.TapeBuilder(2)
    x = tb.inputs[0]
    with pytest.raises(TypeError):
        bool(x)
    with pytest.raises(TypeError):
        x == 1                                                  # noqa: B015
    with pytest.raises(TypeError):
        int(x)


@pytest.mark.parametrize("table", range(9))
def test_fast_table_proof_equals_python_restatement(oracle, table):
    """one table of a nine-table segment at 16-32 rows: CTL data with the real all_stark.rs wiring, lookups, and the
    whole prove_with_commitment (aux cap, quotient cap, openings, FRI) -- C row loops == Python big-int loops."""
    from tests.test_gpu_segment import LOG_N, make_traces
    ol.setup_fri_api(oracle)
    rng = np.random.default_rng(300 + table)
    traces = make_traces(rng)
    reg = oas.Registry(False)
    cfg = ol.make_cfg(pow_bits=2, queries=2)
    chal = [S.GrandProductChallenge(int(rng.integers(1, P, dtype=np.uint64)), int(rng.integers(1, P, dtype=np.uint64)))
            for _ in range(2)]
    per_table = oseg.cross_table_lookup_data(traces, reg.ctls, chal, 3)
    pairs = [(c.beta, c.gamma) for c in chal]
    air = oairs.AIRS[reg.TABLE_AIR[table]][0]
    tr = traces[table]
    commit = oracle.commit_values(tr, rate_bits=1, cap_height=4, hasher=0)
    # column generators
    cols = [[int(x) % P for x in c] for c in tr]
    for l in reg.lookups[table]:
        exp = S.lookup_helper_columns(l, cols, pairs[0][0], 3)
        got = FS.lookup_helper_columns(oracle.lib, l, np.ascontiguousarray(tr), pairs[0][0], 3)
        assert [[int(x) for x in c] for c in got] == exp
    zd = per_table[table][0]
    exp = S.partial_sums(cols, zd.columns_filters, zd.challenge, 3)
    got = FS.partial_sums(oracle.lib, np.ascontiguousarray(tr), zd.columns_filters, zd.challenge, 3)
    assert [[int(x) for x in c] for c in got] == exp
    # whole table proof, identical transcripts
    import copy
    outs = []
    for prove in (SP.prove_with_commitment, FS.prove_with_commitment):
        och = ol.new_challenger(oracle, 0)
        oracle.lib.orc_challenger_observe_cap(C.byref(och), commit["cap"], commit["cap"].shape[0])
        zds = copy.deepcopy(per_table[table])
        p = prove(oracle, ol, cfg, air, tr, commit, reg.lookups[table], zds, pairs, och)
        outs.append((p, oracle.lib.orc_challenger_get(C.byref(och))))
    (a, ca), (b, cb) = outs
    assert ca == cb
    assert (a["aux_cap"] is None) == (b["aux_cap"] is None)
    if a["aux_cap"] is not None:
        assert np.array_equal(a["aux_values"], b["aux_values"])
        assert np.array_equal(a["aux_cap"], b["aux_cap"])
    assert np.array_equal(a["quotient_coeffs"], b["quotient_coeffs"])
    assert np.array_equal(a["quotient_cap"], b["quotient_cap"])
    assert np.array_equal(a["openings"], b["openings"]) and np.array_equal(a["fri"], b["fri"])


def test_non_binary_filter_is_rejected(oracle):
    tr = np.zeros((12, 16), dtype=np.uint64)
    tr[0, 3] = 2
    cf = [(S.Column.singles(range(1, 4)), S.Filter.new_simple(S.Column.single(0)))]
    with pytest.raises(AssertionError):
        FS.partial_sums(oracle.lib, tr, cf, S.GrandProductChallenge(3, 5), 3)
