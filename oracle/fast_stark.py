"""oracle/fast_stark.py -- TEST INFRASTRUCTURE ONLY (CPU oracle; never imported by the product).

The same functions as oracle/stark.py / oracle/stark_prover.py (starky 1.0.0 `lookup_helper_columns`,
`partial_sums`, `compute_quotient_polys`, `prove_with_commitment`; reference call sites prover.rs:137,322) with the
per-row loops run by oracle/stark.c over tapes traced from those very Python restatements (oracle/tape.py).  Nothing
is restated a second time here: this module only moves the loop over rows from Python to C/OpenMP, which is what lets
the word-for-word parity tests run at 2^12..2^16 rows for every table and lets bench.py time a whole table proof on
the CPU.  tests/test_oracle_fast_stark.py pins it to the pure-Python path at small sizes."""
import ctypes as C

import numpy as np

from . import stark as S
from . import tape as T

P = S.P
_vp = C.c_void_p


def _setup(L):
    if getattr(L, "_fast_stark_ready", False):
        return
    L.orc_tape_rows.restype = C.c_int
    L.orc_tape_rows.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp, _vp, C.c_int, C.c_size_t, _vp,
                                C.c_size_t, _vp]
    L.orc_masked_inverse_accumulate.restype = C.c_int
    L.orc_masked_inverse_accumulate.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.orc_lookup_z.argtypes = [_vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]
    L.orc_ctl_z.argtypes = [_vp, C.c_size_t, C.c_size_t, _vp]
    L.orc_quotient_values.restype = C.c_int
    L.orc_quotient_values.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp,
                                      C.c_size_t, C.c_uint, C.c_uint, C.c_uint, _vp, C.c_size_t, _vp]
    L._fast_stark_ready = True


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def tape_rows(L, tape, trace, wrap=False):
    """trace: (C, n) uint64 column-major.  Tape inputs are lv[C], nv[C].  -> (n_out, n) uint64"""
    _setup(L)
    n_cols, n = trace.shape
    assert tape.n_in == 2 * n_cols
    cols = [trace[c] for c in range(n_cols)]
    ins = _ptrs(cols + cols)
    offs = np.array([0] * n_cols + [1] * n_cols, dtype=np.int64)
    out = np.zeros((len(tape.outputs), n), dtype=np.uint64)
    outs = _ptrs([out[k] for k in range(out.shape[0])])
    ops = tape.ops if tape.ops.size else np.zeros((1, 3), dtype=np.uint32)
    consts = tape.consts if tape.consts.size else np.zeros(1, dtype=np.uint64)
    rc = L.orc_tape_rows(ops.ctypes.data, len(tape.ops), consts.ctypes.data, len(tape.consts), tape.n_in, ins,
                         offs.ctypes.data, 1 if wrap else 0, n, tape.outputs.ctypes.data, len(tape.outputs), outs)
    assert rc == 0
    return out


def get_helper_cols(L, trace, columns_filters, challenge, constraint_degree):
    """[EXT] lookup.rs `get_helper_cols`: one column per chunk of (constraint_degree - 1) entries."""
    _setup(L)
    n = trace.shape[1]
    chunk = constraint_degree - 1
    helpers = []
    for s in range(0, len(columns_filters), chunk):
        part = columns_filters[s:s + chunk]
        fv = tape_rows(L, T.trace_entries(trace.shape[0], part, challenge), trace)
        acc = np.zeros(n, dtype=np.uint64)
        for e in range(len(part)):
            f, v = np.ascontiguousarray(fv[2 * e]), np.ascontiguousarray(fv[2 * e + 1])
            if L.orc_masked_inverse_accumulate(f.ctypes.data, v.ctypes.data, n, acc.ctypes.data) != 0:
                raise AssertionError("Non-binary filter?")
        helpers.append(acc)
    return helpers


def lookup_helper_columns(L, lookup, trace, challenge, constraint_degree):
    assert constraint_degree in (2, 3)
    n = trace.shape[1]
    cf = [([c], f) for c, f in zip(lookup.columns, lookup.filter_columns)]
    helpers = get_helper_cols(L, trace, cf, S.GrandProductChallenge(1, challenge), constraint_degree)
    tf = tape_rows(L, T.trace_columns(trace.shape[0], [lookup.table_column, lookup.frequencies_column]), trace)
    den = (tf[0].astype(object) + challenge) % P
    den = np.array(den, dtype=np.uint64)
    z = np.zeros(n, dtype=np.uint64)
    L.orc_lookup_z(_ptrs(helpers), len(helpers), np.ascontiguousarray(tf[1]).ctypes.data, den.ctypes.data, n,
                   z.ctypes.data)
    return helpers + [z]


def partial_sums(L, trace, columns_filters, challenge, constraint_degree):
    n = trace.shape[1]
    helpers = get_helper_cols(L, trace, columns_filters, challenge, constraint_degree)
    z = np.zeros(n, dtype=np.uint64)
    L.orc_ctl_z(_ptrs(helpers), len(helpers), n, z.ctypes.data)
    return helpers + [z] if len(columns_filters) > 1 else [z]


def compute_quotient_values(L, air_eval, n_cols, lookups, lookup_challenges, zdatas, alphas, degree_bits, rate_bits,
                            constraint_degree, trace_leaves, aux_leaves):
    """-> (num_challenges, n << qdb) uint64: quotient VALUES on the coset (before the coset_ifft)."""
    _setup(L)
    n = 1 << degree_bits
    qdf = max(1, constraint_degree - 1)
    qdb = (qdf - 1).bit_length()
    assert qdb <= rate_bits
    n_aux = 0 if aux_leaves is None else aux_leaves.shape[1]
    tp = T.trace_constraints(air_eval, n_cols, lookups, lookup_challenges, zdatas, n_aux, constraint_degree)
    out = np.zeros((len(alphas), n << qdb), dtype=np.uint64)
    al = np.array([a % P for a in alphas], dtype=np.uint64)
    ops = tp.ops if tp.ops.size else np.zeros((1, 3), dtype=np.uint32)
    consts = tp.consts if tp.consts.size else np.zeros(1, dtype=np.uint64)
    tl = np.ascontiguousarray(trace_leaves, dtype=np.uint64)
    axl = np.ascontiguousarray(aux_leaves, dtype=np.uint64) if n_aux else None
    rc = L.orc_quotient_values(ops.ctypes.data, len(tp.ops), consts.ctypes.data, len(tp.consts),
                               tp.outputs.ctypes.data, tp.kinds.ctypes.data, len(tp.outputs), tl.ctypes.data, n_cols,
                               axl.ctypes.data if n_aux else None, n_aux, degree_bits, rate_bits, qdb, al.ctypes.data,
                               len(al), _ptrs([out[k] for k in range(out.shape[0])]))
    assert rc == 0
    return out


def prove_with_commitment(o, fri_api, cfg, air_eval, trace_values, trace_commit, lookups, zdatas, ctl_challenges, och,
                          constraint_degree=3, requires_ctls=True, ctl_columns=None, timing=None):
    """oracle/stark_prover.py `prove_with_commitment` (same transcript, same outputs) with C row loops.
    ctl_columns: optional precomputed [helpers..., z] per z-data (from cross_table_lookup_data above).
    timing: optional dict receiving seconds per stage."""
    import time
    L = o.lib
    _setup(L)
    trace_values = np.ascontiguousarray(trace_values, dtype=np.uint64)
    n_cols, n = trace_values.shape
    degree_bits = n.bit_length() - 1
    hasher, rate_bits, cap_height = cfg.hasher, cfg.rate_bits, cfg.cap_height
    nchal = cfg.num_challenges
    t0 = time.perf_counter()

    def lap(key):
        nonlocal t0
        if timing is not None:
            t1 = time.perf_counter()
            timing[key] = timing.get(key, 0.0) + (t1 - t0)
            t0 = t1
    aux_cols, lookup_challenges = [], []
    if lookups:
        if ctl_challenges is not None:
            lookup_challenges = [b for b, _ in ctl_challenges]
        else:
            lookup_challenges = [L.orc_challenger_get(C.byref(och)) for _ in range(nchal)]
        for l in lookups:
            for ch in lookup_challenges:
                aux_cols += lookup_helper_columns(L, l, trace_values, ch, constraint_degree)
    lap("lookup helper columns")
    if zdatas:
        helpers, zs = [], []
        for i, zd in enumerate(zdatas):
            cols = ctl_columns[i] if ctl_columns is not None else \
                partial_sums(L, trace_values, zd.columns_filters, zd.challenge, constraint_degree)
            zd.n_helpers = len(cols) - 1
            helpers += cols[:-1]
            zs.append(cols[-1])
        aux_cols += helpers + zs
    lap("ctl columns")
    aux_commit = None
    if aux_cols:
        aux_vals = np.ascontiguousarray(np.stack(aux_cols), dtype=np.uint64)
        aux_commit = o.commit_values(aux_vals, rate_bits=rate_bits, cap_height=cap_height, hasher=hasher)
        L.orc_challenger_observe_cap(C.byref(och), aux_commit["cap"], aux_commit["cap"].shape[0])
    lap("auxiliary commitment")
    alphas = [L.orc_challenger_get(C.byref(och)) for _ in range(nchal)]
    qvals = compute_quotient_values(L, air_eval, n_cols, lookups, lookup_challenges, zdatas, alphas, degree_bits,
                                    rate_bits, constraint_degree, trace_commit["leaves"],
                                    aux_commit["leaves"] if aux_commit else None)
    lap("quotient values")
    qdf = max(1, constraint_degree - 1)
    qdb = (qdf - 1).bit_length()
    chunks = []
    for a in qvals:
        a = np.ascontiguousarray(a)
        L.orc_coset_ifft(a, degree_bits + qdb, S.G)
        for j in range(qdf):
            chunks.append(a[j * n:(j + 1) * n].copy())
    qco = np.stack(chunks)
    N = n << rate_bits
    leaves = np.zeros((N, qco.shape[0]), dtype=np.uint64)
    nd = L.orc_merkle_num_digests(degree_bits + rate_bits, cap_height)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    L.orc_commit_coeffs(np.ascontiguousarray(qco), qco.shape[0], degree_bits, rate_bits, cap_height, hasher,
                        leaves.ctypes.data, digests.ctypes.data, cap.ctypes.data)
    q_commit = dict(coeffs=qco, leaves=leaves, digests=digests, cap=cap)
    L.orc_challenger_observe_cap(C.byref(och), cap, cap.shape[0])
    lap("quotient commitment")
    zeta = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(och), zeta)
    zeta = (int(zeta[0]), int(zeta[1]))
    g = S.root_of_unity(degree_bits)
    gz = (zeta[0] * g % P, zeta[1] * g % P)
    n_aux = len(aux_cols)
    n_ctl_zs = len(zdatas)
    ctl_range = (n_aux - n_ctl_zs, n_aux) if (requires_ctls and n_ctl_zs) else None
    inst = fri_api.stark_fri_instance(zeta, gz, n_cols, n_aux, qco.shape[0], ctl_zs_range=ctl_range)
    commits = [trace_commit] + ([aux_commit] if aux_commit else []) + [q_commit]
    opn, proof = fri_api.oracle_fri_prove(o, cfg, degree_bits, commits, inst, och)
    lap("openings + FRI")
    return dict(aux_cap=aux_commit["cap"] if aux_commit else None, quotient_cap=cap, openings=opn, fri=proof,
                aux_values=np.stack(aux_cols) if aux_cols else None, quotient_coeffs=qco,
                alphas=alphas, zeta=zeta, instance=inst, commits=commits)
