"""oracle/stark_prover.py -- TEST INFRASTRUCTURE ONLY.
Restatement of starky 1.0.0 `prove_with_commitment` ([EXT] starky/src/prover.rs) on top of the C
oracle (NTT / Merkle / FRI through `o`, the tests' ctypes handle) and the Python checks in
oracle/stark.py.  Small sizes only."""
import ctypes as C

import numpy as np

from . import stark as S

P = S.P


def get_ctl_auxiliary(trace, zdatas, constraint_degree):
    """-> (all helper columns of all z-data, all Z columns), updates zdata.n_helpers."""
    helpers, zs = [], []
    for zd in zdatas:
        cols = S.partial_sums(trace, zd.columns_filters, zd.challenge, constraint_degree)
        zd.n_helpers = len(cols) - 1
        helpers += cols[:-1]
        zs.append(cols[-1])
    return helpers, zs


def prove_with_commitment(o, fri_api, cfg, air_eval, trace_values, trace_commit, lookups, zdatas,
                          ctl_challenges, och, constraint_degree=3, requires_ctls=True):
    """o: tests.oracle_lib.Oracle; fri_api: the tests.oracle_lib module (FRI helpers);
    trace_values: (C, n) uint64; trace_commit: o.commit_values(trace_values, ...) result;
    och: OrcChallenger (advanced in place).  Returns a dict mirroring StarkProof."""
    L = o.lib
    n_cols, n = trace_values.shape
    degree_bits = n.bit_length() - 1
    hasher, rate_bits, cap_height = cfg.hasher, cfg.rate_bits, cfg.cap_height
    trace = [[int(x) % P for x in col] for col in trace_values]
    nchal = cfg.num_challenges
    aux_cols = []
    lookup_challenges = []
    if lookups:
        if ctl_challenges is not None:
            lookup_challenges = [b for b, _ in ctl_challenges]
        else:
            lookup_challenges = [L.orc_challenger_get(C.byref(och)) for _ in range(nchal)]
        for l in lookups:
            for ch in lookup_challenges:
                aux_cols += S.lookup_helper_columns(l, trace, ch, constraint_degree)
    if zdatas:
        h, zs = get_ctl_auxiliary(trace, zdatas, constraint_degree)
        aux_cols += h + zs
    aux_commit = None
    if aux_cols:
        aux_vals = np.array(aux_cols, dtype=np.uint64)
        aux_commit = o.commit_values(aux_vals, rate_bits=rate_bits, cap_height=cap_height, hasher=hasher)
        L.orc_challenger_observe_cap(C.byref(och), aux_commit["cap"], aux_commit["cap"].shape[0])
    alphas = [L.orc_challenger_get(C.byref(och)) for _ in range(nchal)]
    qvals = S.compute_quotient_values(air_eval, lookups, lookup_challenges, zdatas, alphas, degree_bits,
                                      rate_bits, constraint_degree, trace_commit["leaves"],
                                      aux_commit["leaves"] if aux_commit else None)
    qdf = max(1, constraint_degree - 1)
    qdb = (qdf - 1).bit_length()
    chunks = []
    for vals in qvals:
        a = np.array(vals, dtype=np.uint64)
        L.orc_coset_ifft(a, degree_bits + qdb, S.G)
        # trim_to_len(degree * quotient_degree_factor) then chunks(degree)
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
    return dict(aux_cap=aux_commit["cap"] if aux_commit else None, quotient_cap=cap, openings=opn, fri=proof,
                aux_values=np.array(aux_cols, dtype=np.uint64) if aux_cols else None, quotient_coeffs=qco,
                alphas=alphas, zeta=zeta, instance=inst, commits=commits)
