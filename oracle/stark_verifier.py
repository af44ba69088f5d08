"""oracle/stark_verifier.py -- TEST INFRASTRUCTURE ONLY.
Restatement of starky 1.0.0 `verify_stark_proof_with_challenges` + `get_challenges` ([EXT]
starky/src/verifier.rs, get_challenges.rs) as evm_arithmetization's test verifier drives them
(evm_arithmetization/src/verifier.rs:233-242, with `ignore_trace_cap`).  The constraint side runs in
F_{p^2} (class Ext) on the opened values; the FRI side goes through the C oracle's
verify_fri_proof restatement."""
import ctypes as C

import numpy as np

from . import stark as S

P = S.P


class Ext:
    """a + b*X, X^2 = 7; interoperates with plain ints so the AIR restatements run unchanged."""
    __slots__ = ("a", "b")

    def __init__(self, a=0, b=0):
        self.a, self.b = a % P, b % P

    @staticmethod
    def lift(x):
        return x if isinstance(x, Ext) else Ext(int(x), 0)

    def __add__(self, o): o = Ext.lift(o); return Ext(self.a + o.a, self.b + o.b)
    __radd__ = __add__
    def __sub__(self, o): o = Ext.lift(o); return Ext(self.a - o.a, self.b - o.b)
    def __rsub__(self, o): return Ext.lift(o) - self
    def __neg__(self): return Ext(-self.a, -self.b)
    def __mul__(self, o):
        o = Ext.lift(o)
        return Ext(self.a * o.a + 7 * self.b * o.b, self.a * o.b + self.b * o.a)
    __rmul__ = __mul__
    def __mod__(self, m): return self
    def __eq__(self, o): o = Ext.lift(o); return self.a == o.a and self.b == o.b
    def __hash__(self): return hash((self.a, self.b))
    def inv(self):
        n = S.inv(self.a * self.a - 7 * self.b * self.b)
        return Ext(self.a * n, -self.b * n)
    def pow(self, e):
        r, b = Ext(1), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r
    def __repr__(self): return f"Ext({self.a},{self.b})"


def verify_stark_proof(o, fri_api, cfg, air_eval, n_cols, degree_bits, lookups, zdatas, ctl_challenges, proof, och,
                       constraint_degree=3, requires_ctls=True, check_identity=True):
    """proof: dict(trace_cap, aux_cap, quotient_cap, openings (flat u64), fri).  zdatas: oracle CtlZData
    with n_helpers filled in.  och: challenger in the state `prove_with_commitment` started from.
    Returns (ok, reason)."""
    L = o.lib
    nchal = cfg.num_challenges
    lookup_challenges = [b for b, _ in ctl_challenges] if lookups else []
    n_lookup = sum(l.num_helper_columns(constraint_degree) for l in lookups) * len(lookup_challenges)
    n_help = sum(z.n_helpers for z in zdatas)
    n_aux = n_lookup + n_help + len(zdatas)
    qdf = max(1, constraint_degree - 1)
    n_quot = nchal * qdf
    # ---- get_challenges (trace cap ignored: observed up-front by the caller) ----
    if proof["aux_cap"] is not None:
        L.orc_challenger_observe_cap(C.byref(och), proof["aux_cap"], proof["aux_cap"].shape[0])
    alphas = [L.orc_challenger_get(C.byref(och)) for _ in range(nchal)]
    L.orc_challenger_observe_cap(C.byref(och), proof["quotient_cap"], proof["quotient_cap"].shape[0])
    z2 = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(och), z2)
    zeta = Ext(int(z2[0]), int(z2[1]))
    opn = np.ascontiguousarray(proof["openings"], dtype=np.uint64).reshape(-1)
    vals = [Ext(int(opn[2 * i]), int(opn[2 * i + 1])) for i in range(opn.size // 2)]
    exp_len = (n_cols + n_aux + n_quot) + (n_cols + n_aux) + (len(zdatas) if requires_ctls and zdatas else 0)
    if len(vals) != exp_len:
        return False, "opening set shape"
    local = vals[:n_cols]
    aux_local = vals[n_cols:n_cols + n_aux]
    quot = vals[n_cols + n_aux:n_cols + n_aux + n_quot]
    off = n_cols + n_aux + n_quot
    nxt = vals[off:off + n_cols]
    aux_next = vals[off + n_cols:off + n_cols + n_aux]
    # ---- vanishing polynomial at zeta ----
    n = 1 << degree_bits
    g = S.root_of_unity(degree_bits)
    last = S.inv(g)
    zeta_n = zeta.pow(n)
    z_h = zeta_n - 1
    n_inv = S.inv(n)
    l_0 = z_h * n_inv * (zeta - 1).inv()
    l_last = z_h * n_inv * (zeta * g - 1).inv()
    cons = S.ConstraintConsumer(alphas, zeta - last, l_0, l_last)
    cons.accs = [Ext(0) for _ in alphas]
    # check_identity=False: only the commitment / opening / FRI part (used for proofs of random, non-satisfying
    # traces at full size, whose quotient is not a polynomial multiple of Z_H by construction)
    if check_identity:
        air_eval(local, nxt, cons)
        if lookups:
            S.eval_packed_lookups(lookups, lookup_challenges, local, nxt, aux_local, aux_next, cons, constraint_degree)
        if zdatas:
            S.eval_cross_table_lookup_checks(zdatas, local, nxt, aux_local, aux_next, n_lookup, cons, constraint_degree)
    for i, acc in enumerate(cons.accs if check_identity else []):
        chunk = quot[i * qdf:(i + 1) * qdf]
        red = Ext(0)
        for c in reversed(chunk):
            red = red * zeta_n + c
        if not (Ext.lift(acc) == z_h * red):
            return False, "quotient identity"
    # ---- FRI ----
    gz = (zeta.a * g % P, zeta.b * g % P)
    ctl_range = (n_aux - len(zdatas), n_aux) if (requires_ctls and zdatas) else None
    inst = fri_api.stark_fri_instance((zeta.a, zeta.b), gz, n_cols, n_aux, n_quot, ctl_zs_range=ctl_range)
    caps = [proof["trace_cap"]] + ([proof["aux_cap"]] if proof["aux_cap"] is not None else []) + [proof["quotient_cap"]]
    cols = [n_cols] + ([n_aux] if n_aux else []) + [n_quot]
    ok, why = fri_api.oracle_fri_verify(o, cfg, degree_bits, caps, cols, inst, opn.copy(), proof["fri"], och)
    return (ok == 1), f"fri:{why}"
