"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        for name in ("orc_gl_add", "orc_gl_sub", "orc_gl_mul", "orc_gl_pow"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gl_inv.restype = C.c_uint64
        L.orc_gl_inv.argtypes = [C.c_uint64]
        L.orc_gl_root_of_unity.restype = C.c_uint64
        L.orc_gl_root_of_unity.argtypes = [C.c_uint]
        L.orc_gl2_mul.argtypes = [u64p, u64p, u64p]
        L.orc_gl2_inv.argtypes = [u64p, u64p]
        L.orc_poseidon_permute.argtypes = [u64p]
        L.orc_poseidon_permute_fast.argtypes = [u64p]
        L.orc_poseidon_use_fast.argtypes = [C.c_int]
        L.orc_poseidon_perms_per_second.restype = C.c_double
        L.orc_poseidon_perms_per_second.argtypes = [C.c_int, C.c_size_t]
        L.orc_poseidon_hash_no_pad.argtypes = [u64p, C.c_size_t, u64p]
        L.orc_poseidon_hash_or_noop.argtypes = [u64p, C.c_size_t, u64p]
        L.orc_poseidon_two_to_one.argtypes = [u64p, u64p, u64p]
        L.orc_keccak_f1600.argtypes = [u64p]
        L.orc_keccak256.argtypes = [u8p, C.c_size_t, u8p]
        L.orc_keccak25_hash_no_pad.argtypes = [u64p, C.c_size_t, u8p]
        L.orc_keccak25_hash_or_noop.argtypes = [u64p, C.c_size_t, u8p]
        L.orc_keccak25_two_to_one.argtypes = [u8p, u8p, u8p]
        for name in ("orc_fft", "orc_ifft"):
            getattr(L, name).argtypes = [u64p, C.c_uint]
        for name in ("orc_coset_fft", "orc_coset_ifft"):
            getattr(L, name).argtypes = [u64p, C.c_uint, C.c_uint64]
        L.orc_lde.argtypes = [u64p, C.c_uint, C.c_uint, u64p]
        L.orc_eval_poly.restype = C.c_uint64
        L.orc_eval_poly.argtypes = [u64p, C.c_size_t, C.c_uint64]
        L.orc_eval_poly_ext.argtypes = [u64p, C.c_size_t, u64p, u64p]
        L.orc_merkle_num_digests.restype = C.c_size_t
        L.orc_merkle_num_digests.argtypes = [C.c_uint, C.c_uint]
        L.orc_merkle_build.argtypes = [u64p, C.c_uint, C.c_size_t, C.c_uint, C.c_int, u64p]
        L.orc_merkle_prove.argtypes = [u64p, C.c_uint, C.c_uint, C.c_size_t, u64p]
        L.orc_merkle_verify.restype = C.c_int
        L.orc_merkle_verify.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p, C.c_uint, u64p, C.c_int]
        vp = C.c_void_p
        L.orc_commit_values.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                        vp, vp, vp, vp]
        L.orc_commit_coeffs.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                        vp, vp, vp]
        L.orc_num_threads.restype = C.c_int

    # ---- convenience wrappers -------------------------------------------------
    def poseidon_permute(self, state):
        st = np.array(state, dtype=np.uint64)
        assert st.shape == (12,)
        self.lib.orc_poseidon_permute(st)
        return st

    def poseidon_hash_no_pad(self, elems):
        a = np.ascontiguousarray(elems, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.orc_poseidon_hash_no_pad(a if a.size else np.zeros(1, np.uint64), a.size, out)
        return out

    def keccak256(self, data: bytes) -> bytes:
        a = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        self.lib.orc_keccak256(a, len(data), out)
        return out.tobytes()

    def commit_values(self, values, rate_bits=1, cap_height=4, hasher=0, want_leaves=True):
        """values: (n_cols, n) uint64.  Returns dict(coeffs, leaves, digests, cap)."""
        values = np.ascontiguousarray(values, dtype=np.uint64)
        n_cols, n = values.shape
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        if cap_height > log_n + rate_bits:               # plonky2's MerkleTree::new asserts the same; the C code does not
            raise ValueError(f"cap_height {cap_height} exceeds tree height {log_n + rate_bits}")
        N = n << rate_bits
        nd = self.lib.orc_merkle_num_digests(log_n + rate_bits, cap_height)
        coeffs = np.zeros((n_cols, n), dtype=np.uint64)
        leaves = np.zeros((N, n_cols), dtype=np.uint64) if want_leaves else None
        digests = np.zeros((nd, 4), dtype=np.uint64)
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        self.lib.orc_commit_values(values, n_cols, log_n, rate_bits, cap_height, hasher,
                                   coeffs.ctypes.data, leaves.ctypes.data if want_leaves else None,
                                   digests.ctypes.data, cap.ctypes.data)
        return dict(coeffs=coeffs, leaves=leaves, digests=digests, cap=cap)

    def merkle_prove(self, digests, log_leaves, cap_height, idx):
        sib = np.zeros((log_leaves - cap_height, 4), dtype=np.uint64)
        self.lib.orc_merkle_prove(np.ascontiguousarray(digests), log_leaves, cap_height, idx,
                                  sib if sib.size else np.zeros((1, 4), np.uint64))
        return sib


_cached = None


def usable_cores() -> int:
    """Threads this process can actually run: affinity mask and cgroup CPU quota (the GPU box shows 256 CPUs inside a
    16-core quota; 256 OpenMP threads there spend their time being throttled at barriers)."""
    import math
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def load_oracle() -> Oracle:
    global _cached
    if _cached is None:
        if "OMP_NUM_THREADS" not in os.environ:
            os.environ["OMP_NUM_THREADS"] = str(usable_cores())     # read by libgomp when the library is loaded
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so) or os.path.exists(os.path.join(ORACLE_DIR, "Makefile")) and \
                any(os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(so)
                    for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
            build_oracle()
        _cached = Oracle(C.CDLL(so))
        if os.environ.get("OMP_NUM_THREADS", "").isdigit():
            try:                                            # libgomp may have been initialised before the variable was set
                C.CDLL("libgomp.so.1").omp_set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
            except OSError:
                pass
    return _cached


def splitmix64(seed: int, n: int) -> np.ndarray:
    """Deterministic u64 stream (SURVEY 8(d) config 2 input generator)."""
    out = np.empty(n, dtype=np.uint64)
    M = (1 << 64) - 1
    idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
           + np.uint64(seed & M))
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    out[:] = z
    return out


# ---- FRI / challenger helpers (oracle side) ---------------------------------------------------
class OrcCfg(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("rate_bits", "cap_height", "hasher", "num_challenges",
                                          "proof_of_work_bits", "num_query_rounds", "arity_bits",
                                          "final_poly_bits")]


class OrcChallenger(C.Structure):
    _fields_ = [("hasher", C.c_int), ("state", C.c_uint64 * 12), ("inb", C.c_uint64 * 8),
                ("n_in", C.c_int), ("out", C.c_uint64 * 8), ("n_out", C.c_int)]


class OrcBatch(C.Structure):
    _fields_ = [("n_cols", C.c_size_t), ("log_n", C.c_uint), ("coeffs", C.c_void_p),
                ("leaves", C.c_void_p), ("digests", C.c_void_p)]


class OrcFriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("n_polys", C.c_size_t), ("oracle_idx", C.c_void_p),
                ("poly_idx", C.c_void_p)]


def make_cfg(rate_bits=1, cap_height=4, hasher=0, num_challenges=2, pow_bits=16, queries=84,
             arity_bits=4, final_poly_bits=5):
    return OrcCfg(rate_bits, cap_height, hasher, num_challenges, pow_bits, queries, arity_bits,
                  final_poly_bits)


class FriInstance:
    """FriInstanceInfo in flat form: list of (point(2,), [(oracle, poly), ...])."""

    def __init__(self, batches):
        self.batches = batches
        self._keep = []
        arr = (OrcFriBatch * len(batches))()
        for i, (pt, polys) in enumerate(batches):
            oi = np.array([p[0] for p in polys], dtype=np.uint32)
            pi = np.array([p[1] for p in polys], dtype=np.uint32)
            self._keep += [oi, pi]
            arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
            arr[i].n_polys = len(polys)
            arr[i].oracle_idx = oi.ctypes.data
            arr[i].poly_idx = pi.ctypes.data
        self.c = arr

    @property
    def n_openings(self):
        return sum(len(p) for _, p in self.batches)


def stark_fri_instance(zeta, g_zeta, n_trace, n_aux, n_quot, ctl_zs_range=None):
    """[EXT] starky `Stark::fri_instance`: oracles trace(0), aux(1), quotient(2); batches at zeta
    (all), g*zeta (trace+aux) and, if the table has CTLs, 1 (the ctl Z columns of the aux oracle)."""
    trace = [(0, i) for i in range(n_trace)]
    aux = [(1, i) for i in range(n_aux)]
    qo = 2 if n_aux else 1
    quot = [(qo, i) for i in range(n_quot)]
    batches = [(zeta, trace + aux + quot), (g_zeta, trace + aux)]
    if ctl_zs_range is not None:
        batches.append(((1, 0), [(1, i) for i in range(*ctl_zs_range)]))
    return FriInstance(batches)


def setup_fri_api(o):
    L = o.lib
    vp = C.c_void_p
    L.orc_challenger_init.argtypes = [C.POINTER(OrcChallenger), C.c_int]
    L.orc_challenger_observe.argtypes = [C.POINTER(OrcChallenger), u64p, C.c_size_t]
    L.orc_challenger_observe_cap.argtypes = [C.POINTER(OrcChallenger), u64p, C.c_size_t]
    L.orc_challenger_get.restype = C.c_uint64
    L.orc_challenger_get.argtypes = [C.POINTER(OrcChallenger)]
    L.orc_challenger_get_ext.argtypes = [C.POINTER(OrcChallenger), u64p]
    L.orc_challenger_compact.argtypes = [C.POINTER(OrcChallenger), u64p]
    L.orc_fri_reduction_arity_bits.restype = C.c_size_t
    L.orc_fri_reduction_arity_bits.argtypes = [C.c_uint, C.POINTER(OrcCfg), vp, C.c_size_t]
    L.orc_fri_proof_words.restype = C.c_size_t
    L.orc_fri_proof_words.argtypes = [C.POINTER(OrcCfg), C.c_uint, vp, C.c_size_t]
    L.orc_fri_openings.argtypes = [C.POINTER(OrcBatch), C.POINTER(OrcFriBatch), C.c_size_t, u64p]
    L.orc_fri_prove_openings.argtypes = [C.POINTER(OrcCfg), C.c_uint, C.POINTER(OrcBatch), C.c_size_t,
                                         C.POINTER(OrcFriBatch), C.c_size_t, C.POINTER(OrcChallenger), u64p]
    L.orc_fri_verify.restype = C.c_int
    L.orc_fri_verify.argtypes = [C.POINTER(OrcCfg), C.c_uint, vp, C.c_size_t, vp, C.POINTER(OrcFriBatch),
                                 C.c_size_t, u64p, C.POINTER(OrcChallenger), u64p, C.POINTER(C.c_int)]


def oracle_batches(commits):
    """commits: list of dicts from Oracle.commit_values -> (OrcBatch array, keepalive)."""
    arr = (OrcBatch * len(commits))()
    for i, r in enumerate(commits):
        arr[i].n_cols = r["coeffs"].shape[0]
        arr[i].log_n = r["coeffs"].shape[1].bit_length() - 1
        arr[i].coeffs = r["coeffs"].ctypes.data
        arr[i].leaves = r["leaves"].ctypes.data
        arr[i].digests = r["digests"].ctypes.data
    return arr


def new_challenger(o, hasher=0):
    ch = OrcChallenger()
    o.lib.orc_challenger_init(C.byref(ch), hasher)
    return ch


def oracle_fri_prove(o, cfg, degree_bits, commits, inst, ch):
    """Returns (openings (n,2) u64, proof words).  Advances `ch` exactly like the prover does:
    observes the openings, then runs prove_openings."""
    L = o.lib
    ob = oracle_batches(commits)
    opn = np.zeros(2 * inst.n_openings, dtype=np.uint64)
    L.orc_fri_openings(ob, inst.c, len(inst.batches), opn)
    L.orc_challenger_observe(C.byref(ch), opn, opn.size)
    cols = np.array([r["coeffs"].shape[0] for r in commits], dtype=np.uint64)
    nw = L.orc_fri_proof_words(C.byref(cfg), degree_bits, cols.ctypes.data, len(commits))
    proof = np.zeros(nw, dtype=np.uint64)
    L.orc_fri_prove_openings(C.byref(cfg), degree_bits, ob, len(commits), inst.c, len(inst.batches),
                             C.byref(ch), proof)
    return opn, proof


def oracle_fri_verify(o, cfg, degree_bits, commits_caps, cols, inst, opn, proof, ch):
    L = o.lib
    caps = (C.c_void_p * len(commits_caps))(*[c.ctypes.data for c in commits_caps])
    colsa = np.array(cols, dtype=np.uint64)
    L.orc_challenger_observe(C.byref(ch), opn, opn.size)
    why = C.c_int(0)
    ok = L.orc_fri_verify(C.byref(cfg), degree_bits, colsa.ctypes.data, len(cols), caps, inst.c,
                          len(inst.batches), opn, C.byref(ch), proof, C.byref(why))
    return ok, why.value
