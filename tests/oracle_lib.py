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


def load_oracle() -> Oracle:
    global _cached
    if _cached is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so) or os.path.exists(os.path.join(ORACLE_DIR, "Makefile")) and \
                any(os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(so)
                    for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
            build_oracle()
        _cached = Oracle(C.CDLL(so))
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
