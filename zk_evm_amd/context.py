"""``zk_ctx`` wrapper: one per GPU / stream (SURVEY 8(b) threading row)."""
import ctypes as C
import threading

from ._lib import ZkStarkError, load_library


class Context:
    """Owns a ``zk_ctx``.  ``stream`` may be a ``torch.cuda.Stream`` (its ``cuda_stream`` handle is
    used) so the kernels interleave correctly with torch work on the same stream."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.zk_ctx_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise ZkStarkError(rc, f"zk_ctx_create(device={device}) failed: no usable HIP device "
                                   "(this backend has no CPU fallback)")
        self.handle = h
        self.device = int(device)
        self._abort = None
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        raw = getattr(stream, "cuda_stream", stream)
        self.check(self.lib.zk_ctx_set_stream(self.handle, C.c_void_p(raw)))

    def use_torch_current_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device))

    def set_abort_flag(self, flag: "C.c_int | C.c_uint8 | C.c_bool | None"):
        """`flag` is a ctypes c_int -- or a one-byte c_uint8 / c_bool, the layout of the reference's `AtomicBool` -- that
        the caller may set non-zero from another thread (abort_signal, evm_arithmetization/src/prover.rs:56,346-354).
        None clears both forms."""
        self._abort = flag
        if flag is None:
            self.check(self.lib.zk_ctx_set_abort_flag(self.handle, None))
            self.check(self.lib.zk_ctx_set_abort_flag_u8(self.handle, None))
            return
        ptr = C.cast(C.byref(flag), C.c_void_p)
        if C.sizeof(flag) == 1:
            self.check(self.lib.zk_ctx_set_abort_flag_u8(self.handle, ptr))
        else:
            self.check(self.lib.zk_ctx_set_abort_flag(self.handle, ptr))

    def set_plans(self, plans: "str | None"):
        """The ctx's plan table (include/zkstark.h zk_ctx_set_plans): which of two equivalent kernels serves a shape, as a
        string of items ("v20f0=2;b20r1=96x1;T=1;").  None = the initial table (ZK_NTT_SWAP_PLANS or the one compiled in),
        "" = the first implementation everywhere.  Results never depend on it."""
        self.check(self.lib.zk_ctx_set_plans(self.handle, None if plans is None else plans.encode("ascii")))

    def get_plans(self) -> str:
        buf = C.create_string_buffer(4100)
        self.lib.zk_ctx_get_plans(self.handle, buf, len(buf))
        return buf.value.decode("ascii")

    def synchronize(self):
        self.check(self.lib.zk_ctx_synchronize(self.handle))

    def mem_reserve(self, n_bytes: int):
        """Pre-size the ctx's HBM arena (include/zkstark.h zk_ctx_mem_reserve)."""
        self.check(self.lib.zk_ctx_mem_reserve(self.handle, n_bytes))

    def mem_trim(self) -> int:
        """Synchronise and give every idle slab of the arena back to the driver; returns the bytes released."""
        r = C.c_size_t(0)
        self.check(self.lib.zk_ctx_mem_trim(self.handle, C.byref(r)))
        return int(r.value)

    def mem_stats(self) -> dict:
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self.check(self.lib.zk_ctx_mem_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"reserved": int(a.value), "in_use": int(b.value), "peak_in_use": int(c.value)}

    def last_timings(self):
        arr = (C.c_float * 4)()
        self.check(self.lib.zk_ctx_last_timings(self.handle, arr))
        return dict(zip(("ifft", "lde", "leaf_hash", "tree"), [float(x) for x in arr]))

    def commit_totals(self, reset: bool = False) -> dict:
        """Accumulated stage timings (ms) and algorithmic work of every commit since the last reset."""
        ms = (C.c_double * 4)()
        n, lb, lp, nb = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0)
        self.check(self.lib.zk_ctx_commit_totals(self.handle, ms, C.byref(n), C.byref(lb), C.byref(lp), C.byref(nb),
                                                 1 if reset else 0))
        out = dict(zip(("ifft", "lde", "leaf_hash", "tree"), [float(x) for x in ms]))
        out.update(commits=int(n.value), leaf_hash_bytes=float(lb.value), leaf_hash_perms=float(lp.value),
                   ntt_bytes=float(nb.value))
        return out

    def side_commit_totals(self, reset: bool = False) -> dict:
        """The same for the commitments zk_prove_segment ran on the ctx's side lane (overlapped with the main stream)."""
        ms = (C.c_double * 4)()
        n, lb, nb = C.c_uint64(0), C.c_double(0), C.c_double(0)
        self.check(self.lib.zk_ctx_side_commit_totals(self.handle, ms, C.byref(n), C.byref(lb), C.byref(nb), 1 if reset else 0))
        out = dict(zip(("ifft", "lde", "leaf_hash", "tree"), [float(x) for x in ms]))
        out.update(commits=int(n.value), leaf_hash_bytes=float(lb.value), ntt_bytes=float(nb.value))
        return out

    def last_error(self) -> str:
        return self.lib.zk_last_error(self.handle).decode()

    def check(self, rc: int):
        if rc != 0:
            raise ZkStarkError(rc, self.last_error())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.zk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context(device: int = 0) -> Context:
    """The calling THREAD's context for `device`.  A zk_ctx is not thread-safe (arena free list, stream, error string:
    include/zkstark.h), and ctypes releases the GIL during calls, so a process-wide default would let two threads that
    omit `ctx=` race inside one arena.  Workers that want explicit control create their own `Context`
    (zk_evm_amd/scheduler.py does)."""
    d = getattr(_tls, "ctx", None)
    if d is None:
        d = _tls.ctx = {}
    if device not in d:
        d[device] = Context(device)
    return d[device]
