"""Time the device witness-table generators at production-like sizes.
Prints one JSON object: per generator, rows, host->device bytes of the operation log, the bytes of the table it
replaces, and the wall time of the C-ABI call (log upload + kernels + sync).  Usage: python tools/bench_tracegen.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import zk_evm_amd as zk
    from zk_evm_amd.context import default_context
    ctx = default_context(0)
    lib, h = ctx.lib, ctx.handle
    rng = np.random.default_rng(1)
    res = {}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    # Memory: ~1M-row table (also through the Python binding with packed arrays: same call underneath)
    n_ops = 900_000
    ops = np.zeros((n_ops, 9), dtype=np.uint64)
    ops[:, 0] = rng.integers(0, 2, n_ops) | 2
    ops[:, 1] = np.arange(1, n_ops + 1)
    ops[:, 2] = rng.integers(0, 40, n_ops)
    ops[:, 3] = rng.integers(0, 36, n_ops)
    ops[:, 4] = rng.integers(0, 4000, n_ops)
    ops[:, 5:9] = rng.integers(0, 1 << 63, (n_ops, 4))
    before = np.zeros((50_000, 7), dtype=np.uint64)
    before[:, 0] = 41
    before[:, 1] = rng.integers(0, 36, 50_000)
    before[:, 2] = np.arange(50_000)
    before[:, 3:7] = rng.integers(0, 1 << 63, (50_000, 4))
    out = {}

    def mem():
        gen = C.c_void_p()
        ctx.check(lib.zk_memory_trace_begin(h, ops.ctypes.data, n_ops, before.ctypes.data, before.shape[0], C.byref(gen)))
        log_n = lib.zk_memory_gen_log_n(gen)
        t = torch.empty((30, 1 << log_n), dtype=torch.int64, device="cuda:0")
        k = C.c_size_t()
        ctx.check(lib.zk_memory_trace_finish(h, gen, None, 0, C.c_void_p(t.data_ptr()), 1 << log_n, C.byref(k)))
        out["mem"] = (log_n, k.value)
        lib.zk_memory_gen_free(gen)
    ms = timed(mem)
    log_n, n_after = out["mem"]
    res["memory"] = dict(ops=n_ops + before.shape[0], rows=1 << log_n, mem_after_entries=n_after, log_bytes=ops.nbytes + before.nbytes,
                         table_bytes=30 * 8 << log_n, ms=round(ms, 2))

    # Arithmetic: 2^20 rows, mix of one- and two-row operations
    n_ops = 660_000
    a = np.zeros((n_ops, 18), dtype=np.uint64)
    a[:, 0] = rng.integers(0, 16, n_ops)
    a[:, 2:14] = rng.integers(0, 1 << 63, (n_ops, 12)) * 2 + 1
    sh = (a[:, 0] == 14) | (a[:, 0] == 15) | (a[:, 0] == 13)
    a[sh, 2] = rng.integers(0, 256, int(sh.sum()))
    a[sh, 3:6] = 0
    fp = (a[:, 0] >= 7) & (a[:, 0] <= 9)
    a[fp, 5] >>= np.uint64(4)                      # < BN254 modulus
    a[fp, 9] >>= np.uint64(4)
    t = torch.empty((116, 1 << 20), dtype=torch.int64, device="cuda:0")
    used = C.c_size_t()
    ms = timed(lambda: ctx.check(lib.zk_arithmetic_generate_trace(h, a.ctypes.data, n_ops, 20, C.c_void_p(t.data_ptr()), 1 << 20,
                                                                  C.byref(used))))
    rc_ms = timed(lambda: ctx.check(lib.zk_range_check_columns(h, C.c_void_p(t.data_ptr()), 1 << 20, 116, 20, 18, 96, 114, 115, 65536)))
    res["arithmetic_range_check_only"] = dict(ms=round(rc_ms, 2))
    res["arithmetic"] = dict(ops=n_ops, rows_used=used.value, rows=1 << 20, log_bytes=a.nbytes, table_bytes=116 * 8 << 20, ms=round(ms, 2))
    del t

    # Keccak: 2^20 rows = 43690 permutations
    n_perms = (1 << 20) // 24
    inp = rng.integers(0, 1 << 63, (n_perms, 25)).astype(np.uint64)
    ts = np.arange(n_perms, dtype=np.uint64)
    t = torch.empty((2431, 1 << 20), dtype=torch.int64, device="cuda:0")
    ms = timed(lambda: ctx.check(lib.zk_keccak_generate_trace(h, inp.ctypes.data, ts.ctypes.data, n_perms, 20,
                                                              C.c_void_p(t.data_ptr()), 1 << 20)))
    res["keccak"] = dict(ops=n_perms, rows=1 << 20, log_bytes=inp.nbytes + ts.nbytes, table_bytes=2431 * 8 << 20, ms=round(ms, 2))
    del t
    print(json.dumps(res))


if __name__ == "__main__":
    main()
