#!/usr/bin/env python3
"""Prove one synthetic recursion-config PLONK circuit (bench.py's: all fourteen gate kinds) `reps` times (for rocprofv3 --kernel-trace --stats: kernel time per
proof against wall time per proof = how launch- / sync-bound the small proofs are).
Usage: plonk_trace.py [log_n=13] [reps=20] [workers=1]   (workers > 1: that many proofs in flight, one thread + ctx + stream each)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import zk_evm_amd
import zk_evm_amd.plonk as zp

P = 0xFFFFFFFF00000001


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    ctx = zk_evm_amd.Context(0)
    from tools import bench_secondary as bench
    gates, k_is = bench.PLONK_RECURSION_GATES, bench.PLONK_K_IS
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    cs, wires = bench.plonk_synthetic_circuit(dev, lb, g)
    cd = zp.CircuitData(zp.CircuitConfig(), lb, gates, 4, cs, k_is, [1, 2, 3, 4], 123, ctx=ctx)
    pis = [5, 6, 7]
    pr = cd.prove(wires, pis)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pr = cd.prove(wires, pis)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print("log_n %d: %.3f ms per proof; stages %s" % (lb, 1e3 * el, {k: round(v, 3) for k, v in pr.stage_ms.items()}))
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if W > 1:
        import threading
        bar = threading.Barrier(W + 1)

        def worker():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                c2 = zk_evm_amd.Context(0)
                d2 = zp.CircuitData(zp.CircuitConfig(), lb, gates, 4, cs, k_is, [1, 2, 3, 4], 123, ctx=c2)
                d2.prove(wires, pis)
                bar.wait()
                for _ in range(reps):
                    d2.prove(wires, pis)
                st.synchronize()
                bar.wait()
                d2.free()
                c2.close()
        th = [threading.Thread(target=worker) for _ in range(W)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        el = time.perf_counter() - t0
        for t in th:
            t.join()
        print("%d in flight: %.1f proofs/s (%.3f ms per proof effective)" % (W, W * reps / el, 1e3 * el / (W * reps)))


if __name__ == "__main__":
    main()
