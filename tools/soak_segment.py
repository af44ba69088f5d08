"""Soak: prove the same 2^k nine-table segment repeatedly; the proof must be bit-identical every time (smallest valid
PoW witness => deterministic) and the arena must not grow.  Usage: python tools/soak_segment.py [log_n] [iterations]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(proof):
    h = hashlib.sha256()
    for sp in proof.multi_proof.stark_proofs:
        if sp is None:
            continue
        p = sp.proof
        for a in (p.trace_cap, p.auxiliary_polys_cap, p.quotient_polys_cap, p.openings, p.opening_proof, sp.init_challenger_state):
            if a is not None:
                h.update(a.tobytes())
    return h.hexdigest()


def main():
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from bench import synthetic_segment_traces
    from zk_evm_amd.all_stark import AllStark
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    ctx = zk.Context(0)
    ctx.use_torch_current_stream()
    traces = synthetic_segment_traces([log_n] * 9, dev, seed=3)
    cfg = zk.StarkConfig()
    st = AllStark((1, 2, 3, 4))
    first, peak0, times = None, None, []
    for i in range(iters):
        t0 = time.perf_counter()
        proof = sg.prove_with_traces(st, cfg, traces, [True] * 9, sg.PublicValues(), ctx=ctx)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        d = digest(proof)
        mem = ctx.mem_stats()
        if first is None:
            first = d
        elif i == 1:
            peak0 = mem["reserved"]
        assert d == first, f"iteration {i}: proof differs"
        assert mem["in_use"] == 0, mem
        if peak0 is not None:
            assert mem["reserved"] == peak0, (i, mem, peak0)
    print(json.dumps({"log_n": log_n, "iterations": iters, "proof_sha256": first, "arena_reserved_GB": peak0 / 1e9 if peak0 else None,
                      "s_min": min(times[1:]), "s_max": max(times[1:]), "s_first": times[0]}))


if __name__ == "__main__":
    main()
