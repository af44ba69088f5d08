"""Randomised differential parity: replay every per-table case of tests/test_gpu_stark_prove.py (all eleven AIRs with
their lookups / CTL shapes) at randomly drawn heights, hashers, FRI shapes and seeds, device prover against the oracle
prover word for word.  The pinned tests fix one (height, seed) per table; this walks the neighbourhood.
Usage: python -m tests.fuzz_parity [seconds] [seed] [segment|tracegen|plonk]   (GPU box; prints one JSON line)"""
import inspect
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    import tests.test_gpu_stark_prove as t
    from tests.oracle_lib import load_oracle
    oracle = load_oracle()
    drawn = []

    def draw(n_cols, log_n, hasher, seed, kw):
        # keep the oracle's pure-Python prover to about a second per case: cells = n_cols << log_n
        hi = max(4, min(10, int(np.log2(max(1, 150_000 // n_cols)))))
        c = (int(rng.integers(4, hi + 1)), int(rng.integers(0, 2)), int(rng.integers(1, 1 << 30)),
             dict(pow_bits=int(rng.integers(0, 9)), queries=int(rng.integers(1, 7))))
        drawn.append(c)
        return c

    t.FUZZ = draw
    if len(sys.argv) > 3 and sys.argv[3] == "segment":
        return fuzz_segments(oracle, rng, budget)
    if len(sys.argv) > 3 and sys.argv[3] == "plonk":
        return fuzz_plonk(oracle, rng, budget)
    if len(sys.argv) > 3 and sys.argv[3] == "tracegen":
        return fuzz_tracegen(rng, budget)
    cases = [(n, f) for n, f in inspect.getmembers(t, inspect.isfunction) if n.startswith("test_")]
    t0, runs, per_case = time.perf_counter(), 0, {}
    while time.perf_counter() - t0 < budget:
        for name, fn in cases:
            kwargs = {"oracle": oracle}
            if "hasher" in inspect.signature(fn).parameters:
                kwargs["hasher"] = 0
            mark = len(drawn)
            try:
                fn(**kwargs)
            except AssertionError:
                print(json.dumps({"FAILED": name, "draws": drawn[mark:]}))
                raise
            runs += len(drawn) - mark
            per_case[name] = per_case.get(name, 0) + len(drawn) - mark
            if time.perf_counter() - t0 >= budget:
                break
    print(json.dumps({"cases": runs, "seconds": round(time.perf_counter() - t0, 1), "per_table_case": per_case,
                      "log_n_range": [min(d[0] for d in drawn), max(d[0] for d in drawn)], "mismatches": 0}))


def fuzz_plonk(oracle, rng, budget):
    """The PLONK slice: circuits with all fourteen gate kinds (valid witness and junk wires alike) at random heights, FRI
    shapes and seeds, device proof against the oracle's restatement of plonky2's prove() word for word."""
    import tests.oracle_lib as ol
    import tests.test_gpu_plonk as tp
    from oracle import plonk as PK
    from tests.gpu_util import to_dev
    ol.setup_fri_api(oracle)
    t0, cases, heights = time.perf_counter(), 0, set()
    while time.perf_counter() - t0 < budget:
        db = int(rng.integers(6, 12))
        kw = dict(proof_of_work_bits=int(rng.integers(0, 9)), num_query_rounds=int(rng.integers(1, 9)))
        seed = int(rng.integers(1, 1 << 30))
        build = PK.build_mixed_circuit if rng.random() < 0.7 else PK.build_arithmetic_circuit
        circ, wires, pis = build(db, seed=seed, cfg=PK.CircuitConfig(**kw))
        wires, _ = PK.set_public_input_wires(oracle, circ, wires, pis)
        if rng.random() < 0.5:                           # unsatisfying wires: every constraint non-zero on both sides
            wires = np.random.default_rng(seed).integers(0, 1 << 64, size=wires.shape, dtype=np.uint64)
        exp = PK.prove(oracle, ol, circ, wires, pis)
        cd = tp._device_circuit(circ)
        got = cd.prove(to_dev(wires), pis)
        assert np.array_equal(got.wires_cap, exp["wires_cap"]) and np.array_equal(got.plonk_zs_partial_products_cap, exp["zs_pp_cap"]), (db, seed)
        assert np.array_equal(got.quotient_polys_cap, exp["quotient_cap"]), (db, seed)
        assert np.array_equal(got.openings.reshape(-1), exp["openings"]) and np.array_equal(got.opening_proof, exp["fri"]), (db, seed)
        cd.free()
        cases += 1
        heights.add(db)
    print(json.dumps({"plonk_cases": cases, "seconds": round(time.perf_counter() - t0, 1), "degree_bits": sorted(heights),
                      "mismatches": 0}))


def fuzz_tracegen(rng, budget):
    """The two generators with real control flow -- Memory (sort, fill_gaps, padding, final memory) and Arithmetic
    (modular / division / shift operations on 256-bit integers) -- on random logs of random sizes against the oracle's
    literal restatements, cell for cell."""
    import tests.test_gpu_arithtrace as ta
    import tests.test_gpu_memtrace as tm
    from tests.test_oracle_tracegen import sample_memory_ops
    t0, n_mem, n_arith, cells = time.perf_counter(), 0, 0, 0
    while time.perf_counter() - t0 < budget:
        r = np.random.default_rng(int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.5:                           # small logs with wide gaps in every address component
            ops, before, stale = sample_memory_ops(r, int(rng.integers(1, 300)))
        else:                                            # dense logs: reads of earlier writes, few timestamp gaps
            ops, before = tm.random_log(r, int(rng.integers(1, 3000)), int(rng.integers(0, 200)),
                                        int(rng.choice([4, 64, 3000, 1 << 20])))
            stale = sorted({int(c) for c in rng.integers(0, 6, size=int(rng.integers(0, 3)))})
        cells += 30 * tm._check(ops, before, stale).shape[1]
        n_mem += 1
        cells += 116 * ta._check(ta.random_operations(r, int(rng.integers(0, 400)))).shape[1]
        n_arith += 1
    print(json.dumps({"memory_logs": n_mem, "arithmetic_logs": n_arith, "cells_compared": cells,
                      "seconds": round(time.perf_counter() - t0, 1), "mismatches": 0}))


def fuzz_segments(oracle, rng, budget):
    """Whole-segment proofs (all tables, 10 CTLs, lookups, public values) at random heights / live-table sets /
    hashers / FRI shapes against the oracle's prove_with_traces, word for word."""
    import tests.test_gpu_segment as ts
    caps = [7, 7, 7, 5, 6, 7, 7, 7, 7]                  # per-table height cap: keeps the Python prover to seconds
    optional = [1, 3, 4, 5, 8]
    drawn = []

    def draw(log_n):
        log_n[:] = [int(rng.integers(4, c + 1)) for c in caps]
        in_use = [True] * 9
        for t in optional:
            in_use[t] = bool(rng.integers(0, 4))          # each optional table absent a quarter of the time
        c = (int(rng.integers(0, 2)), in_use, int(rng.integers(1, 1 << 30)),
             dict(pow_bits=int(rng.integers(0, 9)), queries=int(rng.integers(1, 5))))
        drawn.append((list(log_n),) + c)
        return c

    ts.FUZZ = draw
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        try:
            ts.test_segment_proof_matches_oracle(oracle, 0, [True] * 9)
        except AssertionError:
            print(json.dumps({"FAILED": "segment", "draw": drawn[-1]}))
            raise
    print(json.dumps({"segment_cases": len(drawn), "seconds": round(time.perf_counter() - t0, 1),
                      "absent_table_sets": len({tuple(d[2]) for d in drawn}), "mismatches": 0}))


if __name__ == "__main__":
    main()
