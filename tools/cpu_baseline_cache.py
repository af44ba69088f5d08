#!/usr/bin/env python3
"""MEASURED whole-segment times of the CPU oracle, cached (r05 verdict, weak 7 / next 6): `cpu_baseline` on the bench line is a
measurement whenever one exists for exactly this (oracle sources, CPU model, cores, segment shape, hasher), and only otherwise the
bounded 2^15-row sample scaled linearly in committed cells (which understates an n log n workload, i.e. flatters the CPU).

    python tools/cpu_baseline_cache.py measure [--log-ns realistic | 20,20,...] [--hasher 0]     # CPU only, no GPU in the process
    python tools/cpu_baseline_cache.py show

The cache is tools/cpu_baseline_cache.json (committed).  A whole 2^20 nine-table segment takes the oracle ~23 minutes on 16 cores:
that is why the bench cannot measure it inside its own run and why a dedicated run is cached.  The oracle is test infrastructure:
this tool, bench.py's cpu_baseline leg and the tests are its only users."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tools", "cpu_baseline_cache.json")


def oracle_source_hash() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "oracle")
    for f in sorted(os.listdir(d)):
        if f.endswith((".c", ".h", ".py")) or f == "Makefile":
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load(path=CACHE):
    try:
        with open(path) as f:
            return json.load(f).get("entries", [])
    except (OSError, ValueError):
        return []


def key_of(e):
    return (e.get("oracle_hash"), e.get("cpu_model"), int(e.get("cores", 0)), tuple(e.get("log_ns", [])), int(e.get("hasher", 0)),
            bool(e.get("cdk_erigon", False)))


def lookup(cpu_model, cores, log_ns, hasher=0, cdk_erigon=False, entries=None, oracle_hash=None):
    want = (oracle_hash or oracle_source_hash(), cpu_model, int(cores), tuple(int(x) for x in log_ns), int(hasher), bool(cdk_erigon))
    for e in (load() if entries is None else entries):
        if key_of(e) == want:
            return e
    return None


def store(entry, path=CACHE):
    entries = [e for e in load(path) if key_of(e) != key_of(entry)] + [entry]
    with open(path, "w") as f:
        json.dump({"entries": entries}, f, indent=1)
        f.write("\n")


def measure(log_ns, hasher=0):
    """One whole nine-table segment proof (standard_fast_config) of synthetic traces by the oracle alone, on every core the process
    may use.  No GPU, no libzkstark in the process."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import tests.oracle_lib as ol
    from oracle import airs as oairs
    from oracle import segment as oseg
    from tests.test_gpu_segment import make_pv
    from tools.bench_secondary import _cpu_model, _host_cores
    from tools.benchlib import segment_committed_cells, synthetic_segment_traces
    o = ol.load_oracle()
    ol.setup_fri_api(o)
    cores = _host_cores(o)
    traces = synthetic_segment_traces(log_ns, "cpu", seed=11)
    host = [t.numpy().view(np.uint64) % np.uint64(0xFFFFFFFF00000001) for t in traces]
    del traces
    pvd = make_pv(np.random.default_rng(4))
    cfg = ol.make_cfg(hasher=hasher)
    t0 = time.perf_counter()
    oseg.prove_with_traces(o, ol, cfg, host, [True] * 9, pvd, oairs.CPU_TEST_CONSTS, fast=True)
    sec = time.perf_counter() - t0
    return {"oracle_hash": oracle_source_hash(), "cpu_model": _cpu_model(), "cores": cores, "log_ns": [int(x) for x in log_ns],
            "hasher": int(hasher), "cdk_erigon": False, "cpu_seconds": sec, "committed_cells": segment_committed_cells(list(log_ns)),
            "measured_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
            "how": "tools/cpu_baseline_cache.py measure: oracle/segment.py prove_with_traces(fast=True), standard_fast_config, synthetic traces"}


def main(argv):
    if len(argv) < 2 or argv[1] not in ("measure", "show"):
        sys.stderr.write(__doc__)
        return 2
    if argv[1] == "show":
        print(json.dumps({"oracle_hash_now": oracle_source_hash(), "entries": load()}, indent=1))
        return 0
    sys.path.insert(0, ROOT)
    from tools.benchlib import REALISTIC_LOG_NS
    log_ns = list(REALISTIC_LOG_NS)
    hasher = 0
    if "--log-ns" in argv:
        v = argv[argv.index("--log-ns") + 1]
        log_ns = list(REALISTIC_LOG_NS) if v == "realistic" else [int(x) for x in v.split(",")]
    if "--hasher" in argv:
        hasher = int(argv[argv.index("--hasher") + 1])
    assert len(log_ns) == 9
    e = measure(log_ns, hasher)
    store(e)
    print(json.dumps(e))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
