"""Segments in flight per GPU: W worker threads, each with its own Context + stream, prove the same shape back to back;
aggregate proofs/s for W = 1, 2, 3 (the input traces are shared and read-only; a W whose arenas would not fit is skipped).  At 2^20 a single segment already keeps the GPU 98.5 % busy; at the realistic table
heights the small tables leave SIMDs idle that a second in-flight segment can use.
Usage: python tools/bench_concurrent.py [realistic|<log_n>] [proofs per worker] [stagger seconds] [worker counts, e.g. 1,2,4]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from bench import REALISTIC_LOG_NS, synthetic_segment_traces
    from zk_evm_amd.all_stark import AllStark
    shape = sys.argv[1] if len(sys.argv) > 1 else "realistic"
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    stagger = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0    # seconds between worker starts
    log_ns = REALISTIC_LOG_NS if shape == "realistic" else [int(shape)] * 9
    dev = torch.device("cuda:0")
    cfg = zk.StarkConfig()
    out = {"log_ns": log_ns, "proofs_per_worker": per, "stagger_s": stagger, "proofs_per_s": {}}
    shared = synthetic_segment_traces(log_ns, dev, seed=3)     # read-only inputs, resident once, proven by every worker
    total_hbm = torch.cuda.get_device_properties(0).total_memory
    peak = None
    for workers in ([int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (1, 2, 3)):
        if peak is not None and workers * peak + torch.cuda.memory_allocated() > 0.88 * total_hbm:
            out["proofs_per_s"][str(workers)] = None         # would not fit: workers x arena peak + inputs
            continue
        traces = [shared] * workers
        ctxs = [zk.Context(0) for _ in range(workers)]
        streams = [torch.cuda.Stream() for _ in range(workers)]
        torch.cuda.synchronize()
        errors = []

        def run(w, n):
            try:
                time.sleep(w * stagger)                          # offset the workers' phases
                with torch.cuda.stream(streams[w]):
                    st = AllStark((1, 2, 3, 4))
                    for _ in range(n):
                        sg.prove_with_traces(st, cfg, traces[w], [True] * 9, sg.PublicValues(), ctx=ctxs[w])
                    streams[w].synchronize()
            except Exception as e:
                errors.append(repr(e))
        for n in (2, per):                               # warm-up (arena growth), then the timed round
            th = [threading.Thread(target=run, args=(w, n)) for w in range(workers)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            el = time.perf_counter() - t0
        assert not errors, errors
        out["proofs_per_s"][str(workers)] = round(workers * per / el, 3)
        if peak is None:
            peak = ctxs[0].mem_stats()["peak_in_use"]
            out["arena_peak_GB_per_worker"] = round(peak / 1e9, 2)
        del traces, ctxs
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
