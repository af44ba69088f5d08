"""Soak of the several-segments-in-flight mode: W workers (own Context + stream + thread) prove DIFFERENT segments at the
realistic table heights concurrently, many rounds; every proof must equal the digest of the same segment proven alone,
and every arena must end empty.  Usage: python tools/soak_concurrent.py [workers] [rounds]"""
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
    from tools.soak_segment import digest
    from zk_evm_amd.all_stark import AllStark
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    cfg = zk.StarkConfig()
    shapes = [REALISTIC_LOG_NS, [16, 13, 18, 15, 12, 15, 20, 18, 18], [17, 15, 17, 16, 14, 16, 19, 17, 17]]
    traces = [synthetic_segment_traces(shapes[w % len(shapes)], dev, seed=50 + w) for w in range(workers)]
    serial = [digest(sg.prove_with_traces(AllStark((1, 2, 3, 4)), cfg, tr, [True] * 9, sg.PublicValues())) for tr in traces]
    torch.cuda.synchronize()
    ctxs = [zk.Context(0) for _ in range(workers)]
    bad, errors = [0] * workers, []

    def run(w):
        try:
            st = AllStark((1, 2, 3, 4))
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(rounds):
                    d = digest(sg.prove_with_traces(st, cfg, traces[w], [True] * 9, sg.PublicValues(), ctx=ctxs[w]))
                    bad[w] += d != serial[w]
                torch.cuda.current_stream().synchronize()
        except Exception as e:
            errors.append(repr(e))
    th = [threading.Thread(target=run, args=(w,)) for w in range(workers)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    print(json.dumps({"workers": workers, "rounds": rounds, "proofs": workers * rounds, "seconds": round(el, 1),
                      "proofs_per_s": round(workers * rounds / el, 2), "mismatching_proofs": sum(bad), "errors": errors,
                      "arenas_in_use_after": [c.mem_stats()["in_use"] for c in ctxs]}))
    assert not errors and not sum(bad)


if __name__ == "__main__":
    main()
