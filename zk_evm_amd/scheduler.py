"""Segment scheduler: the product-side analogue of the reference's mapping of segments onto workers
(`zero/src/prover.rs:221-224`: the segment iterator is folded over paladin workers, one `SegmentProof` op per
segment, `zero/src/ops.rs:24-67`).

Segments are independent units (fresh `Challenger` per segment, `evm_arithmetization/src/prover.rs:118`), so the
unit of distribution is a whole segment and no data-path collective exists (level 1):

  * inside one process a `SegmentScheduler` owns one worker per (GPU, slot): a thread with its own `zk_ctx`
    (arena, tables) and its own HIP stream, pulling jobs from one shared queue (work stealing by construction:
    the next free worker takes the next segment).  `in_flight` > 1 keeps several segments resident on one GPU so
    that one segment's latency-bound stretches (small Merkle levels, Fiat-Shamir read-backs) are filled by another's
    kernels;
  * across processes (one process per GPU, `torch.distributed`, RCCL on the GPU box / gloo in the CPU tests)
    `run_distributed` deals the job list round-robin (`sharding.assign_segments`), proves the local share through a
    local scheduler and gathers the proofs on rank 0.  The only traffic is the final gather of proof objects.

A job carries a `load(device)` callable instead of tensors: the trace must be materialised on the GPU that proves it
(witness generation hands over host logs; `tracegen.Traces.into_tables` builds the tables on that device).
There is no CPU fallback: a worker without a usable GPU raises `ZkStarkError` on its first job."""
import queue
import threading
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence

from .sharding import assign_segments


@dataclass
class SegmentJob:
    """One segment to prove.  `load(device)` returns the per-table column-major trace tensors on `device`
    (`trace_poly_values` of prover.rs:72-80)."""
    load: Callable[[Any], Sequence]
    table_in_use: Sequence[bool]
    public_values: Any
    tag: Any = None
    timing: Optional[dict] = None


@dataclass
class WorkerStats:
    device: int
    slot: int
    segments: int = 0
    busy_s: float = 0.0
    errors: List[str] = field(default_factory=list)


class SegmentScheduler:
    """`in_flight` workers per device, each = thread + `Context` + HIP stream, fed from one queue."""

    def __init__(self, all_stark, config, devices: Sequence[int] = (0,), in_flight: int = 1, prove_fn=None):
        if in_flight < 1 or not devices:
            raise ValueError("need at least one device and one worker per device")
        self.all_stark, self.config = all_stark, config
        self._prove_fn = prove_fn          # injection point for the CPU tests (no GPU): (worker, job) -> proof
        self._q: "queue.Queue" = queue.Queue()
        self._threads: List[threading.Thread] = []
        self.stats: List[WorkerStats] = []
        self._closed = False
        for d in devices:
            for s in range(in_flight):
                st = WorkerStats(int(d), s)
                self.stats.append(st)
                th = threading.Thread(target=self._run, args=(st,), name=f"zk-segment-worker-{d}.{s}", daemon=True)
                th.start()
                self._threads.append(th)

    # ---- worker ------------------------------------------------------------------------------------------
    def _run(self, st: WorkerStats):
        import time
        ctx = stream = torch = None
        while True:
            item = self._q.get()
            if item is None:
                break
            job, fut = item
            if not fut.set_running_or_notify_cancel():
                continue
            t0 = time.perf_counter()
            try:
                if self._prove_fn is not None:
                    fut.set_result(self._prove_fn(st, job))
                else:
                    if ctx is None:                       # lazily: a scheduler can be built without touching the GPU
                        import torch as _t
                        from .context import Context
                        torch = _t
                        torch.cuda.set_device(st.device)
                        stream = torch.cuda.Stream(device=st.device)
                        ctx = Context(st.device)
                    from . import segment as sg
                    dev = torch.device("cuda", st.device)
                    with torch.cuda.device(st.device), torch.cuda.stream(stream):
                        traces = job.load(dev)
                        proof = sg.prove_with_traces(self.all_stark, self.config, traces, job.table_in_use,
                                                     job.public_values, ctx=ctx, timing=job.timing)
                        stream.synchronize()
                        del traces
                    fut.set_result(proof)
                st.segments += 1
            except Exception as e:                        # a failed segment fails its future, not the worker
                                                          # (KeyboardInterrupt / SystemExit are not swallowed)
                st.errors.append(repr(e))
                fut.set_exception(e)
            finally:
                st.busy_s += time.perf_counter() - t0
        if ctx is not None:
            ctx.close()

    # ---- client side ---------------------------------------------------------------------------------------
    def submit(self, job: SegmentJob) -> Future:
        if self._closed:
            raise RuntimeError("scheduler is shut down")
        fut: Future = Future()
        self._q.put((job, fut))
        return fut

    def map(self, jobs: Sequence[SegmentJob]) -> list:
        """Prove every job; results in job order (the aggregation layer above needs segment order)."""
        futs = [self.submit(j) for j in jobs]
        return [f.result() for f in futs]

    def shutdown(self):
        if self._closed:
            return
        self._closed = True
        for _ in self._threads:
            self._q.put(None)
        for th in self._threads:
            th.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()


class SegmentFailure(RuntimeError):
    """One or more segments of a `run_distributed` job list failed; `.failures` = [(job index, rank, message)]."""

    def __init__(self, failures):
        self.failures = list(failures)
        super().__init__("; ".join("segment %d failed on rank %d: %s" % f for f in self.failures))


def run_distributed(all_stark, config, jobs: Sequence[SegmentJob], device: int = 0, in_flight: int = 1,
                    group=None, prove_fn=None, gather: bool = True, encode=None, decode=None):
    """One process per GPU: rank r proves jobs r, r + W, r + 2W, ... on `device` and rank 0 receives all proofs in
    job order (None elsewhere).  Without an initialised process group this is the single-process scheduler.

    What crosses ranks is u64 words in tensors (collectives.py; RCCL on the GPU node): `encode(result) -> u64 array`
    on the proving rank, `decode(words, job) -> result` on rank 0 -- by default the flat `AllProof` words of
    segment.all_proof_to_words (the job's `PublicValues` stay with the submitter).  A failed segment never strands the
    other ranks: failures are caught per job, travel through the same gather as (index, message) records, and every
    rank raises `SegmentFailure` AFTER the collective (so all proofs that did succeed have been delivered)."""
    import numpy as np
    import torch.distributed as dist

    from . import collectives as co
    world, rank = 1, 0
    multi = dist.is_available() and dist.is_initialized()
    if multi:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = assign_segments(len(jobs), world)[rank]
    results, failures = [], []
    with SegmentScheduler(all_stark, config, [device], in_flight, prove_fn) as sch:
        futs = [sch.submit(jobs[i]) for i in mine]
        for i, f in zip(mine, futs):
            try:
                results.append((i, f.result()))
            except Exception as e:             # the failure is reported through the collective below, not instead of it
                failures.append((i, rank, repr(e)))
    if not multi or not gather:                  # (a process group of ONE rank still goes through the collectives below)
        if multi:
            co.agree(None if not failures else SegmentFailure(failures), "a segment proof", group)
        elif failures:
            raise SegmentFailure(failures)
        out = [None] * len(jobs)
        for i, p in results:
            out[i] = p
        return out
    if encode is None:
        from . import segment as sg
        encode = sg.all_proof_to_words
        decode = lambda w, job: sg.all_proof_from_words(w, job.public_values)          # noqa: E731
    recs = []
    for i, p in results:
        try:
            recs.append(np.concatenate([np.array([i, 0], dtype=np.uint64), np.asarray(encode(p), dtype=np.uint64).reshape(-1)]))
        except Exception as e:
            failures.append((i, rank, "encode: " + repr(e)))
    for i, r, msg in failures:
        recs.append(np.concatenate([np.array([i, 1], dtype=np.uint64), co.text_words(msg)]))
    parts = co.gather_varlen_words(co.pack_records(recs), dst=0, group=group)
    # every rank learns whether anything failed anywhere (one more tiny all-reduce), so that all of them raise
    n_failed = sum(int(x[0]) for x in co.all_gather_words(np.array([len(failures)], dtype=np.uint64), 1, group))
    if rank != 0:
        if failures:
            raise SegmentFailure(failures)
        if n_failed:
            raise co.RemoteRankError("%d segment(s) failed on other ranks" % n_failed)
        return None
    out = [None] * len(jobs)
    all_failures = []
    for r, part in enumerate(parts):
        for rec in co.unpack_records(part):
            i, status = int(rec[0]), int(rec[1])
            if status:
                all_failures.append((i, r, co.words_text(rec[2:])))
            else:
                out[i] = decode(rec[2:], jobs[i])
    if all_failures:
        err = SegmentFailure(sorted(all_failures))
        err.partial = out                      # the proofs that did succeed
        raise err
    return out
