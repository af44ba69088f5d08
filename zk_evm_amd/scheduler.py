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
            except BaseException as e:                    # a failed segment fails its future, not the worker
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


def run_distributed(all_stark, config, jobs: Sequence[SegmentJob], device: int = 0, in_flight: int = 1,
                    group=None, prove_fn=None, gather: bool = True):
    """One process per GPU: rank r proves jobs r, r + W, r + 2W, ... on `device` and rank 0 receives all proofs in
    job order (None elsewhere).  Without an initialised process group this is the single-process scheduler."""
    import torch.distributed as dist
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = assign_segments(len(jobs), world)[rank]
    with SegmentScheduler(all_stark, config, [device], in_flight, prove_fn) as sch:
        local = sch.map([jobs[i] for i in mine])
    if world == 1 or not gather:
        out = [None] * len(jobs)
        for i, p in zip(mine, local):
            out[i] = p
        return out
    parts = [None] * world if rank == 0 else None
    dist.gather_object(list(zip(mine, local)), parts, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * len(jobs)
    for part in parts:
        for i, p in part:
            out[i] = p
    return out
