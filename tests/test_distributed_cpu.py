"""CPU (gloo, world_size 2) coverage of the N>1 path: segment / table sharding and the cap
all-gather.  The GPU commit is replaced by the oracle here -- this test exercises the
distribution logic only; the kernels are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zk_evm_amd.sharding import assign_segments, assign_tables, gather_caps, table_cost

# trace column counts of the nine tables; realistic witness_b19807080 size profile
SHAPES = [(116, 17), (71, 14), (85, 19), (2431, 17), (438, 13), (523, 16), (30, 21), (12, 19), (12, 19)]


def test_assign_segments_partition():
    for n, w in [(0, 2), (1, 2), (7, 2), (8, 8), (20, 8)]:
        parts = assign_segments(n, w)
        assert sorted(x for p in parts for x in p) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_assign_tables_partition_and_balance():
    for w in (1, 2, 4, 8):
        parts = assign_tables(SHAPES, w)
        assert sorted(t for p in parts for t in p) == list(range(len(SHAPES)))
        loads = [sum(table_cost(*SHAPES[t]) for t in p) for p in parts]
        biggest = max(table_cost(*s) for s in SHAPES)
        assert max(loads) <= max(biggest, sum(loads) / w * 1.34) + 1e-9
    assert assign_tables(SHAPES, 2) == assign_tables(SHAPES, 2)  # deterministic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.oracle_lib import load_oracle, splitmix64
    o = load_oracle()
    shapes = [(5, 4), (12, 6), (3, 5), (9, 4), (30, 5)]
    mine = assign_tables(shapes, world)[rank]
    local = {}
    for t in mine:
        c, ln = shapes[t]
        vals = np.stack([splitmix64(100 * t + k, 1 << ln) for k in range(c)])
        local[t] = o.commit_values(vals, want_leaves=False)["cap"]
    caps = gather_caps(local, len(shapes))
    q.put((rank, [c.tolist() for c in caps]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_caps_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    # equals the single-process result
    from tests.oracle_lib import load_oracle, splitmix64
    o = load_oracle()
    shapes = [(5, 4), (12, 6), (3, 5), (9, 4), (30, 5)]
    for t, (c, ln) in enumerate(shapes):
        vals = np.stack([splitmix64(100 * t + k, 1 << ln) for k in range(c)])
        assert o.commit_values(vals, want_leaves=False)["cap"].tolist() == res[0][t]


# ---- segment scheduler (zk_evm_amd/scheduler.py) -- no GPU: the prove step is injected -----------------------
def _fake_prove(st, job):
    import time
    time.sleep(0.01 * (job.tag % 3))
    if job.tag == 13:
        raise RuntimeError("segment 13 is cursed")
    return ("proof", job.tag, job.load(None))


def test_scheduler_orders_results_and_isolates_failures():
    from zk_evm_amd.scheduler import SegmentJob, SegmentScheduler
    jobs = [SegmentJob(lambda dev, i=i: i * i, [True] * 9, None, tag=i) for i in range(12)]
    with SegmentScheduler(None, None, devices=[0, 1], in_flight=2, prove_fn=_fake_prove) as sch:
        out = sch.map(jobs)
        assert out == [("proof", i, i * i) for i in range(12)]
        assert sum(st.segments for st in sch.stats) == 12 and len(sch.stats) == 4
        bad = sch.submit(SegmentJob(lambda dev: 0, [True] * 9, None, tag=13))
        with pytest.raises(RuntimeError):
            bad.result(timeout=30)
        # the worker that hit the failure keeps serving
        assert sch.map([SegmentJob(lambda dev: 5, [True] * 9, None, tag=1)]) == [("proof", 1, 5)]
    with pytest.raises(RuntimeError):
        sch.submit(jobs[0])


def _sched_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zk_evm_amd.scheduler import SegmentJob, run_distributed
    jobs = [SegmentJob(lambda dev, i=i: 100 + i, [True] * 9, None, tag=i) for i in range(7)]
    out = run_distributed(None, None, jobs, device=0, in_flight=2,
                          prove_fn=lambda st, job: (dist.get_rank(), job.tag, job.load(None)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_run_distributed_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    # rank 0 holds every proof in segment order; segment i was proven by rank i % 2
    assert res[0] == [(i % 2, i, 100 + i) for i in range(7)]
