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


def test_library_assignment_equals_the_python_one():
    """zk_assign_tables (what zk_prove_segment_table_parallel uses) == sharding.assign_tables, row-sharded tables aside"""
    from zk_evm_amd.sharding import table_owners
    rng = np.random.default_rng(5)
    for world in (1, 2, 4, 8):
        for _ in range(20):
            shapes = [(int(rng.integers(1, 2500)), int(rng.integers(4, 23))) for _ in range(9)]
            wide = [t for t in range(9) if rng.random() < 0.2]
            solo = [t for t in range(9) if t not in wide]
            own = table_owners(shapes, world, wide)
            want = assign_tables([shapes[t] for t in solo], world)
            for r, ts in enumerate(want):
                assert all(own[solo[k]] == r for k in ts)
            assert all(own[t] == 0 for t in wide)


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


def _tuple_codec():
    """results of the injected prove step are (rank, tag, value) int triples: three words"""
    return (lambda r: np.array(r, dtype=np.uint64)), (lambda w, job: tuple(int(x) for x in w))


def _sched_worker(rank, world, port, q, cursed=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zk_evm_amd.collectives import RemoteRankError
    from zk_evm_amd.scheduler import SegmentFailure, SegmentJob, run_distributed
    jobs = [SegmentJob(lambda dev, i=i: 100 + i, [True] * 9, None, tag=i) for i in range(7)]

    def prove(st, job):
        if job.tag == cursed:
            raise RuntimeError("segment %d is cursed" % job.tag)
        return (dist.get_rank(), job.tag, job.load(None))
    enc, dec = _tuple_codec()
    try:
        out = run_distributed(None, None, jobs, device=0, in_flight=2, prove_fn=prove, encode=enc, decode=dec)
        q.put((rank, "ok", out))
    except SegmentFailure as e:
        q.put((rank, "SegmentFailure", (e.failures, getattr(e, "partial", None))))
    except RemoteRankError as e:
        q.put((rank, "RemoteRankError", str(e)))
    dist.barrier()                                    # reached by every rank: nobody is stuck in a collective
    dist.destroy_process_group()


def _run_sched(cursed=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, q, cursed)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, kind, payload = q.get(timeout=120)
        res[r] = (kind, payload)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_run_distributed_gloo_world2():
    res = _run_sched()
    assert res[1] == ("ok", None)
    # rank 0 holds every proof in segment order; segment i was proven by rank i % 2
    assert res[0] == ("ok", [(i % 2, i, 100 + i) for i in range(7)])


def test_run_distributed_failure_reaches_every_rank_without_deadlock():
    """a failed segment used to raise before the gather and strand the other ranks in it.  Now the
    failure travels through the gather; both ranks leave the collective, both raise, rank 0 names segment and rank and
    still holds the six proofs that succeeded."""
    res = _run_sched(cursed=3)                         # segment 3 is rank 1's
    kind1, pay1 = res[1]
    assert kind1 == "SegmentFailure" and pay1[0][0][:2] == (3, 1)
    kind0, (failures, partial) = res[0]
    assert kind0 == "SegmentFailure" and failures[0][:2] == (3, 1) and "cursed" in failures[0][2]
    assert partial == [(i % 2, i, 100 + i) if i != 3 else None for i in range(7)]
    res = _run_sched(cursed=4)                         # segment 4 is rank 0's: rank 1 learns of it too
    assert res[0][0] == "SegmentFailure" and res[1][0] == "RemoteRankError"


def test_collectives_word_codecs():
    from zk_evm_amd import collectives as co
    recs = [np.arange(5, dtype=np.uint64), np.zeros(0, np.uint64), np.array([2 ** 64 - 1], dtype=np.uint64)]
    back = co.unpack_records(co.pack_records(recs))
    assert len(back) == 3 and all(np.array_equal(a, b) for a, b in zip(recs, back))
    assert co.words_text(co.text_words("ZkStarkError(-3, 'out of memory \u2713')")) == "ZkStarkError(-3, 'out of memory \u2713')"
    # single process: the collectives degrade to identities
    assert np.array_equal(co.all_gather_words(np.array([7, 8]), 2)[0], [7, 8])
    assert np.array_equal(co.gather_varlen_words(np.array([1, 2, 3]))[0], [1, 2, 3])
    co.agree(None, "nothing")
    with pytest.raises(ValueError):
        co.agree(ValueError("x"), "something")


def test_stark_proof_words_roundtrip():
    from zk_evm_amd.prover import StarkProof
    import zk_evm_amd.segment as sg
    rng = np.random.default_rng(1)
    r = lambda *sh: rng.integers(0, 2 ** 63, size=sh, dtype=np.uint64)           # noqa: E731
    protos = [StarkProof(r(16, 4), r(16, 4), r(16, 4), r(40, 2), r(1000), r(12), 3, 17),
              StarkProof(r(16, 4), None, r(16, 4), r(9, 2), r(10), None, 0, None)]
    for p in protos:
        q, used = StarkProof.from_words(p.to_words())
        assert used == p.to_words().size
        for f in ("trace_cap", "auxiliary_polys_cap", "quotient_polys_cap", "openings", "opening_proof", "init_challenger_state"):
            a, b = getattr(p, f), getattr(q, f)
            assert (a is None and b is None) or np.array_equal(a, b)
        assert (q.num_ctl_zs, q.degree_bits) == (p.num_ctl_zs, p.degree_bits)
    pv = sg.PublicValues()
    pv.mem_before = sg.MemCap([[1, 2, 3, 4]] * 16)
    pv.mem_after = sg.MemCap([[5, 6, 7, 8]] * 16)
    ap = sg.AllProof(sg.MultiProof([sg.StarkProofWithMetadata(protos[0], protos[0].init_challenger_state), None],
                                   [(11, 12), (13, 14)]), pv, [True, False])
    back = sg.all_proof_from_words(sg.all_proof_to_words(ap), sg.PublicValues())
    assert back.table_in_use == [True, False] and back.multi_proof.ctl_challenges == [(11, 12), (13, 14)]
    assert back.multi_proof.stark_proofs[1] is None and back.public_values.mem_after.mem_cap == pv.mem_after.mem_cap
    assert np.array_equal(back.multi_proof.stark_proofs[0].proof.opening_proof, protos[0].opening_proof)


def test_no_pickled_collective_in_the_product():
    """r02 verdict: the multi-GPU paths move fixed-shape tensors, never pickled Python objects."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zk_evm_amd")
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            for bad in ("all_gather_object(", "gather_object(", "broadcast_object_list(", "scatter_object_list(", "send_object_list("):
                assert bad not in src, (fn, bad)


def test_split_columns_and_residue_classes():
    from zk_evm_amd.sharding import _bitrev, split_columns
    for n, w in [(2431, 8), (116, 2), (5, 4), (12, 16)]:
        parts = split_columns(n, w)
        assert [c for p in parts for c in p] == list(range(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    # leaf slot s = q * (N / W) + t  <->  natural row bitrev_N(s) = bitrev(t) * W + bitrev_W(q): the row residue classes
    log_n, log_w = 6, 2
    for q in range(1 << log_w):
        for t in range(1 << (log_n - log_w)):
            s = (q << (log_n - log_w)) | t
            assert _bitrev(s, log_n) == _bitrev(t, log_n - log_w) * (1 << log_w) + _bitrev(q, log_w)


# ---- bench.py's process groups: the safety net the scaling line runs on -------------------------------------------------------
def _rungroup_worker(rank, world, port, q, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import argparse
    import bench
    a = argparse.Namespace(dist_backend=backend, dist_timeout_s=20.0)
    rg = bench.RunGroup(a, rank, world, 0, True)
    rg.barrier()
    mx = rg.max_over_ranks(10.0 + rank)
    per = rg.per_rank(100.0 + rank)
    st = bench.dist_selftest(rank, world, rg.backend, rg.group())
    q.put((rank, dict(rg.info), rg.backend, mx, per, st))
    rg.dist.barrier(group=rg.gloo)          # (not rg.close(): after a fallback that ends the process before the queue flushes)
    rg.dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_rungroup_falls_back_to_gloo_and_never_fails(backend):
    """bench.py --gpus N must print its line whatever RCCL does (r03 verdict, next-round item 1).  There is no GPU here, so
    the `nccl` group cannot come up: both ranks must agree on the gloo fallback, record why, and still get their barrier,
    the MAX over ranks, the per-rank times and the product collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rungroup_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=180)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        info, used, mx, per, st = res[r]
        assert used == "gloo" and mx == 11.0 and per == [100.0, 101.0]
        assert st["ok"], st
        if backend == "nccl":
            assert info["ok"] is False and info["fallback"] == "gloo" and info["tried"] == "nccl" and info["error"]
        else:
            assert info["ok"] is True and "fallback" not in info
