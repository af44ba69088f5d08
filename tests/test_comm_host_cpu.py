"""CPU, world 2 and 4: the library's host-staged transport (csrc/comm_host.inc, `zk_comm_create_host` with no zk_ctx) -- the
communicator the multi-GPU provers run on when ranks share a GPU or RCCL cannot come up -- moving HOST payloads between real
processes: all-gather, broadcast, and the two all-to-alls of the row-sharded table prover (row blocks -> column shards -> row
shards, column counts the ranks do not divide) through outboxes much smaller than the messages (several rounds each); and the
failure protocol: a rank that never arrives costs its peers the time limit and ZK_ERR_COMM, not a hang.  The kernels between
the exchanges need a GPU (tests/test_gpu_multirank.py); the routing does not."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

ZK_ERR_COMM = -6


def _lib():
    from zk_evm_amd._lib import load_library
    return load_library()


def _split(n, world):
    base, extra = divmod(n, world)
    out, pos = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(range(pos, pos + k))
        pos += k
    return out


def _a2a(lib, h, send, recv):
    W = len(send)
    sp = (C.c_void_p * W)(*[s.ctypes.data for s in send])
    rp = (C.c_void_p * W)(*[r.ctypes.data for r in recv])
    sn = (C.c_size_t * W)(*[s.nbytes for s in send])
    rn = (C.c_size_t * W)(*[r.nbytes for r in recv])
    return lib.zk_comm_all_to_all_host(h, sp, sn, rp, rn)


def _worker(rank, world, name, q, absent=None, rank0_late=0.0):
    try:
        lib = _lib()
        h = C.c_void_p()
        if rank == 0 and rank0_late:
            import time
            time.sleep(rank0_late)
        rc = lib.zk_comm_create_host(None, name.encode(), rank, world, 4096 * world, C.byref(h))      # 4 KiB per (src, dst) and round
        assert rc == 0 and lib.zk_comm_world(h) == world and lib.zk_comm_rank(h) == rank and lib.zk_comm_transport(h) == b"host"
        ok = True
        # all-gather / broadcast of host words
        mine = np.arange(5, dtype=np.uint64) + 100 * rank
        got = np.zeros((world, 5), dtype=np.uint64)
        assert lib.zk_comm_all_gather_host(h, mine.ctypes.data, mine.nbytes, got.ctypes.data) == 0
        ok &= all(np.array_equal(got[r], np.arange(5, dtype=np.uint64) + 100 * r) for r in range(world))
        buf = (np.arange(3000, dtype=np.uint64) * 7 + 1) if rank == world - 1 else np.zeros(3000, dtype=np.uint64)     # 24 KB > one round
        assert lib.zk_comm_broadcast_host(h, buf.ctypes.data, buf.nbytes, world - 1) == 0
        ok &= bool(np.array_equal(buf, np.arange(3000, dtype=np.uint64) * 7 + 1))
        # all-to-all #1 of the level-3 prover: row blocks -> column shards (11 columns over the ranks), 2 KiB... 44 KiB pieces
        K, nb = 11, 512
        cols = _split(K, world)
        full = (np.arange(K * nb * world, dtype=np.uint64).reshape(K, nb * world) * 3 + 1)
        block = np.ascontiguousarray(full[:, rank * nb:(rank + 1) * nb])
        send = [np.ascontiguousarray(block[cols[p].start: cols[p].stop]) for p in range(world)]
        recv = [np.zeros((len(cols[rank]), nb), dtype=np.uint64) for _ in range(world)]
        assert _a2a(lib, h, send, recv) == 0
        values = np.stack(recv, axis=1).reshape(len(cols[rank]), nb * world)
        ok &= bool(np.array_equal(values, full[cols[rank].start: cols[rank].stop]))
        # all-to-all #2: column shards -> row shards
        send2 = [np.ascontiguousarray(values[:, p * nb:(p + 1) * nb]) for p in range(world)]
        rows = np.zeros((K, nb), dtype=np.uint64)
        recv2 = [rows[cols[p].start: cols[p].stop] for p in range(world)]
        assert _a2a(lib, h, send2, recv2) == 0
        ok &= bool(np.array_equal(rows, block))
        # a size the two ends disagree about is an error on the receiver, not silent truncation
        bad_send = [np.zeros(4 if rank == 0 else 2, dtype=np.uint64) for _ in range(world)]
        bad_recv = [np.zeros(2, dtype=np.uint64) for _ in range(world)]
        st = (C.c_uint64 * 3)()
        lib.zk_comm_stats(h, st)
        ok &= st[0] > 0 and st[1] > 0 and st[2] >= 4
        if absent is None:
            assert lib.zk_comm_barrier(h) == 0
            rc = _a2a(lib, h, bad_send, bad_recv)
            ok &= (rc != 0)                                                  # rank 0 sends 32 bytes where 16 are expected: everybody fails
            q.put((rank, bool(ok), 0))
        elif rank == absent:
            q.put((rank, bool(ok), 0))                                       # leaves without reaching the barrier
            q.close()
            q.join_thread()
            os._exit(0)
        else:
            rc = lib.zk_comm_barrier(h)                                      # the absent rank never arrives
            q.put((rank, bool(ok), rc))
        lib.zk_comm_free(h)
    except BaseException as e:                                               # noqa: BLE001
        q.put((rank, False, repr(e)))


def _make_stale_region(name, world, failed):
    """What a crashed earlier job with the same name leaves in /dev/shm: a fully initialised region (ShmHeader of csrc/comm_host.inc:
    magic, world, slot_bytes, arrived, sense, attached, detached, failed) that rank 0 never got to unlink."""
    import struct
    slot = 4096 * world
    head = (104 + world * world * 8 + 4095) & ~4095
    with open("/dev/shm/" + name, "wb") as f:
        f.write(struct.pack("<IIQIIIIi", 0x5A4B434D, world, slot, 0, 0, world, 0, failed))
        f.truncate(head + world * slot)


def _run(world, absent=None, env=None, stale_failed=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "zk_test_%d_%s" % (os.getpid(), os.urandom(4).hex())
    if stale_failed is not None:
        _make_stale_region(name, world, stale_failed)
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        procs = [ctx.Process(target=_worker, args=(r, world, name, q, absent, 1.5 if stale_failed is not None else 0.0)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {r[0]: r[1:] for r in res}


@pytest.mark.parametrize("world", [2, 4])
def test_host_transport_collectives(world):
    res = _run(world)
    assert res == {r: (True, 0) for r in range(world)}, res


def test_host_transport_absent_rank_is_a_timeout_not_a_hang():
    res = _run(2, absent=1, env={"ZK_COMM_TIMEOUT_S": "3"})
    assert res[1] == (True, 0)
    assert res[0][0] is True and res[0][1] == ZK_ERR_COMM, res


@pytest.mark.parametrize("failed", [0, 2])
def test_a_stale_region_of_a_crashed_job_is_not_joined(failed):
    """r05 advisor: a region left in /dev/shm by an earlier job with the same name passes the size and magic checks; a rank that
    opens it before rank 0 has replaced it used to sit at that region's barrier until the 300 s limit.  Rank 0 arrives 1.5 s late
    here: the other rank must end up in rank 0's region (it disowns the old one before unlinking the name) and every collective
    must work -- within seconds.  `failed` = the old job's failure word (0: it looks healthy, 2: it died noisily)."""
    import time
    t0 = time.time()
    res = _run(2, env={"ZK_COMM_TIMEOUT_S": "30"}, stale_failed=failed)
    assert res == {0: (True, 0), 1: (True, 0)}, res
    assert time.time() - t0 < 25


def test_comm_argument_checks():
    lib = _lib()
    h = C.c_void_p()
    assert lib.zk_comm_create_host(None, b"with/slash", 0, 2, 0, C.byref(h)) != 0
    assert lib.zk_comm_create_host(None, b"zk_three", 0, 3, 0, C.byref(h)) != 0              # not a power of two
    assert lib.zk_comm_create_host(None, b"zk_rank", 2, 2, 0, C.byref(h)) != 0
    assert lib.zk_comm_create_host(None, b"zk_one_%d" % os.getpid(), 0, 1, 0, C.byref(h)) == 0 and h
    x = np.arange(4, dtype=np.uint64)
    y = np.zeros(4, dtype=np.uint64)
    assert lib.zk_comm_all_gather_host(h, x.ctypes.data, x.nbytes, y.ctypes.data) == 0 and np.array_equal(x, y)
    assert lib.zk_comm_all_to_all_device(h, None, None, None, None) != 0                      # no ctx: host payloads only
    lib.zk_comm_free(h)
    lib.zk_comm_free(None)
