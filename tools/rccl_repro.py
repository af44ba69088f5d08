#!/usr/bin/env python3
"""Minimal drill for the RCCL large-message corruption found in r04 (this image: RCCL 2.26.6, ROCm 7.0.2 inside torch): a send /
recv -- plain `dist.send` / `dist.recv`, `dist.all_to_all_single`, and the library's own `zk_comm_all_to_all_device` -- of
0.5 / 1.0 / 1.27 / 2.5 GB, to this rank itself and (with two or more ranks) to a peer on another GPU, each checked word for
word against a recomputable pattern.  With ONE rank (the only thing a gpurun box can run) RCCL returned corrupted data above
2^30 bytes for the self copy; whether real peers over xGMI are hit is what this script answers on the first multi-GPU box:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_repro.py
    python tools/rccl_repro.py                       # one rank (self copies only)

Prints ONE JSON line on rank 0: per size and per primitive `intact` (bool), the first bad word's byte offset if not, GB/s.
The library cuts its exchanges into pieces of 256 MiB (csrc/comm_host.inc, ZK_COMM_PIECE_MB): the `library` rows must always be
intact; `rccl_large_piece_intact_peer` / `_self` summarise the raw primitives above 1 GiB."""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES_GB = (0.5, 1.0, 1.27, 2.5)


def pattern(n_words, seed, dev):
    import torch
    x = torch.arange(n_words, dtype=torch.int64, device=dev)
    return (x * 6364136223846793005 + seed * 1442695040888963407) ^ (x >> 7)


def first_bad(got, want):
    import torch
    bad = torch.nonzero(got != want)
    return None if bad.numel() == 0 else int(bad[0].item()) * 8


def run(sizes_gb=SIZES_GB, with_library=True, group=None, ctx=None, host_group=None):
    """`group`: an nccl process group spanning all ranks (None = the default group, which then must be nccl)"""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    peer_to, peer_from = (rank + 1) % world, (rank - 1) % world
    rows = []
    lib_comm = None
    if with_library:
        try:
            import zk_evm_amd
            from zk_evm_amd.comm import Comm
            ctx = ctx or zk_evm_amd.Context(dev.index)
            ctx.use_torch_current_stream()
            lib_comm = Comm.from_group(ctx, group)
        except Exception as e:                       # the drill still answers the RCCL question without the library
            rows.append({"primitive": "library", "error": repr(e)[:200]})
    for gb in sizes_gb:
        n = int(gb * (1 << 30)) // 8
        send = pattern(n, 1000 * rank + 1, dev)
        recv = torch.zeros_like(send)

        def record(name, target, want_seed, fn):
            recv.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            err = None
            try:
                fn()
                torch.cuda.synchronize()
            except Exception as e:                   # noqa: BLE001
                err = repr(e)[:200]
            dt = time.perf_counter() - t0
            row = {"GB": gb, "bytes": n * 8, "primitive": name, "target": target, "rank": rank}
            if err:
                row["error"] = err
            else:
                want = pattern(n, want_seed, dev)
                off = first_bad(recv, want)
                row.update(intact=off is None, first_bad_byte=off, GBs=round(n * 8 / dt / 1e9, 1))
                del want
            rows.append(row)
        # ---- to this rank itself ------------------------------------------------------------------------------------------------
        def p2p_self():
            ops = [dist.P2POp(dist.isend, send, rank, group), dist.P2POp(dist.irecv, recv, rank, group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        record("send_recv", "self", 1000 * rank + 1, p2p_self)
        if world == 1:
            record("all_to_all_single", "self", 1000 * rank + 1, lambda: dist.all_to_all_single(recv, send, group=group))
        # ---- to a peer on another GPU -----------------------------------------------------------------------------------------
        if world > 1:
            def p2p_peer():
                ops = [dist.P2POp(dist.isend, send, peer_to, group), dist.P2POp(dist.irecv, recv, peer_from, group)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            record("send_recv", "peer", 1000 * peer_from + 1, p2p_peer)
            if n % world == 0:
                def a2a():
                    dist.all_to_all_single(recv, send, group=group)
                recv.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                a2a()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                k = n // world
                ok, off = True, None
                for p in range(world):                # block p of recv = block `rank` of rank p's send
                    want = pattern(n, 1000 * p + 1, dev)[rank * k:(rank + 1) * k]
                    o = first_bad(recv[p * k:(p + 1) * k], want)
                    if o is not None and ok:
                        ok, off = False, p * k * 8 + o
                rows.append({"GB": gb, "bytes": n * 8, "primitive": "all_to_all_single", "target": "peers", "rank": rank, "intact": ok,
                             "first_bad_byte": off, "GBs": round(n * 8 / dt / 1e9, 1)})
        # ---- the library's exchange (pieces of 256 MiB through ncclSend / ncclRecv) -----------------------------------------------
        if lib_comm is not None:
            import ctypes as C
            W = world
            sb = (C.c_size_t * W)(*[n * 8 if p == (rank if W == 1 else peer_to) else 0 for p in range(W)])
            rb = (C.c_size_t * W)(*[n * 8 if p == (rank if W == 1 else peer_from) else 0 for p in range(W)])
            sp = (C.c_void_p * W)(*[send.data_ptr()] * W)
            rp = (C.c_void_p * W)(*[recv.data_ptr()] * W)
            record("library zk_comm_all_to_all_device", "self" if W == 1 else "peer", 1000 * (rank if W == 1 else peer_from) + 1,
                   lambda: ctx.check(ctx.lib.zk_comm_all_to_all_device(lib_comm.handle, sp, sb, rp, rb)))
        del send, recv
        torch.cuda.empty_cache()
    if lib_comm is not None:
        lib_comm.close()
    # every rank's rows to rank 0 (host side)
    gathered = [None] * world
    hg = host_group if host_group is not None else group          # (bench.py: the result rows travel over its gloo group)
    dist.all_gather_object(gathered, rows) if hg is None else dist.all_gather_object(gathered, rows, group=hg)
    allrows = [r for part in gathered for r in part]
    big = [r for r in allrows if r.get("bytes", 0) > (1 << 30) and not r["primitive"].startswith("library") and "intact" in r]
    out = {"world": world, "rows": allrows,
           "rccl_large_piece_intact_self": all(r["intact"] for r in big if r["target"] == "self") if any(r["target"] == "self" for r in big) else None,
           "rccl_large_piece_intact_peer": all(r["intact"] for r in big if r["target"] != "self") if any(r["target"] != "self" for r in big) else None,
           "library_pieces_intact": all(r.get("intact", False) for r in allrows if r["primitive"].startswith("library")) if any(r["primitive"].startswith("library") for r in allrows) else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes-gb", type=str, default=",".join(str(s) for s in SIZES_GB))
    ap.add_argument("--no-library", action="store_true")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    out = run(tuple(float(x) for x in a.sizes_gb.split(",")), not a.no_library)
    if rank == 0:
        sys.stdout.flush()
        print("\n" + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
