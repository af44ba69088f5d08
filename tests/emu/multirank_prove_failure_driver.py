"""TEST INFRASTRUCTURE (tests/test_emu_gpu_suite.py): one rank of a two-rank job on the CPU emulation build proving ONE table
row-sharded (zk_prove_table_sharded: about fifteen collectives -- all-to-alls, all-gathers of caps / carries / openings / FRI values).
Rank 1 has its n-th kernel launch refused, for several n spread over the proof.  Recorded per n: both ranks' status and how long the
call took -- the failing rank its own error, the other ZK_ERR_COMM, both at once -- and that the SAME communicator then proves the
table, with the same words on both ranks.
    python tests/emu/multirank_prove_failure_driver.py <rank> <world> <port>"""
import ctypes as C
import hashlib
import json
import os
import sys
import time


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zk_evm_amd
    from tests.test_gpu_multirank import _l3_setup, _l3_table
    from zk_evm_amd.comm import comm_for
    from zk_evm_amd.shard_prover import prove_table_row_sharded, table_ctl_specs
    emu = C.CDLL(os.environ["ZK_STARK_LIB"])
    emu.hipemu_fail_launch_at.argtypes = [C.c_long]
    shape = (12, 8, 7)                                   # MemBefore: 12 columns x 2^8 rows, a looking AND a looked table
    table = shape[2]
    ctx = zk_evm_amd.context.default_context(0)
    cm = comm_for(ctx, None)
    tr = _l3_table(shape)
    nb = tr.shape[1] // world
    block = tr[:, rank * nb:(rank + 1) * nb].contiguous()

    def prove():
        st, cfg, ch, chal = _l3_setup(table)
        p = prove_table_row_sharded(st.table_air[table], cfg, block, table_ctl_specs(st, table, chal), chal, ch,
                                    constraint_degree=st.constraint_degree, air_consts=st.air_consts[table], lookups=st.lookups[table], comm=cm)
        return hashlib.sha256(np.ascontiguousarray(p.to_words()).tobytes()).hexdigest()
    out = {"rank": rank, "reference": prove(), "runs": []}
    for n in [int(x) for x in os.environ.get("ZK_FAIL_AT", "1,5,12,20,30,45,60,80,110,150,5000").split(",")]:
        if rank == 1:
            emu.hipemu_fail_launch_at(n)
        t0 = time.time()
        code, digest = 0, None
        try:
            digest = prove()
        except zk_evm_amd.ZkStarkError as e:
            code = e.code
        dt = time.time() - t0
        emu.hipemu_fail_launch_at(0)
        dist.barrier()
        out["runs"].append({"fail_at": n, "code": code, "seconds": dt, "same_as_reference": digest == out["reference"] if digest else None,
                            "retry_same": prove() == out["reference"]})
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
