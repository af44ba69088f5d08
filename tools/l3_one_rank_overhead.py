"""What the level-3 pipeline costs on ONE rank (RCCL group of one: the all-to-alls and all-gathers are device-local copies)
against the plain single-GPU table proof of the same trace -- the overhead that an N-rank run starts from.
Usage: python tools/l3_one_rank_overhead.py [log_n=18] [table=3 (Keccak)]"""
import json, os, socket, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
table = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import zk_evm_amd
import zk_evm_amd.prover as zp
from tools.benchlib import synthetic_segment_traces
from zk_evm_amd.all_stark import AllStark
from zk_evm_amd.challenger import Challenger
from zk_evm_amd.shard_prover import prove_table_row_sharded, table_ctl_specs
from zk_evm_amd.stark import ctl_partial_sums

st = AllStark((1, 2, 3, 4)); cfg = zk_evm_amd.StarkConfig()
log_ns = [4] * 9; log_ns[table] = log_n
tr = synthetic_segment_traces(log_ns, torch.device("cuda", 0), seed=5)[table]

def setup():
    ch = Challenger(cfg.hasher); ch.observe_elements(list(range(1, 40)))
    return ch, [(ch.get_challenge(), ch.get_challenge()) for _ in range(cfg.num_challenges)]

def sharded(fri):
    ch, chal = setup(); timing = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = prove_table_row_sharded(st.table_air[table], cfg, tr, table_ctl_specs(st, table, chal), chal, ch, constraint_degree=st.constraint_degree,
                                air_consts=st.air_consts[table], lookups=st.lookups[table], timing=timing, fri=fri)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, p, {k: round(v, 4) for k, v in timing.items() if isinstance(v, (int, float))}

def single():
    ch, chal = setup()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tb = zk_evm_amd.PolynomialBatch.from_values(tr, cfg.fri_config.rate_bits, False, cfg.fri_config.cap_height, hasher=cfg.hasher)
    zd = [zp.CtlZData(b, gm, e, ctl_partial_sums(tr, e, b, gm, st.constraint_degree)) for b, gm, e in table_ctl_specs(st, table, chal)]
    p = zp.prove_single_table(st.table_air[table], cfg, tr, tb, st.lookups[table], zd, chal, ch, constraint_degree=st.constraint_degree,
                              air_consts=st.air_consts[table])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tb.free()
    return dt, p

out = {"table": table, "cols": int(tr.shape[0]), "log_n": log_n}
def cks():
    return int(tr.sum().item()) & 0xffffffffffff
c0 = cks()
for rep in range(2):
    ts, ps = single()
    c1 = cks()
    t1, p1, tm1 = sharded("replicated")
    c2 = cks()
    t2, p2, tm2 = sharded("sharded")
out["trace_checksums"] = [c0, c1, c2, cks()]
for name in ("trace_cap", "auxiliary_polys_cap", "quotient_polys_cap", "openings", "opening_proof"):
    a, b = getattr(p1, name, None), getattr(ps, name, None)
    if a is not None and b is not None:
        out["same_" + name] = bool(np.array_equal(np.asarray(a), np.asarray(b)))
out.update(single_gpu_s=round(ts, 4), row_sharded_one_rank_s={"replicated_fri": round(t1, 4), "sharded_fri": round(t2, 4)},
           phases_replicated=tm1, phases_sharded=tm2,
           identical=bool(np.array_equal(p1.to_words(), ps.to_words()) and np.array_equal(p2.to_words(), ps.to_words())))
print(json.dumps(out))
dist.destroy_process_group()
