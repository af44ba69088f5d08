"""TEST INFRASTRUCTURE (tests/test_emu_gpu_suite.py, tools/emu_sanitizers.sh): one process, an emulated node with two devices
(HIPEMU_DEVICES=2), `SegmentScheduler(devices=[0, 1])` -- a worker thread, a zk_ctx and a stream per device -- proving small segments;
prints whether every proof equals the one a single ctx makes, and how many segments each device proved."""
import json, sys
import numpy as np, torch
import zk_evm_amd
import zk_evm_amd.segment as sg
from tests.gpu_util import to_dev
from tests.test_gpu_segment import make_pv, make_traces, to_public_values
from zk_evm_amd.all_stark import AllStark
from zk_evm_amd.scheduler import SegmentJob, SegmentScheduler
st = AllStark((1, 2, 3, 4))
cfg = zk_evm_amd.StarkConfig(fri_config=zk_evm_amd.FriConfig(num_query_rounds=5, proof_of_work_bits=4))
host = [(make_traces(np.random.default_rng(100 + i)), make_pv(np.random.default_rng(200 + i))) for i in range(2)]
words = lambda p: sg.all_proof_to_words(p)
direct = [words(sg.prove_with_traces(st, cfg, [to_dev(t) for t in tr], [True] * 9, to_public_values(pv))) for tr, pv in host]
jobs = [SegmentJob(lambda dev, tr=tr: [to_dev(t) for t in tr], [True] * 9, to_public_values(pv), tag=i) for i, (tr, pv) in enumerate(host)]
with SegmentScheduler(st, cfg, devices=[0, 1], in_flight=1) as sch:
    got = sch.map(jobs)
    per_device = {s.device: s.segments for s in sch.stats}
same = all(np.array_equal(d, words(g)) for d, g in zip(direct, got))
print("RESULT " + json.dumps({"same": bool(same), "per_device": per_device}))
