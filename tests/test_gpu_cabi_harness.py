"""-m gpu: the C ABI driven from plain C, no Python in the process (tests/cabi/harness.c, the role SURVEY 8(b) gives a
"C++ harness" and INTEGRATION.md gives the Rust `-sys` crate): built with gcc against include/zkstark.h, linked to the
in-tree library, its caps compared with tests/golden/commit_caps.json."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flags(hip_runtime=False):
    """compile / link flags of a plain-C driver: the in-tree library and (for the drivers that allocate device memory themselves) the
    HIP runtime -- or, in a process running the CPU emulation build (ZK_STARK_LIB = tests/emu/build*/libzkstark_emu*.so, tests/emu/), that
    library, which carries the stand-in runtime too"""
    emu = os.environ.get("ZK_STARK_LIB", "")
    if "libzkstark_emu" in os.path.basename(emu):
        d = os.path.dirname(emu)
        return ["-I", os.path.join(ROOT, "tests", "emu", "hipemu"), "-L", d, "-l:" + os.path.basename(emu), "-Wl,-rpath," + d]
    libdir = os.path.join(ROOT, "zk_evm_amd")
    out = ["-L", libdir, "-lzkstark_hip", "-Wl,-rpath," + libdir]
    if hip_runtime:
        out = ["-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] + out + ["-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    return out


def _build(tmp_path):
    exe = str(tmp_path / "cabi_harness")
    cmd = [shutil.which("gcc") or "gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cabi", "harness.c"),
           "-I", os.path.join(ROOT, "include"), *_flags(), "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_c_harness_matches_golden_caps(tmp_path):
    exe = _build(tmp_path)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "commit_caps.json")))["cases"]
    for c in cases:
        if c["log_n"] > 10:
            continue
        out = subprocess.run([exe] + [str(c[k]) for k in ("n_cols", "log_n", "rate_bits", "cap_height", "hasher", "seed")],
                             check=True, capture_output=True, text=True, timeout=300).stdout.splitlines()
        assert out[0].startswith("version zkstark-hip")
        cap = [int(x) for x in out[1].split()[1:]]
        assert cap == [x for row in c["cap"] for x in row], c["name"]
        assert out[2] == "memory log_n 3 unpadded 4"          # three operations + one timestamp-gap dummy, padded to 8 rows
        assert out[3].startswith('error "') and "cfg" in out[3]


def _fnv(h, words):
    for b in np.ascontiguousarray(words, dtype=np.uint64).reshape(-1).view(np.uint8).tolist():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("cdk_erigon", [False, True])
def test_c_segment_proof_matches_python_mirror(tmp_path, cdk_erigon):
    """tests/cabi/segment.c: a complete segment proof from plain C -- the generated registry header
    include/zk_all_stark.h, hipMalloc'd traces, zk_prove_segment -- equals, word for word (FNV-1a per table), the proof
    the Python mirror obtains for the same traces; both feature sets, one optional table absent."""
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from tests.test_gpu_segment import make_pv, make_traces, make_traces_cdk_erigon, to_public_values
    from zk_evm_amd.all_stark import AllStark
    exe = str(tmp_path / "cabi_segment")
    subprocess.run([shutil.which("gcc") or "gcc", "-std=c11", "-O1", "-Wall", "-Werror",
                    os.path.join(ROOT, "tests", "cabi", "segment.c"), "-I", os.path.join(ROOT, "include"),
                    *_flags(hip_runtime=True), "-o", exe], check=True)
    rng = np.random.default_rng(77 + cdk_erigon)
    traces = (make_traces_cdk_erigon if cdk_erigon else make_traces)(rng)
    in_use = [True] * len(traces)
    in_use[5] = False                                   # Logic absent: minimal all-zero trace, zero cap observed
    traces[5] = np.zeros((traces[5].shape[0], 16), dtype=np.uint64)
    labels = (11, 22, 33, 44)
    pvd = make_pv(rng)
    if cdk_erigon:                                      # features_check: no eth_mainnet block-metadata fields
        pvd.update(blob_gas_used=0, excess_blob_gas=0, parent_beacon_root=bytes(32))
    pv = to_public_values(pvd)
    if cdk_erigon:
        pv.burn_addr = 0x1234
    scfg = zk.StarkConfig(fri_config=zk.FriConfig(proof_of_work_bits=3, num_query_rounds=2))
    c = scfg.to_c()
    words = [int(cdk_erigon)] + [int(getattr(c, name)) for name, _ in c._fields_] + list(labels)
    elems = sg.public_values_elements(pv)
    words += [len(elems)] + [int(e) for e in elems]
    with open(tmp_path / "segment.bin", "wb") as f:
        f.write(np.array(words, dtype=np.uint64).tobytes())
        for t, used in zip(traces, in_use):
            f.write(np.array([int(used), t.shape[1].bit_length() - 1], dtype=np.uint64).tobytes())
            f.write(np.ascontiguousarray(t).tobytes())
    out = subprocess.run([exe, str(tmp_path / "segment.bin")], check=True, capture_output=True, text=True,
                         timeout=300).stdout.splitlines()
    st = AllStark(labels, cdk_erigon=cdk_erigon)
    got = sg.prove_with_traces(st, scfg, [torch.from_numpy(t.view(np.int64)).cuda() for t in traces], in_use, pv)
    assert out[0].split()[1:] == [str(x) for bg in got.multi_proof.ctl_challenges for x in bg]
    for t, line in enumerate(out[1:1 + st.num_tables]):
        sp = got.multi_proof.stark_proofs[t]
        if sp is None:
            assert line == f"table {st.table_names[t]} absent" and not in_use[t]
            continue
        p = sp.proof
        h = _fnv(0xCBF29CE484222325, sp.init_challenger_state)
        h = _fnv(h, p.trace_cap)
        if p.auxiliary_polys_cap is not None:
            h = _fnv(h, p.auxiliary_polys_cap)
        h = _fnv(_fnv(_fnv(h, p.quotient_polys_cap), p.openings), p.opening_proof)
        assert line.split()[1] == st.table_names[t] and line.split()[3] == str(p.degree_bits)
        assert line.split()[-1] == f"{h:016x}", (st.table_names[t], line)
    mb = np.array(got.public_values.mem_before.mem_cap, dtype=np.uint64)
    ma = np.array(got.public_values.mem_after.mem_cap, dtype=np.uint64)
    assert out[-1] == f"mem_caps fnv {_fnv(0xCBF29CE484222325, mb):016x} {_fnv(0xCBF29CE484222325, ma):016x}"


@pytest.mark.parametrize("world,wide,fri_mode", [(2, (3,), 0), (4, (3, 6), 1), (2, (), 0)])
def test_c_multi_rank_segment_proof_matches_single_gpu(tmp_path, world, wide, fri_mode):
    """tests/cabi/sharded.c: W processes forked from plain C -- no Python, no torch.distributed in them -- prove ONE segment
    together through `zk_prove_segment_table_parallel` on the library's communicator (host-staged transport: the ranks share
    this GPU), Keccak (and Memory: two lookups, one over a next-row column, CTL entries) ROW-SHARDED over all ranks: every
    rank's segment proof equals the single-GPU zk_prove_segment proof of the Python mirror, word for word (FNV-1a per table)."""
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from tests.test_gpu_segment import make_pv, make_traces, to_public_values
    from zk_evm_amd.all_stark import AllStark
    exe = str(tmp_path / "cabi_sharded")
    subprocess.run([shutil.which("gcc") or "gcc", "-std=c11", "-O1", "-Wall", "-Werror",
                    os.path.join(ROOT, "tests", "cabi", "sharded.c"), "-I", os.path.join(ROOT, "include"),
                    *_flags(hip_runtime=True), "-o", exe], check=True)
    rng = np.random.default_rng(91)
    log_ns = [9, 8, 10, 8, 8, 8, 11, 8, 8]
    traces = make_traces(rng, log_ns)
    in_use = [True] * 9
    in_use[8] = False
    labels = (11, 22, 33, 44)
    pv = to_public_values(make_pv(rng))
    scfg = zk.StarkConfig(fri_config=zk.FriConfig(proof_of_work_bits=3, num_query_rounds=4))
    c = scfg.to_c()
    words = [0] + [int(getattr(c, name)) for name, _ in c._fields_] + list(labels)
    elems = sg.public_values_elements(pv)
    words += [len(elems)] + [int(e) for e in elems]
    with open(tmp_path / "segment.bin", "wb") as f:
        f.write(np.array(words, dtype=np.uint64).tobytes())
        for t, used in zip(traces, in_use):
            f.write(np.array([int(used), t.shape[1].bit_length() - 1], dtype=np.uint64).tobytes())
            f.write(np.ascontiguousarray(t).tobytes())
    name = "zk_cabi_%d_%s" % (os.getpid(), os.urandom(3).hex())
    r = subprocess.run([exe, str(tmp_path / "segment.bin"), str(world), name, str(tmp_path / "out"), str(fri_mode)] + [str(t) for t in wide],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, ZK_COMM_TIMEOUT_S="120"))
    assert r.returncode == 0, r.stderr[-3000:]
    st = AllStark(labels)
    got = sg.prove_with_traces(st, scfg, [torch.from_numpy(t.view(np.int64)).cuda() for t in traces], in_use, pv)
    want = ["ctl_challenges " + " ".join(str(x) for bg in got.multi_proof.ctl_challenges for x in bg)]
    for t in range(st.num_tables):
        sp = got.multi_proof.stark_proofs[t]
        if sp is None:
            want.append(f"table {st.table_names[t]} absent")
            continue
        p = sp.proof
        h = _fnv(0xCBF29CE484222325, sp.init_challenger_state)
        h = _fnv(h, p.trace_cap)
        if p.auxiliary_polys_cap is not None:
            h = _fnv(h, p.auxiliary_polys_cap)
        h = _fnv(_fnv(_fnv(h, p.quotient_polys_cap), p.openings), p.opening_proof)
        want.append(f"{h:016x}")
    mb = np.array(got.public_values.mem_before.mem_cap, dtype=np.uint64)
    ma = np.array(got.public_values.mem_after.mem_cap, dtype=np.uint64)
    want.append(f"mem_caps fnv {_fnv(0xCBF29CE484222325, mb):016x} {_fnv(0xCBF29CE484222325, ma):016x}")
    for rank in range(world):
        out = open(str(tmp_path / "out") + ".%d" % rank).read().splitlines()
        assert out[0] == want[0], rank
        for t in range(st.num_tables):
            assert out[1 + t].split()[-1] == want[1 + t].split()[-1], (rank, st.table_names[t], out[1 + t])
        assert out[1 + st.num_tables] == want[-1]
        comm = out[-1].split()
        assert comm[:6] == ["comm", "host", "rank", str(rank), "of", str(world)] and int(comm[7]) > 0 and int(comm[9]) > 0
