"""oracle/segment.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

CPU restatement of the reference's per-segment driver, evm_arithmetization/src/prover.rs:72-298
(`prove_with_traces`, `prove_with_commitments`, `prove_single_table` :301-341), of the transcript encoding of
the public values (get_challenges.rs:11-227, util.rs:40-126) and of starky 1.0.0 `get_ctl_data` /
`cross_table_lookup_data` / `verify_cross_table_lookups` ([EXT] cross_table_lookup.rs), on top of the C oracle
and oracle/stark_prover.py.  Small sizes only."""
import ctypes as C

import numpy as np

from . import all_stark as A
from . import stark as S
from . import stark_prover as SP

P = S.P


def pv_elements(pv: dict):
    """pv: plain dict {roots_before: [3 x 32B], roots_after: [3 x 32B], beneficiary: 20B, timestamp, number,
    difficulty, random: 32B, gaslimit, chain_id, base_fee, gas_used, blob_gas_used, excess_blob_gas,
    parent_beacon_root: 32B, bloom: [8 ints], prev_hashes: [256 x 32B], cur_hash: 32B, checkpoint_root: 32B,
    checkpoint_hash: [4], txn_before, txn_after, gas_before, gas_after}."""
    def limbs(v): return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    def h(b): return limbs(int.from_bytes(b, "big"))          # util.rs:116-126 == observe_root (:11-19)
    def u32(v):
        assert 0 <= v < 1 << 32, "IntegerTooLarge"
        return [v]
    def u64(v):
        assert 0 <= v < 1 << 64, "IntegerTooLarge"
        return [v & 0xFFFFFFFF, v >> 32]
    e = []
    for r in pv["roots_before"] + pv["roots_after"]:
        e += h(r)
    e += limbs(int.from_bytes(pv["beneficiary"], "big"))[:5]
    e += u32(pv["timestamp"]) + u32(pv["number"]) + u32(pv["difficulty"]) + h(pv["random"])
    e += u32(pv["gaslimit"]) + u32(pv["chain_id"]) + u64(pv["base_fee"]) + u32(pv["gas_used"])
    if pv.get("burn_addr") is None:                           # #[cfg(feature = "eth_mainnet")], get_challenges.rs:66-74
        e += u64(pv["blob_gas_used"]) + u64(pv["excess_blob_gas"]) + h(pv["parent_beacon_root"])
    for b in pv["bloom"]:
        e += limbs(b)
    assert len(pv["prev_hashes"]) == 256
    for b in pv["prev_hashes"]:
        e += h(b)
    e += h(pv["cur_hash"]) + h(pv["checkpoint_root"]) + [x % P for x in pv["checkpoint_hash"]]
    e += u32(pv["txn_before"]) + u32(pv["txn_after"]) + u32(pv["gas_before"]) + u32(pv["gas_after"])
    if pv.get("burn_addr") is not None:                       # observe_burn_addr, cdk_erigon (get_challenges.rs:146-154)
        e += limbs(pv["burn_addr"])
    return e


def cross_table_lookup_data(traces, ctls, challenges, constraint_degree):
    """[EXT] starky cross_table_lookup_data: per table, the CtlZData list in append order.  `ctl_helper_zs_cols`
    groups *consecutive* looking entries by table; the z-data's own (columns, filter) list is every looking entry
    of that table."""
    per_table = [[] for _ in traces]
    for ctl in ctls:
        for ch in challenges:
            runs = []
            for t in ctl.looking_tables:
                if runs and runs[-1][0] == t.table:
                    runs[-1][1].append(t)
                else:
                    runs.append((t.table, [t]))
            for table, _ in runs:
                entries = [(t.columns, t.filter) for t in ctl.looking_tables if t.table == table]
                per_table[table].append(S.CtlZData(ch, entries, 0))
            lk = ctl.looked_table
            per_table[lk.table].append(S.CtlZData(ch, [(lk.columns, lk.filter)], 0))
    return per_table


def mem_cap_from_merkle_cap(cap, hasher):
    """`MemCap::from_merkle_cap` (proof.rs:606-621): `h.to_vec()` of every cap hash.  PoseidonHash out = its four
    elements; KeccakHash<25> out (`BytesHash<25>`) = four elements from 7,7,7,4-byte little-endian chunks ([EXT]
    plonky2 1.0.0 hash/hash_types.rs `impl GenericHashOut for BytesHash<N>`: `chunks(7)` + `from_noncanonical_u64`)."""
    slots = np.ascontiguousarray(np.asarray(cap, dtype=np.uint64).reshape(-1, 4))
    if hasher == 0:
        return slots.copy()
    out = np.zeros_like(slots)
    for i, h in enumerate(slots):
        b = h.tobytes()[:25]
        out[i] = [int.from_bytes(b[7 * k: 7 * k + 7], "little") for k in range(4)]
    return out


def prove_with_traces(o, fri_api, cfg, traces, table_in_use, pv, cpu_air_consts, ctls=None, lookups=None, reg=None,
                      fast=False):
    """traces: list of 9 (C_t, n_t) uint64 arrays (10 with reg = A.Registry(cdk_erigon=True)).  Returns
    dict(ctl_challenges, proofs (None if unused), init_states, mem_before, mem_after, trace_caps).
    fast: run the per-row loops in C (oracle/fast_stark.py: the same restatements traced to tapes) -- any size."""
    from . import airs
    L = o.lib
    reg = reg or A.Registry(False)
    ctls = ctls if ctls is not None else reg.ctls
    lookups = lookups if lookups is not None else reg.lookups
    commits = [o.commit_values(t, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, hasher=cfg.hasher) for t in traces]
    och = fri_api.new_challenger(o, cfg.hasher)
    for i, c in enumerate(commits):
        if i in reg.OPTIONAL_TABLES and not table_in_use[i]:
            z = np.zeros(c["cap"].size, dtype=np.uint64)
            L.orc_challenger_observe(C.byref(och), z, z.size)
        else:
            L.orc_challenger_observe_cap(C.byref(och), c["cap"], c["cap"].shape[0])
    e = np.array(pv_elements(pv), dtype=np.uint64)
    L.orc_challenger_observe(C.byref(och), e, e.size)
    challenges = []
    for _ in range(cfg.num_challenges):
        b = L.orc_challenger_get(C.byref(och))
        g = L.orc_challenger_get(C.byref(och))
        challenges.append(S.GrandProductChallenge(b, g))
    ctl_pairs = [(c.beta, c.gamma) for c in challenges]
    per_table = cross_table_lookup_data(traces, ctls, challenges, 3)
    proofs, inits = [], []
    for t in range(reg.NUM_TABLES):
        if not table_in_use[t]:
            proofs.append(None)
            inits.append(None)
            continue
        st = np.zeros(12, dtype=np.uint64)
        L.orc_challenger_compact(C.byref(och), st)
        inits.append(st)
        air = airs.AIRS[reg.TABLE_AIR[t]][0] if t != A.CPU else airs.make_eval_cpu(*cpu_air_consts, cdk_erigon=reg.cdk_erigon)
        if fast:
            from . import fast_stark as FS
            prove = FS.prove_with_commitment
        else:
            prove = SP.prove_with_commitment
        proofs.append(prove(o, fri_api, cfg, air, traces[t], commits[t], lookups[t], per_table[t], ctl_pairs, och))
    mem_after = commits[A.MEM_AFTER]["cap"].copy()
    if not table_in_use[A.MEM_AFTER]:
        mem_after[:] = 0
    return dict(ctl_challenges=ctl_pairs, proofs=proofs, init_states=inits,
                mem_before=mem_cap_from_merkle_cap(commits[A.MEM_BEFORE]["cap"], cfg.hasher),
                mem_after=mem_cap_from_merkle_cap(mem_after, cfg.hasher), trace_caps=[c["cap"] for c in commits], final_challenge=L.orc_challenger_get(C.byref(och)))


def verify_cross_table_lookups(ctls, ctl_zs_first, extra_looking_sums, num_challenges):
    """[EXT] starky verify_cross_table_lookups (called from verifier.rs:307): for each CTL and challenge, the sum of
    the looking tables' Z(first row) plus the extra looking sum must equal the looked table's Z(first row).
    ctl_zs_first[t]: list of that table's opened Z(1) values in z-data order; extra_looking_sums[ctl][challenge].
    -> (ok, reason)."""
    it = [iter(z) for z in ctl_zs_first]
    for idx, ctl in enumerate(ctls):
        extra = extra_looking_sums[idx] if extra_looking_sums else [0] * num_challenges
        tables = []
        for t in ctl.looking_tables:
            if t.table not in tables:
                tables.append(t.table)
        for c in range(num_challenges):
            looking = sum(next(it[t]) for t in tables) % P
            looked = next(it[ctl.looked_table.table])
            if (looking + extra[c]) % P != looked % P:
                return False, "CTL %d challenge %d" % (idx, c)
    for i in it:
        if next(i, None) is not None:
            return False, "unconsumed Z openings"
    return True, ""


# ---- verifier.rs:184-312 `verify_proof` and :319-512 `get_memory_extra_looking_sum` / `add_data_write` ----
SEG_GLOBAL_METADATA, SEG_GLOBAL_BLOCK_BLOOM, SEG_BLOCK_HASHES, SEG_REGISTERS_STATES = 5, 24, 32, 33   # segments.rs:25-77
# GlobalMetadata ordinals (cpu/kernel/constants/global_metadata.rs:7-77, `unscale()`d)
GM = dict(StateTrieRootDigestBefore=6, TransactionTrieRootDigestBefore=7, ReceiptTrieRootDigestBefore=8,
          StateTrieRootDigestAfter=9, TransactionTrieRootDigestAfter=10, ReceiptTrieRootDigestAfter=11,
          BlockBeneficiary=12, BlockTimestamp=13, BlockNumber=14, BlockDifficulty=15, BlockRandom=16,
          BlockGasLimit=17, BlockChainId=18, BlockBaseFee=19, BlockBlobGasUsed=20, BlockExcessBlobGas=21,
          BlockGasUsed=22, BlockGasUsedBefore=23, BlockGasUsedAfter=24, BlockCurrentHash=25,
          ParentBeaconBlockRoot=26, TxnNumberBefore=42, TxnNumberAfter=43, KernelHash=45, KernelLen=46, BurnAddr=53)
REGISTER_FIELDS = ("program_counter", "is_kernel", "stack_len", "stack_top", "context", "gas_used")
MEMORY_CTL_IDX = 6                                                                     # all_stark.rs:146


def public_memory_writes(pv: dict, kernel_hash: int, kernel_len: int):
    """The (segment, index, U256 value) list `get_memory_extra_looking_sum` turns into Memory writes
    (verifier.rs:330-494, eth_mainnet feature set).  `KERNEL.code_hash` / `KERNEL.code.len()` are parameters: the
    assembled kernel is outside this path's scope.  pv as in `pv_elements`, plus optional registers_before/after
    dicts keyed by REGISTER_FIELDS (default 0)."""
    be = lambda b: int.from_bytes(b, "big")
    rb, ra, rc = pv["roots_before"], pv["roots_after"], None
    fields = [("BlockBeneficiary", be(pv["beneficiary"])), ("BlockTimestamp", pv["timestamp"]),
              ("BlockNumber", pv["number"]), ("BlockRandom", be(pv["random"])), ("BlockDifficulty", pv["difficulty"]),
              ("BlockGasLimit", pv["gaslimit"]), ("BlockChainId", pv["chain_id"]), ("BlockBaseFee", pv["base_fee"]),
              ("ParentBeaconBlockRoot", be(pv["parent_beacon_root"])), ("BlockCurrentHash", be(pv["cur_hash"])),
              ("BlockGasUsed", pv["gas_used"]), ("BlockBlobGasUsed", pv["blob_gas_used"]),
              ("BlockExcessBlobGas", pv["excess_blob_gas"]), ("BurnAddr", pv.get("burn_addr")),
              ("TxnNumberBefore", pv["txn_before"]),
              ("TxnNumberAfter", pv["txn_after"]), ("BlockGasUsedBefore", pv["gas_before"]),
              ("BlockGasUsedAfter", pv["gas_after"]),
              ("StateTrieRootDigestBefore", be(rb[0])), ("TransactionTrieRootDigestBefore", be(rb[1])),
              ("ReceiptTrieRootDigestBefore", be(rb[2])), ("StateTrieRootDigestAfter", be(ra[0])),
              ("TransactionTrieRootDigestAfter", be(ra[1])), ("ReceiptTrieRootDigestAfter", be(ra[2])),
              ("KernelHash", kernel_hash), ("KernelLen", kernel_len)]
    erigon = pv.get("burn_addr") is not None                  # verifier.rs:334-340 (cdk_erigon) vs :370-385 (eth_mainnet)
    skip = ("ParentBeaconBlockRoot", "BlockBlobGasUsed", "BlockExcessBlobGas") if erigon else ("BurnAddr",)
    w = [(SEG_GLOBAL_METADATA, GM[k], v) for k, v in fields if k not in skip]
    w += [(SEG_GLOBAL_BLOCK_BLOOM, i, pv["bloom"][i]) for i in range(8)]
    w += [(SEG_BLOCK_HASHES, i, be(pv["prev_hashes"][i])) for i in range(256)]
    for base, key in ((0, "registers_before"), (len(REGISTER_FIELDS), "registers_after")):
        regs = pv.get(key, {})
        w += [(SEG_REGISTERS_STATES, base + i, regs.get(f, 0)) for i, f in enumerate(REGISTER_FIELDS)]
    return w


def get_memory_extra_looking_sum(pv, challenge, kernel_hash, kernel_len):
    """verifier.rs:319-512: sum over the public-value writes of 1 / combine(is_read=0, ctx=0, segment, index,
    value limbs[8], timestamp=2)."""
    total = 0
    for seg, idx, val in public_memory_writes(pv, kernel_hash, kernel_len):
        row = [0, 0, seg, idx] + [(val >> (32 * j)) & 0xFFFFFFFF for j in range(8)] + [2]
        total = (total + S.inv(challenge.combine(row))) % P
    return total


def verify_proof(o, fri_api, cfg, stark_proofs, table_in_use, pv, cpu_air_consts, kernel_hash, kernel_len,
                 is_initial=False, initial_mem_cap=None, mem_before_cap=None, ctls=None, lookups=None, reg=None):
    """verifier.rs:184-312 (+ get_challenges.rs:270-312).  stark_proofs[t]: None or dict(trace_cap, aux_cap,
    quotient_cap, openings, fri, degree_bits).  -> (ok, reason).  `initial_mem_cap` stands for
    initial_memory_merkle_cap (verifier.rs:14-78) of the kernel image in use; compared with `mem_before_cap`
    (public_values.mem_before.mem_cap) when is_initial."""
    from . import airs
    L = o.lib
    reg = reg or A.Registry(False)
    ctls = ctls if ctls is not None else reg.ctls
    lookups = lookups if lookups is not None else reg.lookups
    och = fri_api.new_challenger(o, cfg.hasher)
    for t, sp in enumerate(stark_proofs):
        if sp is not None:
            cap = np.ascontiguousarray(sp["trace_cap"])
            L.orc_challenger_observe_cap(C.byref(och), cap, cap.shape[0])
        else:
            if t not in reg.OPTIONAL_TABLES or table_in_use[t]:
                return False, "missing stark_proof for table %d" % t
            z = np.zeros((1 << cfg.cap_height) * 4, dtype=np.uint64)
            L.orc_challenger_observe(C.byref(och), z, z.size)
    try:
        e = np.array(pv_elements(pv), dtype=np.uint64)
    except AssertionError:
        return False, "Invalid sampling of proof challenges."
    L.orc_challenger_observe(C.byref(och), e, e.size)
    chal = [S.GrandProductChallenge(L.orc_challenger_get(C.byref(och)), L.orc_challenger_get(C.byref(och)))
            for _ in range(cfg.num_challenges)]
    pairs = [(c.beta, c.gamma) for c in chal]
    per_table = cross_table_lookup_data([None] * reg.NUM_TABLES, ctls, chal, 3)
    from . import stark_verifier as V
    zs_first = []
    for t, sp in enumerate(stark_proofs):
        zd = per_table[t]
        for z in zd:                                     # num_ctl_helpers_zs_all: ceil(k / (degree - 1)) helpers, k > 1
            k = len(z.columns_filters)
            z.n_helpers = -(-k // 2) if k > 1 else 0
        if sp is None:
            zs_first.append([0] * len(zd))
            continue
        st = np.zeros(12, dtype=np.uint64)
        L.orc_challenger_compact(C.byref(och), st)
        air = airs.AIRS[reg.TABLE_AIR[t]][0] if t != A.CPU else airs.make_eval_cpu(*cpu_air_consts, cdk_erigon=reg.cdk_erigon)
        ok, why = V.verify_stark_proof(o, fri_api, cfg, air, reg.TABLE_COLUMNS[t], sp["degree_bits"], lookups[t], zd,
                                       pairs, sp, och)
        if not ok:
            return False, "table %d: %s" % (t, why)
        opn = np.ascontiguousarray(sp["openings"], dtype=np.uint64).reshape(-1)
        first = opn[opn.size - 2 * len(zd):].reshape(-1, 2) if zd else np.zeros((0, 2), dtype=np.uint64)
        if any(int(b) for _, b in first):
            return False, "table %d: ctl_zs_first outside the base field" % t
        zs_first.append([int(a) for a, _ in first])      # ctl_zs_first: base-field openings at 1
    if is_initial:
        # verify_initial_memory (verifier.rs:149-170): `hash1.to_vec()` of initial_memory_merkle_cap limb by limb
        # against public_values.mem_before.mem_cap (itself `to_vec` elements, proof.rs:606-621)
        if initial_mem_cap is None or mem_before_cap is None or not np.array_equal(
                mem_cap_from_merkle_cap(initial_mem_cap, cfg.hasher), np.asarray(mem_before_cap, dtype=np.uint64)):
            return False, "Invalid initial MemBefore Merkle cap."
    extra = [[0] * cfg.num_challenges for _ in ctls]
    extra[MEMORY_CTL_IDX] = [get_memory_extra_looking_sum(pv, c, kernel_hash, kernel_len) for c in chal]
    return verify_cross_table_lookups(ctls, zs_first, extra, cfg.num_challenges)
