"""Host mirror of the reference's per-segment driver:

  * ``prove_with_traces`` / ``prove_with_commitments`` / ``prove_single_table``
    (evm_arithmetization/src/prover.rs:72-194, 213-298, 301-341),
  * ``observe_public_values`` (get_challenges.rs:11-227) with the limb helpers of util.rs:40-126,
  * starky ``get_ctl_data`` -> ``cross_table_lookup_data`` / ``ctl_helper_zs_cols`` and
    ``CrossTableLookup::num_ctl_helpers_zs_all`` ([EXT] starky 1.0.0 cross_table_lookup.rs),
  * the output containers ``AllProof`` / ``MultiProof`` / ``StarkProofWithMetadata`` / ``MemCap`` (proof.rs:29-54,
    587-622).

`prove_with_traces` is a binding of the compiled driver `zk_prove_segment` (csrc/segment_host.inc): this file
flattens the public values and the table / CTL definitions and copies the proof out; `get_ctl_data` and
`prove_single_table` expose the two inner steps for callers that keep their own loop.  There is no CPU fallback:
without the library / a GPU these functions raise."""
import ctypes as C
from dataclasses import dataclass, field
from itertools import groupby
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .all_stark import NUM_TABLES, OPTIONAL_TABLE_INDICES, TABLE_NAMES, AllStark, Table
from .challenger import Challenger
from .config import StarkConfig
from .polynomial_batch import PolynomialBatch
from .prover import CtlZData, StarkProof
from .stark import CrossTableLookup, ctl_partial_sums

P = 0xFFFFFFFF00000001


# ---- public values (proof.rs:70-91, 314-320, 357-363, 398-424, 471-487) ----------------------------------
@dataclass
class TrieRoots:
    state_root: bytes = bytes(32)
    transactions_root: bytes = bytes(32)
    receipts_root: bytes = bytes(32)


@dataclass
class BlockMetadata:
    block_beneficiary: bytes = bytes(20)
    block_timestamp: int = 0
    block_number: int = 0
    block_difficulty: int = 0
    block_random: bytes = bytes(32)
    block_gaslimit: int = 0
    block_chain_id: int = 0
    block_base_fee: int = 0
    block_gas_used: int = 0
    block_blob_gas_used: int = 0
    block_excess_blob_gas: int = 0
    parent_beacon_block_root: bytes = bytes(32)
    block_bloom: List[int] = field(default_factory=lambda: [0] * 8)


@dataclass
class BlockHashes:
    prev_hashes: List[bytes] = field(default_factory=lambda: [bytes(32)] * 256)
    cur_hash: bytes = bytes(32)


@dataclass
class ExtraBlockData:
    checkpoint_state_trie_root: bytes = bytes(32)
    checkpoint_consolidated_hash: List[int] = field(default_factory=lambda: [0] * 4)
    txn_number_before: int = 0
    txn_number_after: int = 0
    gas_used_before: int = 0
    gas_used_after: int = 0


@dataclass
class MemCap:
    """proof.rs:587-622: the MemBefore / MemAfter trace caps, one [U256; 4] per cap hash."""
    mem_cap: List[List[int]] = field(default_factory=list)

    @staticmethod
    def from_merkle_cap(cap: np.ndarray, hasher: int = 0) -> "MemCap":
        """proof.rs:606-621: `h.to_vec()` per cap hash.  `cap`: 32-byte digest slots as (n, 4) u64.  A Poseidon hash is
        its four elements; a Keccak-25 hash (`BytesHash<25>`) is four elements read from 7,7,7,4-byte little-endian
        chunks of the digest ([EXT] plonky2 hash_types.rs `BytesHash::to_vec`)."""
        slots = np.ascontiguousarray(np.asarray(cap, dtype=np.uint64).reshape(-1, 4))
        if hasher == 0:
            if (slots >= np.uint64(P)).any():
                raise ZkStarkError(-1, "non-canonical Poseidon digest in a Merkle cap")
            return MemCap([[int(x) for x in h] for h in slots])
        out = []
        for h in slots:
            b = h.tobytes()
            out.append([int.from_bytes(b[7 * k: 7 * k + (7 if k < 3 else 4)], "little") for k in range(4)])
        return MemCap(out)

    @staticmethod
    def from_elements(elems: np.ndarray) -> "MemCap":
        """from the output of zk_segment_proof_mem_caps (already `to_vec` elements)."""
        return MemCap([[int(x) for x in h] for h in np.asarray(elems, dtype=np.uint64).reshape(-1, 4)])


@dataclass
class PublicValues:
    trie_roots_before: TrieRoots = field(default_factory=TrieRoots)
    trie_roots_after: TrieRoots = field(default_factory=TrieRoots)
    block_metadata: BlockMetadata = field(default_factory=BlockMetadata)
    block_hashes: BlockHashes = field(default_factory=BlockHashes)
    extra_block_data: ExtraBlockData = field(default_factory=ExtraBlockData)
    mem_before: MemCap = field(default_factory=MemCap)
    mem_after: MemCap = field(default_factory=MemCap)
    # proof.rs:73-78: "Address to store the base fee to be burnt: only used when `cdk_erigon` is active" (a U256);
    # None = eth_mainnet.  Observed after extra_block_data (get_challenges.rs:146-154,211-219).
    burn_addr: Optional[int] = None
    # proof.rs:85-87 `RegistersData` before / after the segment (program_counter, is_kernel, stack_len, stack_top,
    # context, gas_used).  Not part of the transcript; they enter the Memory CTL as extra looking rows (verifier.rs).
    registers_before: dict = field(default_factory=dict)
    registers_after: dict = field(default_factory=dict)


class PublicValuesError(ZkStarkError):
    """`ProgramError::IntegerTooLarge` surfaced as "Invalid conversion of public values." (prover.rs:130)."""

    def __init__(self):
        super().__init__(-1, "Invalid conversion of public values.")


def u256_limbs(v: int) -> List[int]:
    """util.rs:101-113: eight little-endian 32-bit limbs."""
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def h256_limbs(h: bytes) -> List[int]:
    """util.rs:116-126: the hash read as a big-endian integer, in little-endian 32-bit limbs."""
    if len(h) != 32:
        raise PublicValuesError()
    return u256_limbs(int.from_bytes(h, "big"))


def u256_to_u32(v: int) -> int:
    if not 0 <= v < (1 << 32):
        raise PublicValuesError()
    return v


def u256_to_u64(v: int) -> Tuple[int, int]:
    if not 0 <= v < (1 << 64):
        raise PublicValuesError()
    return v & 0xFFFFFFFF, v >> 32


def public_values_elements(pv: PublicValues) -> List[int]:
    """The exact element sequence `observe_public_values` feeds the transcript (get_challenges.rs:195-218).
    `pv.burn_addr is None` = the eth_mainnet feature set; a `burn_addr` = the cdk_erigon build, where the three
    `#[cfg(feature = "eth_mainnet")]` block-metadata fields are not observed (:66-74; `features_check`,
    prover.rs:356-368, requires them to be zero) and the burn address is appended (:146-154)."""
    erigon = pv.burn_addr is not None
    out: List[int] = []
    for roots in (pv.trie_roots_before, pv.trie_roots_after):          # :21-29
        for r in (roots.state_root, roots.transactions_root, roots.receipts_root):
            out += h256_limbs(r)
    m = pv.block_metadata                                               # :46-83
    out += u256_limbs(int.from_bytes(m.block_beneficiary, "big"))[:5]
    out += [u256_to_u32(m.block_timestamp), u256_to_u32(m.block_number), u256_to_u32(m.block_difficulty)]
    out += h256_limbs(m.block_random)
    out += [u256_to_u32(m.block_gaslimit), u256_to_u32(m.block_chain_id)]
    out += list(u256_to_u64(m.block_base_fee))
    out.append(u256_to_u32(m.block_gas_used))
    if not erigon:
        out += list(u256_to_u64(m.block_blob_gas_used))
        out += list(u256_to_u64(m.block_excess_blob_gas))
        out += h256_limbs(m.parent_beacon_block_root)
    elif m.block_blob_gas_used or m.block_excess_blob_gas or any(m.parent_beacon_block_root):
        raise ZkStarkError(-1, "features_check: the eth_mainnet block-metadata fields must be zero in a cdk_erigon proof")
    for i in range(8):
        out += u256_limbs(m.block_bloom[i])
    if len(pv.block_hashes.prev_hashes) != 256:                         # :170-180
        raise PublicValuesError()
    for h in pv.block_hashes.prev_hashes:
        out += h256_limbs(h)
    out += h256_limbs(pv.block_hashes.cur_hash)
    e = pv.extra_block_data                                             # :114-128
    out += h256_limbs(e.checkpoint_state_trie_root)
    out += [int(x) % P for x in e.checkpoint_consolidated_hash]
    out += [u256_to_u32(e.txn_number_before), u256_to_u32(e.txn_number_after), u256_to_u32(e.gas_used_before),
            u256_to_u32(e.gas_used_after)]
    if pv.burn_addr is not None:                                        # observe_burn_addr (cdk_erigon)
        if not 0 <= pv.burn_addr < 1 << 256:
            raise PublicValuesError()
        out += u256_limbs(pv.burn_addr)
    return out


# memory/segments.rs:25-77 (unscaled) and cpu/kernel/constants/global_metadata.rs:7-77 ordinals used by the
# public-value writes of the Memory CTL
_SEG_GLOBAL_METADATA, _SEG_GLOBAL_BLOCK_BLOOM, _SEG_BLOCK_HASHES, _SEG_REGISTERS_STATES = 5, 24, 32, 33
_GM = dict(StateTrieRootDigestBefore=6, TransactionTrieRootDigestBefore=7, ReceiptTrieRootDigestBefore=8,
           StateTrieRootDigestAfter=9, TransactionTrieRootDigestAfter=10, ReceiptTrieRootDigestAfter=11,
           BlockBeneficiary=12, BlockTimestamp=13, BlockNumber=14, BlockDifficulty=15, BlockRandom=16, BlockGasLimit=17,
           BlockChainId=18, BlockBaseFee=19, BlockBlobGasUsed=20, BlockExcessBlobGas=21, BlockGasUsed=22,
           BlockGasUsedBefore=23, BlockGasUsedAfter=24, BlockCurrentHash=25, ParentBeaconBlockRoot=26,
           TxnNumberBefore=42, TxnNumberAfter=43, KernelHash=45, KernelLen=46, BurnAddr=53)
REGISTER_FIELDS = ("program_counter", "is_kernel", "stack_len", "stack_top", "context", "gas_used")
MEMORY_CTL_INDEX = 6                                                    # all_stark.rs:146-172: position of ctl_memory


def get_memory_extra_looking_values(pv: PublicValues, kernel_hash: int, kernel_len: int) -> List[List[int]]:
    """`verifier::debug_utils::get_memory_extra_looking_values` (verifier.rs:547-...): the Memory-CTL rows
    (is_read = 0, context 0, segment, index, eight 32-bit value limbs, timestamp 2) of the public values the kernel reads
    from memory -- block metadata, trie roots, block bloom, the 256 previous block hashes, the registers before / after.
    `KERNEL.code_hash` and `KERNEL.code.len()` are parameters (the assembled kernel is the caller's)."""
    be = lambda b: int.from_bytes(b, "big")
    m, e = pv.block_metadata, pv.extra_block_data
    erigon = pv.burn_addr is not None
    fields = [("BlockBeneficiary", be(m.block_beneficiary))]
    if erigon:
        fields.append(("BurnAddr", pv.burn_addr))
    fields += [("BlockTimestamp", m.block_timestamp), ("BlockNumber", m.block_number), ("BlockRandom", be(m.block_random)),
               ("BlockDifficulty", m.block_difficulty), ("BlockGasLimit", m.block_gaslimit), ("BlockChainId", m.block_chain_id),
               ("BlockBaseFee", m.block_base_fee)]
    if not erigon:
        fields.append(("ParentBeaconBlockRoot", be(m.parent_beacon_block_root)))
    fields += [("BlockCurrentHash", be(pv.block_hashes.cur_hash)), ("BlockGasUsed", m.block_gas_used)]
    if not erigon:
        fields += [("BlockBlobGasUsed", m.block_blob_gas_used), ("BlockExcessBlobGas", m.block_excess_blob_gas)]
    fields += [("TxnNumberBefore", e.txn_number_before), ("TxnNumberAfter", e.txn_number_after),
               ("BlockGasUsedBefore", e.gas_used_before), ("BlockGasUsedAfter", e.gas_used_after),
               ("StateTrieRootDigestBefore", be(pv.trie_roots_before.state_root)),
               ("TransactionTrieRootDigestBefore", be(pv.trie_roots_before.transactions_root)),
               ("ReceiptTrieRootDigestBefore", be(pv.trie_roots_before.receipts_root)),
               ("StateTrieRootDigestAfter", be(pv.trie_roots_after.state_root)),
               ("TransactionTrieRootDigestAfter", be(pv.trie_roots_after.transactions_root)),
               ("ReceiptTrieRootDigestAfter", be(pv.trie_roots_after.receipts_root)),
               ("KernelHash", kernel_hash), ("KernelLen", kernel_len)]
    writes = [(_SEG_GLOBAL_METADATA, _GM[k], v) for k, v in fields]
    writes += [(_SEG_GLOBAL_BLOCK_BLOOM, i, m.block_bloom[i]) for i in range(8)]
    writes += [(_SEG_BLOCK_HASHES, i, be(pv.block_hashes.prev_hashes[i])) for i in range(256)]
    for base, regs in ((0, pv.registers_before), (len(REGISTER_FIELDS), pv.registers_after)):
        writes += [(_SEG_REGISTERS_STATES, base + i, regs.get(f, 0)) for i, f in enumerate(REGISTER_FIELDS)]
    return [[0, 0, seg, idx] + u256_limbs(val) + [2] for seg, idx, val in writes]


def observe_public_values(challenger: Challenger, pv: PublicValues) -> None:
    challenger.observe_elements(public_values_elements(pv))


# ---- CTL data ([EXT] starky cross_table_lookup.rs) -------------------------------------------------------
def num_ctl_helpers_zs_all(ctls: Sequence[CrossTableLookup], table: int, num_challenges: int,
                           constraint_degree: int) -> Tuple[int, int, List[int]]:
    """`CrossTableLookup::num_ctl_helpers_zs_all` -> (total helper columns, total Z columns, helpers per CTL)."""
    num_helpers, num_ctls, by_ctl = 0, 0, [0] * len(ctls)
    for i, ctl in enumerate(ctls):
        n = sum(1 for t in [ctl.looked_table] + list(ctl.looking_tables) if t.table == table)
        if n > 1:
            by_ctl[i] = -(-n // (constraint_degree - 1))
            num_helpers += by_ctl[i]
        if n > 0:
            num_ctls += 1
    return num_helpers * num_challenges, num_ctls * num_challenges, by_ctl


def get_ctl_data(config: StarkConfig, trace_values: Sequence, ctls: Sequence[CrossTableLookup],
                 challenger: Challenger, constraint_degree: int, ctx=None):
    """starky `get_ctl_data` (prover.rs:134-144): draw `num_challenges` (beta, gamma) pairs, then for every CTL,
    every challenge: running-sum columns of each *run* of looking entries of one table (`ctl_helper_zs_cols`
    groups consecutive entries) and of the looked table.  Returns (challenges, z-data per table) in the order
    `cross_table_lookup_data` appends them -- that order is the order of the auxiliary polynomials."""
    ctl_challenges = [(challenger.get_challenge(), challenger.get_challenge()) for _ in range(config.num_challenges)]
    per_table: List[List[CtlZData]] = [[] for _ in trace_values]
    for ctl in ctls:
        looked = ctl.looked_table
        for beta, gamma in ctl_challenges:
            z_looked = ctl_partial_sums(trace_values[looked.table], [(looked.columns, looked.filter)], beta, gamma,
                                        constraint_degree, ctx=ctx)
            for table, group in groupby(ctl.looking_tables, key=lambda t: t.table):
                entries = [(t.columns, t.filter) for t in group]
                aux = ctl_partial_sums(trace_values[table], entries, beta, gamma, constraint_degree, ctx=ctx)
                # CtlZData.columns/filter: every looking entry of this table in the CTL
                all_entries = [(t.columns, t.filter) for t in ctl.looking_tables if t.table == table]
                per_table[table].append(CtlZData(beta, gamma, all_entries, aux))
            per_table[looked.table].append(CtlZData(beta, gamma, [(looked.columns, looked.filter)], z_looked))
    return ctl_challenges, per_table


# ---- proof containers --------------------------------------------------------------------------------------
@dataclass
class StarkProofWithMetadata:
    """prover.rs:335-338: the table proof plus the challenger state it started from."""
    proof: StarkProof
    init_challenger_state: np.ndarray


@dataclass
class MultiProof:
    stark_proofs: List[Optional[StarkProofWithMetadata]]
    ctl_challenges: List[Tuple[int, int]]


@dataclass
class AllProof:
    multi_proof: MultiProof
    public_values: PublicValues
    table_in_use: List[bool]

    def degree_bits(self) -> List[Optional[int]]:
        """proof.rs:58-64 (`recover_degree_bits`): from the FRI proof shape; here recorded by the prover."""
        return [p.proof.degree_bits if p is not None else None for p in self.multi_proof.stark_proofs]


def all_proof_to_words(p: AllProof) -> np.ndarray:
    """Flat u64 words of an `AllProof` minus the `PublicValues` the submitter already holds (only the two `MemCap`s, which
    the prover fills in, travel): what `scheduler.run_distributed` / `sharding` gather on rank 0 with tensor collectives."""
    from .collectives import pack_records
    mp = p.multi_proof
    mb = np.array(p.public_values.mem_before.mem_cap, dtype=np.uint64).reshape(-1)
    ma = np.array(p.public_values.mem_after.mem_cap, dtype=np.uint64).reshape(-1)
    head = np.array([len(mp.stark_proofs), len(mp.ctl_challenges)] + [int(bool(u)) for u in p.table_in_use] +
                    [x % (1 << 64) for bg in mp.ctl_challenges for x in bg], dtype=np.uint64)
    recs = [head, mb, ma]
    for sp in mp.stark_proofs:
        recs.append(np.zeros(0, np.uint64) if sp is None else sp.proof.to_words())
    return pack_records(recs)


def all_proof_from_words(w: np.ndarray, public_values: PublicValues) -> AllProof:
    from .collectives import unpack_records
    recs = unpack_records(w)
    head = recs[0]
    n_tab, nchal = int(head[0]), int(head[1])
    in_use = [bool(x) for x in head[2: 2 + n_tab]]
    cc = head[2 + n_tab: 2 + n_tab + 2 * nchal]
    public_values.mem_before = MemCap.from_elements(recs[1]) if recs[1].size else MemCap()
    public_values.mem_after = MemCap.from_elements(recs[2]) if recs[2].size else MemCap()
    proofs: List[Optional[StarkProofWithMetadata]] = []
    for r in recs[3: 3 + n_tab]:
        if r.size == 0:
            proofs.append(None)
        else:
            sp, _ = StarkProof.from_words(r)
            proofs.append(StarkProofWithMetadata(sp, sp.init_challenger_state))
    return AllProof(MultiProof(proofs, [(int(cc[2 * i]), int(cc[2 * i + 1])) for i in range(nchal)]), public_values, in_use)


class Aborted(ZkStarkError):
    def __init__(self):
        super().__init__(-4, "Stopping job from abort signal.")


def check_abort_signal(abort_signal) -> None:
    """prover.rs:346-354; `abort_signal` is any object with a truthy `.is_set()` or a 0/1 ctypes int."""
    if abort_signal is None:
        return
    flag = abort_signal.is_set() if hasattr(abort_signal, "is_set") else bool(getattr(abort_signal, "value", abort_signal))
    if flag:
        raise Aborted()


def prove_single_table(all_stark: AllStark, table: int, config: StarkConfig, trace_values, trace_commitment,
                       ctl_data: Sequence[CtlZData], ctl_challenges, challenger: Challenger,
                       abort_signal=None, aux_commitment=None) -> StarkProofWithMetadata:
    """prover.rs:301-341 for one table of `all_stark` (zk_prove_table: compact, then starky prove_with_commitment)."""
    from .prover import prove_single_table as _prove
    check_abort_signal(abort_signal)
    proof = _prove(all_stark.table_air[table], config, trace_values, trace_commitment, all_stark.lookups[table],
                   ctl_data, ctl_challenges, challenger, constraint_degree=all_stark.constraint_degree,
                   air_consts=all_stark.air_consts[table], aux_commitment=aux_commitment)
    return StarkProofWithMetadata(proof, proof.init_challenger_state)


class ZkTableIn(C.Structure):
    """include/zkstark.h zk_table_in"""
    _fields_ = [("d_trace", C.c_void_p), ("col_stride", C.c_size_t), ("n_cols", C.c_size_t), ("log_n", C.c_uint),
                ("air_id", C.c_uint32), ("air_consts", C.c_void_p), ("n_air_consts", C.c_size_t),
                ("lookup_program", C.c_void_p), ("lookup_words", C.c_size_t), ("in_use", C.c_int),
                ("optional", C.c_int)]


def encode_ctl_wiring(ctls: Sequence[CrossTableLookup]) -> np.ndarray:
    """`all_stark.cross_table_lookups` in the flat wiring encoding of zk_prove_segment:
    n_ctls, offset[n_ctls], per CTL: n_looking, (table, Entry) of the looked table, (table, Entry) x n_looking."""
    head = 1 + len(ctls)
    offs, payload = [], []

    def entry(t):
        w = [t.table, len(t.columns)]
        for c in t.columns:
            w += c.encode()
        return w + t.filter.encode()
    for ctl in ctls:
        offs.append(head + len(payload))
        payload.append(len(ctl.looking_tables))
        payload += entry(ctl.looked_table)
        for t in ctl.looking_tables:
            payload += entry(t)
    return np.array([len(ctls)] + offs + payload, dtype=np.uint64)


def segment_tables(all_stark: AllStark, trace_poly_values: Sequence, table_in_use: Sequence[bool], shapes=None, row_blocks=None):
    """The `zk_table_in` array of a segment, the encoded CTL wiring, and the arrays that must outlive the call.
    trace_poly_values[t] may be None for a table this rank does not hold (zk_prove_segment_table_parallel): its height then
    comes from shapes[t] = log_n.  row_blocks: {table: this rank's ROW BLOCK (C, n / W)} for row-sharded tables (log_n from shapes)."""
    from .prover import encode_lookup_set
    from .stark import _trace_args
    NUM_TABLES, TABLE_NAMES = all_stark.num_tables, all_stark.table_names
    # the encoded CTL wiring / lookup programs depend only on the table definitions: built once per AllStark
    cache = all_stark.__dict__.setdefault("_encoded", {})
    if "wiring" not in cache:
        cache["wiring"] = encode_ctl_wiring(all_stark.cross_table_lookups)
        cache["lookups"] = [encode_lookup_set(all_stark.lookups[t]) for t in range(all_stark.num_tables)]
    tables = (ZkTableIn * NUM_TABLES)()
    keep = []
    row_blocks = row_blocks or {}
    for t in range(NUM_TABLES):
        ti = tables[t]
        if t in row_blocks:
            blk = row_blocks[t]
            n_cols, stride, log_n = int(blk.shape[0]), (int(blk.stride(0)) if blk.shape[0] > 1 else int(blk.shape[1])), int(shapes[t])
            ti.d_trace = blk.data_ptr()
        elif trace_poly_values[t] is None:
            n_cols, stride, log_n = all_stark.table_columns[t], 0, int(shapes[t])
            ti.d_trace = None
        else:
            tr = trace_poly_values[t]
            n_cols, n, log_n, stride = _trace_args(tr)
            ti.d_trace = tr.data_ptr()
        if n_cols != all_stark.table_columns[t]:
            raise ZkStarkError(-1, "table %s: expected %d columns, got %d" % (TABLE_NAMES[t], all_stark.table_columns[t], n_cols))
        lp = cache["lookups"][t]
        ac = np.array(list(all_stark.air_consts[t]), dtype=np.uint64)
        keep += [lp, ac]
        ti.col_stride, ti.n_cols, ti.log_n = stride, n_cols, log_n
        ti.air_id = all_stark.table_air[t]
        ti.air_consts, ti.n_air_consts = (ac.ctypes.data if ac.size else None), ac.size
        ti.lookup_program, ti.lookup_words = (lp.ctypes.data, lp.size) if lp is not None else (None, 0)
        ti.in_use = 1 if table_in_use[t] else 0
        ti.optional = 1 if t in all_stark.optional_table_indices else 0
    return tables, cache["wiring"], keep


def segment_proof_from_handle(lib, h, all_stark: AllStark, config: StarkConfig, table_in_use, public_values: PublicValues,
                              timing: Optional[dict] = None) -> AllProof:
    """Copy a library-owned zk_segment_proof into an AllProof (the caller still frees the handle)."""
    from .prover import table_proof_from_handle
    NUM_TABLES = all_stark.num_tables
    nchal = config.num_challenges
    cc = np.zeros(2 * nchal, dtype=np.uint64)
    lib.zk_segment_proof_ctl_challenges(h, cc.ctypes.data, cc.size)
    ctl_challenges = [(int(cc[2 * i]), int(cc[2 * i + 1])) for i in range(nchal)]
    stark_proofs: List[Optional[StarkProofWithMetadata]] = []
    for t in range(NUM_TABLES):
        th = lib.zk_segment_proof_table(h, t)
        if not th:
            stark_proofs.append(None)
            continue
        p = table_proof_from_handle(lib, th)
        stark_proofs.append(StarkProofWithMetadata(p, p.init_challenger_state))
    nd = 1 << config.fri_config.cap_height
    mb, ma = np.zeros(4 * nd, dtype=np.uint64), np.zeros(4 * nd, dtype=np.uint64)
    lib.zk_segment_proof_mem_caps(h, mb.ctypes.data, ma.ctypes.data, 4 * nd)
    public_values.mem_before = MemCap.from_elements(mb)
    public_values.mem_after = MemCap.from_elements(ma)
    if timing is not None:
        ms = (C.c_double * (2 + NUM_TABLES))()
        lib.zk_segment_proof_stage_ms(h, ms, 2 + NUM_TABLES)
        timing["compute all trace commitments"] = timing.get("compute all trace commitments", 0.0) + ms[0] / 1e3
        timing["compute CTL data"] = timing.get("compute CTL data", 0.0) + ms[1] / 1e3
        for t in range(NUM_TABLES):
            if table_in_use[t]:
                k = "prove %s STARK" % all_stark.stark_field_names[t]     # prover.rs:232
                timing[k] = timing.get(k, 0.0) + ms[2 + t] / 1e3
    return AllProof(MultiProof(stark_proofs, ctl_challenges), public_values, list(table_in_use))


def prove_with_traces(all_stark: AllStark, config: StarkConfig, trace_poly_values: Sequence,
                      table_in_use: Sequence[bool], public_values: PublicValues, abort_signal=None,
                      hasher: Optional[int] = None, ctx=None, timing: Optional[dict] = None,
                      check_ctls: Optional[Tuple[int, int]] = None) -> AllProof:
    """prover.rs:72-194 as ONE C-ABI call (zk_prove_segment; the sequencing is compiled, csrc/segment_host.inc).
    `trace_poly_values[t]`: CUDA int64/uint64 tensor (columns, 2^k) -- the column-major
    `Vec<PolynomialValues<F>>` of table t (values may be non-canonical).  `abort_signal`: a ctypes
    c_int (or one-byte c_uint8 / c_bool, the reference's `AtomicBool` layout) that another thread may set (polled between
    kernels) or an object with `.is_set()` (checked on entry).
    `timing`: optional dict receiving the reference's TimingTree scopes in seconds.
    `check_ctls`: (kernel_hash, kernel_len) switches on the reference's debug-build `check_ctls` (prover.rs:164-184): the
    library verifies every cross-table lookup (Memory with the public values' extra looking rows) right after the CTL
    data and raises naming the unbalanced CTL instead of producing a proof the verifier would reject."""
    from .context import default_context
    NUM_TABLES = all_stark.num_tables
    if len(trace_poly_values) != NUM_TABLES or len(table_in_use) != NUM_TABLES:
        raise ZkStarkError(-1, "expected one trace and one in-use flag per table")
    if all_stark.cdk_erigon and public_values.burn_addr is None:
        raise ZkStarkError(-1, "There should be an address set in cdk_erigon.")      # get_challenges.rs:216-218
    check_abort_signal(abort_signal)
    hasher = config.hasher if hasher is None else hasher
    ctx = ctx or default_context(trace_poly_values[0].device.index or 0)
    ctx.use_torch_current_stream()
    if isinstance(abort_signal, (C.c_int, C.c_uint8, C.c_bool)):
        ctx.set_abort_flag(abort_signal)
    cfg = config.to_c()
    cfg.hasher = hasher
    pv = np.array(public_values_elements(public_values), dtype=np.uint64)      # may raise PublicValuesError
    tables, wiring, keep = segment_tables(all_stark, trace_poly_values, table_in_use)
    if check_ctls is not None:
        rows = np.array(get_memory_extra_looking_values(public_values, *check_ctls), dtype=np.uint64)
        ctx.check(ctx.lib.zk_ctx_set_check_ctls(ctx.handle, 1))
        ctx.check(ctx.lib.zk_ctx_set_ctl_extra_looking(ctx.handle, MEMORY_CTL_INDEX, rows.ctypes.data, rows.shape[0], rows.shape[1]))
    h = C.c_void_p()
    try:
        rc = ctx.lib.zk_prove_segment(ctx.handle, C.byref(cfg), C.cast(tables, C.c_void_p), NUM_TABLES,
                                      wiring.ctypes.data, wiring.size, pv.ctypes.data, pv.size,
                                      all_stark.constraint_degree, Table.MemBefore, Table.MemAfter, C.byref(h))
        if rc == -4:
            raise Aborted()
        ctx.check(rc)
    finally:
        if isinstance(abort_signal, (C.c_int, C.c_uint8, C.c_bool)):
            ctx.set_abort_flag(None)
        if check_ctls is not None:
            ctx.lib.zk_ctx_set_check_ctls(ctx.handle, 0)
    try:
        return segment_proof_from_handle(ctx.lib, h, all_stark, config, table_in_use, public_values, timing)
    finally:
        ctx.lib.zk_segment_proof_free(h)
