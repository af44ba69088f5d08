"""Host mirror of the plonky2 PLONK prover interface for the recursion layer:
`CircuitData::prove` as the reference calls it on every `StarkWrapperCircuit` / `PlonkWrapperCircuit` / root circuit
(evm_arithmetization/src/fixed_recursive_verifier.rs:2146, 3167-3179).  Marshalling only: circuit data and witness in,
`ProofWithPublicInputs`-shaped flat proof out; the protocol runs in csrc/plonk_host.inc + csrc/plonk.cuh.  No CPU
fallback: without the library / a GPU every call raises."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ._lib import ZkStarkError
from .config import ZkCfg

GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC = 0, 1, 2, 3


class ZkPlonkGate(C.Structure):
    """include/zkstark.h zk_plonk_gate"""
    _fields_ = [(n, C.c_uint32) for n in ("kind", "param", "selector_index", "group_start", "group_end")]


class ZkPlonkCommon(C.Structure):
    """include/zkstark.h zk_plonk_common"""
    _fields_ = [(n, C.c_uint32) for n in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors",
                                          "quotient_degree_factor", "num_gate_constraints")] + [("fri", ZkCfg)]


class ZkPlonkProofView(C.Structure):
    """include/zkstark.h zk_plonk_proof_view"""
    _fields_ = [("cap_digests", C.c_size_t), ("wires_cap", C.c_void_p), ("plonk_zs_partial_products_cap", C.c_void_p),
                ("quotient_polys_cap", C.c_void_p), ("openings", C.c_void_p), ("n_openings", C.c_size_t),
                ("opening_proof", C.c_void_p), ("proof_words", C.c_size_t), ("public_inputs_hash", C.c_uint64 * 4),
                ("stage_ms", C.c_double * 6)]


@dataclass(frozen=True)
class CircuitConfig:
    """`CircuitConfig::standard_recursion_config()` (the defaults) -- the fields the prover consumes."""
    num_wires: int = 135
    num_routed_wires: int = 80
    num_challenges: int = 2
    max_quotient_degree_factor: int = 8
    rate_bits: int = 3
    cap_height: int = 4
    proof_of_work_bits: int = 16
    num_query_rounds: int = 28
    arity_bits: int = 4
    final_poly_bits: int = 5
    hasher: int = 0                     # 0 = PoseidonGoldilocksConfig, 1 = KeccakGoldilocksConfig (Merkle trees and transcript;
                                        # the public-input hash is C::InnerHasher = Poseidon in both)

    def fri_cfg(self) -> ZkCfg:
        return ZkCfg(rate_bits=self.rate_bits, cap_height=self.cap_height, hasher=self.hasher, num_challenges=self.num_challenges,
                     proof_of_work_bits=self.proof_of_work_bits, num_query_rounds=self.num_query_rounds,
                     arity_bits=self.arity_bits, final_poly_bits=self.final_poly_bits)


@dataclass
class ProofWithPublicInputs:
    """plonky2 `ProofWithPublicInputs` in the flat layout of zk_plonk_proof_view."""
    wires_cap: np.ndarray
    plonk_zs_partial_products_cap: np.ndarray
    quotient_polys_cap: np.ndarray
    openings: np.ndarray            # (n, 2): constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys
    opening_proof: np.ndarray       # (at zeta), then plonk_zs_next (at g * zeta); flat FriProof over the four oracles
    public_inputs: List[int]
    public_inputs_hash: List[int]
    stage_ms: dict


class CircuitData:
    """Prover-side `CircuitData`: owns the `constants_sigmas_commitment` in HBM (committed once, reused by every proof of
    the circuit -- the recursion layer proves the same handful of circuits for every segment)."""

    def __init__(self, config: CircuitConfig, degree_bits: int, gates: Sequence[tuple], num_selectors: int,
                 constants_sigmas, k_is: Sequence[int], circuit_digest: Sequence[int], num_gate_constraints: int,
                 quotient_degree_factor: int = 8, ctx=None):
        """gates: (kind, param, selector_index, group_start, group_end) per gate in `common_data.gates` order;
        constants_sigmas: CUDA int64/uint64 tensor (num_constants + num_routed_wires, 2^degree_bits) of VALUES."""
        from .context import default_context
        from .stark import _trace_args
        self.ctx = ctx or default_context(constants_sigmas.device.index or 0)
        self.ctx.use_torch_current_stream()
        self.config, self.degree_bits = config, degree_bits
        n_cols, n, log_n, stride = _trace_args(constants_sigmas)
        if log_n != degree_bits or n_cols <= config.num_routed_wires:
            raise ZkStarkError(-1, "constants_sigmas must have (num_constants + num_routed_wires) columns of 2^degree_bits rows")
        self.num_constants = n_cols - config.num_routed_wires
        common = ZkPlonkCommon(degree_bits, config.num_wires, config.num_routed_wires, self.num_constants, num_selectors,
                               quotient_degree_factor, num_gate_constraints, config.fri_cfg())
        garr = (ZkPlonkGate * len(gates))(*[ZkPlonkGate(*g) for g in gates])
        kis = np.array([int(k) for k in k_is], dtype=np.uint64)
        dig = np.array([int(x) for x in circuit_digest], dtype=np.uint64)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.zk_plonk_circuit_create(self.ctx.handle, C.byref(common), C.cast(garr, C.c_void_p), len(gates),
                                                            C.c_void_p(constants_sigmas.data_ptr()), stride, kis.ctypes.data,
                                                            dig.ctypes.data, C.byref(h)))
        self.handle = h

    def constants_sigmas_cap(self) -> np.ndarray:
        out = np.zeros((1 << self.config.cap_height, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.zk_plonk_circuit_cap(self.handle, out.ctypes.data))
        return out

    def prove(self, wires, public_inputs: Sequence[int]) -> ProofWithPublicInputs:
        """`CircuitData::prove` from the full witness on.  wires: CUDA tensor (num_wires, 2^degree_bits), the
        `MatrixWitness::wire_values`."""
        from .stark import _trace_args
        n_cols, n, log_n, stride = _trace_args(wires)
        if n_cols != self.config.num_wires or log_n != self.degree_bits:
            raise ZkStarkError(-1, "witness must be (num_wires, 2^degree_bits)")
        self.ctx.use_torch_current_stream()
        pis = np.array([int(x) for x in public_inputs], dtype=np.uint64)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.zk_plonk_prove(self.handle, C.c_void_p(wires.data_ptr()), stride,
                                                   pis.ctypes.data if pis.size else None, pis.size, C.byref(h)))
        try:
            return self._proof_from_handle(h, pis)
        finally:
            self.ctx.lib.zk_plonk_proof_free(h)

    def _proof_from_handle(self, h, pis) -> ProofWithPublicInputs:
        lib = self.ctx.lib
        v = ZkPlonkProofView()
        self.ctx.check(lib.zk_plonk_proof_get(h, C.byref(v)))

        def arr(p, words):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(words,)).copy()
        nd = int(v.cap_digests)
        names = ("wires commitment", "partial products and Zs", "quotient", "openings", "FRI")
        return ProofWithPublicInputs(arr(v.wires_cap, 4 * nd).reshape(nd, 4), arr(v.plonk_zs_partial_products_cap, 4 * nd).reshape(nd, 4),
                                     arr(v.quotient_polys_cap, 4 * nd).reshape(nd, 4), arr(v.openings, 2 * v.n_openings).reshape(-1, 2),
                                     arr(v.opening_proof, v.proof_words), [int(x) for x in pis],
                                     [int(x) for x in v.public_inputs_hash], dict(zip(names, [float(x) for x in v.stage_ms])))

    def prove_batch(self, wires_list, public_inputs_list, in_flight: int = 0) -> list:
        """K proofs of this circuit from one call (zk_plonk_prove_batch): the library runs them through `in_flight` worker
        contexts (0 = its default of 4) so that the small proofs of the recursion layer fill the GPU; every proof equals
        `prove` of the same witness.  wires_list[k]: CUDA tensor (num_wires, 2^degree_bits)."""
        from .stark import _trace_args
        K = len(wires_list)
        if K == 0:
            return []
        if len(public_inputs_list) != K:
            raise ZkStarkError(-1, "one public-input list per witness")
        strides = set()
        for w in wires_list:
            n_cols, n, log_n, stride = _trace_args(w)
            if n_cols != self.config.num_wires or log_n != self.degree_bits:
                raise ZkStarkError(-1, "witness must be (num_wires, 2^degree_bits)")
            strides.add(stride)
        if len(strides) != 1:
            raise ZkStarkError(-1, "the witnesses of one batch must share a column stride")
        self.ctx.use_torch_current_stream()
        pis = [np.array([int(x) for x in p], dtype=np.uint64) for p in public_inputs_list]
        n_pi = pis[0].size
        if any(p.size != n_pi for p in pis):
            raise ZkStarkError(-1, "every proof of a circuit has the same number of public inputs")
        wptr = (C.c_void_p * K)(*[w.data_ptr() for w in wires_list])
        pptr = (C.c_void_p * K)(*[(p.ctypes.data if n_pi else None) for p in pis])
        outs = (C.c_void_p * K)()
        self.ctx.check(self.ctx.lib.zk_plonk_prove_batch(self.handle, wptr, strides.pop(), pptr if n_pi else None, n_pi, K,
                                                         in_flight, outs))
        try:
            return [self._proof_from_handle(C.c_void_p(outs[k]), pis[k]) for k in range(K)]
        finally:
            for k in range(K):
                if outs[k]:
                    self.ctx.lib.zk_plonk_proof_free(C.c_void_p(outs[k]))

    def free(self):
        if getattr(self, "handle", None):
            self.ctx.lib.zk_plonk_circuit_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
