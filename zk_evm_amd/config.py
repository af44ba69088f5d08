"""Host mirrors of plonky2 ``FriConfig`` / starky ``StarkConfig`` (the knobs the hot path reads).

Reference: ``StarkConfig::standard_fast_config()`` is what production uses
(zero/src/prover_state/mod.rs:283, evm_arithmetization/src/fixed_recursive_verifier.rs:637);
``TEST_STARK_CONFIG`` is evm_arithmetization/src/testing_utils.rs:41-51.
"""
import ctypes as C
from dataclasses import dataclass, field

HASH_POSEIDON = 0
HASH_KECCAK25 = 1


class ZkCfg(C.Structure):
    """C layout of ``zk_cfg`` (include/zkstark.h)."""
    _fields_ = [
        ("rate_bits", C.c_uint32), ("cap_height", C.c_uint32), ("hasher", C.c_uint32),
        ("num_challenges", C.c_uint32), ("proof_of_work_bits", C.c_uint32),
        ("num_query_rounds", C.c_uint32), ("arity_bits", C.c_uint32),
        ("final_poly_bits", C.c_uint32),
    ]


@dataclass(frozen=True)
class FriConfig:
    rate_bits: int = 1
    cap_height: int = 4
    proof_of_work_bits: int = 16
    # FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
    arity_bits: int = 4
    final_poly_bits: int = 5
    num_query_rounds: int = 84


@dataclass(frozen=True)
class StarkConfig:
    security_bits: int = 100
    num_challenges: int = 2
    fri_config: FriConfig = field(default_factory=FriConfig)
    hasher: int = HASH_POSEIDON  # GenericConfig::Hasher: Poseidon (production) or Keccak-25

    @staticmethod
    def standard_fast_config() -> "StarkConfig":
        return StarkConfig()

    @staticmethod
    def test_config() -> "StarkConfig":
        # TEST_STARK_CONFIG (testing_utils.rs:41-51): security_bits 1, num_challenges 1,
        # rate_bits 1, cap_height 4, proof_of_work_bits 1, ConstantArityBits(4, 5), 1 query round
        return StarkConfig(security_bits=1, num_challenges=1,
                           fri_config=FriConfig(proof_of_work_bits=1, num_query_rounds=1))

    def to_c(self, *, rate_bits=None, cap_height=None) -> ZkCfg:
        f = self.fri_config
        return ZkCfg(
            rate_bits=f.rate_bits if rate_bits is None else rate_bits,
            cap_height=f.cap_height if cap_height is None else cap_height,
            hasher=self.hasher, num_challenges=self.num_challenges,
            proof_of_work_bits=f.proof_of_work_bits, num_query_rounds=f.num_query_rounds,
            arity_bits=f.arity_bits, final_poly_bits=f.final_poly_bits)
