"""not gpu: the product's HOST-side transcript code (csrc/host_hash.hpp behind zk_challenger_*: the Poseidon permutation in
its blocked schedule, the Keccak hash onion, the duplex buffering) against the oracle's challenger, and against the
reference-held Poseidon known answers.  These entry points touch no device, so the comparison runs on the CPU box too; the
same check under -m gpu: tests/test_gpu_fri.py::test_challenger_matches_oracle."""
import numpy as np
import pytest


@pytest.mark.parametrize("hasher", [0, 1])
def test_host_challenger_matches_oracle_without_a_gpu(oracle, hasher):
    from tests.test_gpu_fri import _host_challenger_matches_oracle
    _host_challenger_matches_oracle(oracle, hasher)


def test_host_permutation_reproduces_hash_zeros():
    """HASH_ZEROS = first four lanes of Poseidon([0; 12]) (reference smt_trie/src/keys.rs:10-15): observing eight zeros
    makes the challenger permute the zero state once; compact() returns it."""
    from zk_evm_amd import Challenger
    ch = Challenger(0)
    ch.observe_elements(np.zeros(8, dtype=np.uint64))
    st = ch.compact()
    assert [int(x) for x in st[:4]] == [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202]
