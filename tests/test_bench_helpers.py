"""CPU: the pieces of the bench / multi-GPU tooling that need no GPU -- the block-shaped job list of BASELINE configs[3] and
the z-data order the row-sharded prover hands to the library."""
import numpy as np

from tools.benchlib import B19807080_RANGES, block_segment_shapes


def test_block_segment_shapes_follow_the_reference_ranges():
    shapes = block_segment_shapes(40, seed=3)
    assert shapes == block_segment_shapes(40, seed=3) and shapes != block_segment_shapes(40, seed=4)
    some_absent = set()
    for i, (log_ns, in_use) in enumerate(shapes):
        assert len(log_ns) == 9 and len(in_use) == 9
        for t, (lo, hi) in enumerate(B19807080_RANGES):
            assert lo <= log_ns[t] < hi                                   # scripts/prove_stdio.rs:89-101 (Rust ranges: hi exclusive)
            if not in_use[t]:
                assert log_ns[t] == lo and t in (1, 3, 4, 5, 8)           # only OPTIONAL_TABLE_INDICES may be absent
                some_absent.add(t)
        assert in_use[3] == in_use[4]                                     # Keccak and KeccakSponge go together (generation/mod.rs:588-593)
        assert in_use[8] == (i != len(shapes) - 1)                        # MemAfter empties in the block's last segment
    assert {1, 3, 4, 5, 8} <= some_absent
    assert all(max(l) <= 11 for l, _ in block_segment_shapes(10, max_log=11))


def test_table_ctl_specs_order_is_starkys():
    """per CTL, per challenge: the run of the table's looking entries, then the looked entry (cross_table_lookup_data)"""
    from zk_evm_amd.all_stark import AllStark, Table
    from zk_evm_amd.shard_prover import table_ctl_specs
    st = AllStark((1, 2, 3, 4))
    chal = [(11, 12), (21, 22)]
    spec = table_ctl_specs(st, Table.MemBefore, chal)
    # MemBefore: one looking entry in the memory CTL, then the looked side of mem_before -- each under both challenges
    assert [(b, g, len(e)) for b, g, e in spec] == [(11, 12, 1), (21, 22, 1), (11, 12, 1), (21, 22, 1)]
    spec = table_ctl_specs(st, Table.Keccak, chal)
    assert [(b, g, len(e)) for b, g, e in spec] == [(11, 12, 1), (21, 22, 1)] * 2          # looked by keccak_inputs and keccak_outputs
    ks = table_ctl_specs(st, Table.KeccakSponge, chal)
    assert max(len(e) for _, _, e in ks) == 136                           # its run of 136 memory reads is ONE z-data with helper columns
    n_z = {t: len(table_ctl_specs(st, t, chal)) for t in range(9)}
    from zk_evm_amd.segment import num_ctl_helpers_zs_all
    for t in range(9):
        assert n_z[t] == num_ctl_helpers_zs_all(st.cross_table_lookups, t, 2, 3)[1], t
