"""CPU: the product's table/CTL definitions (zk_evm_amd/all_stark.py, mirrors the reference's function structure)
against the oracle's independent transcription by column number (oracle/all_stark.py), plus the counts
SURVEY.md 8(a)/8(a') derives from the reference (tuple widths, looker counts, aux columns per table)."""
import numpy as np
import pytest

from oracle import all_stark as oas
from zk_evm_amd import all_stark as pas
from zk_evm_amd.stark import encode_program


def _enc(cols, filt):
    w = []
    for c in cols:
        w += [len(c.linear_combination), len(c.next_row_linear_combination), c.constant]
        w += [x for pr in c.linear_combination for x in pr] + [x for pr in c.next_row_linear_combination for x in pr]
    w += [len(filt.products), len(filt.constants)]
    for a, b in filt.products:
        for c in (a, b):
            w += [len(c.linear_combination), len(c.next_row_linear_combination), c.constant]
            w += [x for pr in c.linear_combination for x in pr] + [x for pr in c.next_row_linear_combination for x in pr]
    for c in filt.constants:
        w += [len(c.linear_combination), len(c.next_row_linear_combination), c.constant]
        w += [x for pr in c.linear_combination for x in pr] + [x for pr in c.next_row_linear_combination for x in pr]
    return w


@pytest.mark.parametrize("cdk_erigon", [False, True])
def test_ctl_definitions_agree(cdk_erigon):
    prod, orc = pas.all_cross_table_lookups(cdk_erigon), oas.build_ctls(cdk_erigon)
    assert len(prod) == len(orc) == (13 if cdk_erigon else pas.NUM_CTLS)       # all_stark.rs:148
    assert pas._C.num_columns == 85                                               # the column map swap is scoped
    for i, (a, b) in enumerate(zip(prod, orc)):
        assert len(a.looking_tables) == len(b.looking_tables), i
        for j, (x, y) in enumerate(zip(a.looking_tables + [a.looked_table], b.looking_tables + [b.looked_table])):
            assert x.table == y.table, (i, j)
            assert _enc(x.columns, x.filter) == _enc(y.columns, y.filter), (i, j)


def test_lookup_definitions_agree():
    orc = oas.build_lookups()
    for t in pas.Table.all():
        a, b = pas.table_lookups(t), orc[t]
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert _enc(x.columns + [x.table_column, x.frequencies_column], x.filter_columns[0]) == \
                _enc(y.columns + [y.table_column, y.frequencies_column], y.filter_columns[0])
            for f, g in zip(x.filter_columns, y.filter_columns):
                assert _enc([], f) == _enc([], g)


def test_counts_match_survey():
    ctls = pas.all_cross_table_lookups()
    widths = [len(c.looked_table.columns) for c in ctls]
    # arithmetic, byte_packing, keccak_sponge, keccak_inputs, keccak_outputs, logic, memory, mem_before, mem_after, pruning
    assert widths == [33, 14, 13, 51, 51, 25, 13, 11, 11, 1]
    for c in ctls:
        assert all(len(l.columns) == len(c.looked_table.columns) for l in c.looking_tables)
    assert len(ctls[pas.MEMORY_CTL_IDX].looking_tables) == 176
    from zk_evm_amd.segment import num_ctl_helpers_zs_all
    aux = [sum(num_ctl_helpers_zs_all(ctls, t, 2, 3)[:2]) for t in pas.Table.all()]
    assert aux == [2, 36, 24, 4, 152, 2, 8, 4, 2]
    lookup_aux = [sum(l.num_helper_columns(3) for l in pas.table_lookups(t)) * 2 for t in pas.Table.all()]
    assert lookup_aux == [98, 34, 0, 0, 138, 0, 8, 0, 0]
    assert pas.TABLE_COLUMNS == list(oas.TABLE_COLUMNS) and pas.TABLE_AIR == list(oas.TABLE_AIR)
    # every referenced column exists in its table
    for c in ctls:
        for t in c.looking_tables + [c.looked_table]:
            prog = encode_program([(t.columns, t.filter)])
            assert prog.dtype == np.uint64
            for col in t.columns:
                for idx, _ in col.linear_combination + col.next_row_linear_combination:
                    assert 0 <= idx < pas.TABLE_COLUMNS[t.table]


def test_cdk_erigon_counts():
    """all_stark.rs:103-172,344-366,419-441: ten tables, 13 CTLs, 176 + 56 Memory lookers; tuple widths of the
    three Poseidon CTLs (12 inputs + 8 digest limbs; 5; 8 + timestamp); every referenced column inside its table."""
    st = pas.AllStark((1, 2, 3, 4), cdk_erigon=True)
    reg = oas.Registry(True)
    ctls = st.cross_table_lookups
    assert st.num_tables == reg.NUM_TABLES == 10 and len(ctls) == 13
    assert st.table_columns == list(reg.TABLE_COLUMNS) and st.table_air == list(reg.TABLE_AIR)
    assert st.optional_table_indices == list(reg.OPTIONAL_TABLES)
    assert len(ctls[pas.MEMORY_CTL_IDX].looking_tables) == 176 + 56
    assert [len(c.looked_table.columns) for c in ctls[10:]] == [20, 5, 9]
    assert [c.looked_table.table for c in ctls[10:]] == [9, 9, 9] and all(c.looking_tables[0].table == pas.Table.Cpu for c in ctls[10:])
    for c in ctls:
        for t in c.looking_tables + [c.looked_table]:
            assert len(t.columns) == len(c.looked_table.columns)
            for col in t.columns + [x for pr in t.filter.products for x in pr] + t.filter.constants:
                for idx, _ in col.linear_combination + col.next_row_linear_combination:
                    assert 0 <= idx < st.table_columns[t.table]
    # the code-read filter sums all 19 operation flags, the new one included
    assert sorted(i for i, _ in ctls[6].looking_tables[0].filter.constants[0].linear_combination) == list(range(6, 25))
    from zk_evm_amd.segment import num_ctl_helpers_zs_all
    aux = [sum(num_ctl_helpers_zs_all(ctls, t, 2, 3)[:2]) for t in range(10)]
    assert aux == [2, 36, 24 + 6, 4, 152, 2, 8, 4, 2, 28 * 2 + 2 + 6]


def test_cdk_erigon_public_values():
    """get_challenges.rs:66-74,146-154,211-219: no eth_mainnet block-metadata fields, the burn address appended."""
    import zk_evm_amd.segment as sg
    from oracle import segment as oseg
    from tests.consistent_segment import make_public_values
    from tests.test_gpu_segment import to_public_values
    d = make_public_values(np.random.default_rng(3))
    d.update(burn_addr=(1 << 160) - 12345, blob_gas_used=0, excess_blob_gas=0, parent_beacon_root=bytes(32))
    pv = to_public_values(d)
    pv.burn_addr = d["burn_addr"]
    e = sg.public_values_elements(pv)
    assert e == oseg.pv_elements(d) and len(e) == 2217 - 12 + 8
    pv.block_metadata.block_blob_gas_used = 1
    with pytest.raises(sg.ZkStarkError):
        sg.public_values_elements(pv)


def test_public_values_range_errors():
    import zk_evm_amd.segment as sg
    pv = sg.PublicValues()
    pv.block_metadata.block_timestamp = 1 << 32
    with pytest.raises(sg.PublicValuesError):
        sg.public_values_elements(pv)
    pv = sg.PublicValues()
    pv.block_metadata.block_base_fee = 1 << 64
    with pytest.raises(sg.PublicValuesError):
        sg.public_values_elements(pv)
    assert len(sg.public_values_elements(sg.PublicValues())) == 48 + 5 + 3 + 8 + 2 + 2 + 1 + 4 + 8 + 64 + 257 * 8 + 8 + 4 + 4


def test_generated_registry_header_is_current():
    """include/zk_all_stark.h (the registry in zk_prove_segment's flat encodings, for C / C++ / Rust callers) is what
    tools/gen_all_stark_header.py renders from the definitions tested above; `check_num_ctls` (all_stark.rs:451-454)
    holds for the header's counts."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_all_stark_header as g
    text = open(os.path.join(root, "include", "zk_all_stark.h")).read()
    assert text == g.render(), "stale: run python tools/gen_all_stark_header.py"
    assert "#define ZK_ALLSTARK_NUM_CTLS 10\n" in text and "#define ZK_ALLSTARK_ERIGON_NUM_CTLS 13\n" in text
    assert "#define ZK_ALLSTARK_NUM_TABLES 9\n" in text and "#define ZK_ALLSTARK_ERIGON_NUM_TABLES 10\n" in text
