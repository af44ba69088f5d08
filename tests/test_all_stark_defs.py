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


def test_ctl_definitions_agree():
    prod, orc = pas.all_cross_table_lookups(), oas.build_ctls()
    assert len(prod) == len(orc) == pas.NUM_CTLS == 10
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
