"""CPU: the ARITHMETIC of the row-sharded (level-3) auxiliary columns, stated on the oracle's builders -- a table cut into W row
blocks, every block's logUp helper / Z columns and CTL partial sums built from the block alone, then stitched exactly as
zk_evm_amd/shard_prover.py stitches the device builders' output:
  * a lookup's Z is a forward running sum: block q adds the totals of the blocks before it;
  * a CTL's Z is a reverse running sum: block q adds the totals of the blocks after it;
  * columns that read the NEXT row: the block's last row is redone on a "seam" trace (row 0 = the block's last row, then the next
    block's first row) -- except in the last block, where starky takes the next row to be zero.
The result must be the whole-trace builders' columns.  (The device path itself is compared proof-for-proof with the single-GPU
prover in tests/test_gpu_multirank.py; this file pins the algebra on CPU, where the driver runs every round.)"""
import random

import pytest

from oracle.stark import P, Column, Filter, GrandProductChallenge, Lookup, lookup_helper_columns, partial_sums


def _trace(n_cols, n, rng):
    tr = [[rng.randrange(P) for _ in range(n)] for _ in range(n_cols)]
    tr[0] = [rng.randrange(2) for _ in range(n)]          # binary filter columns
    tr[1] = [rng.randrange(2) for _ in range(n)]
    return tr


def _block(tr, q, nb):
    return [c[q * nb:(q + 1) * nb] for c in tr]


def _seam(tr, q, nb, world, rows=4):
    last = [c[(q + 1) * nb - 1] for c in tr]
    nxt = [c[((q + 1) % world) * nb] for c in tr]
    return [[last[k]] + [nxt[k]] * (rows - 1) for k in range(len(tr))]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_lookup_columns_from_row_blocks(world):
    rng = random.Random(100 + world)
    n, nb = 64, 64 // world
    tr = _trace(7, n, rng)
    lk = Lookup([Column.single(2), Column.single_next_row(3), Column.linear_combination_and_next_row_with_constant([(4, 3)], [(2, 5)], 9)],
                Column.single(5), Column.single(6), [Filter.new_simple(Column.single(0)), Filter(), Filter.new_simple(Column.single(1))])
    alpha = rng.randrange(P)
    want = lookup_helper_columns(lk, tr, alpha, 3)
    blocks, totals = [], []
    for q in range(world):
        cols = [list(c) for c in lookup_helper_columns(lk, _block(tr, q, nb), alpha, 3)]
        mini = lookup_helper_columns(lk, _seam(tr, q, nb, world), alpha, 3)
        if q + 1 < world:
            for h in range(len(cols) - 1):
                cols[h][nb - 1] = mini[h][0]              # the last row's helpers with the true next row
        totals.append((cols[-1][nb - 1] + mini[-1][1]) % P)   # Z[last] + the last row's increment (the seam's Z[1])
        blocks.append(cols)
    for q in range(world):
        carry = sum(totals[:q]) % P
        blocks[q][-1] = [(z + carry) % P for z in blocks[q][-1]]
    got = [sum((blocks[q][k] for q in range(world)), []) for k in range(len(want))]
    assert got == want


@pytest.mark.parametrize("world", [2, 4])
def test_ctl_partial_sums_from_row_blocks(world):
    rng = random.Random(200 + world)
    n, nb = 32, 32 // world
    tr = _trace(6, n, rng)
    entries = [([Column.single(2), Column.single_next_row(3)], Filter.new_simple(Column.single(0))),
               ([Column.single(4), Column.single(5)], Filter.new_simple(Column.single(1))),
               ([Column.single_next_row(2), Column.single(3)], Filter())]
    ch = GrandProductChallenge(rng.randrange(P), rng.randrange(P))
    want = partial_sums(tr, entries, ch, 3)
    blocks = []
    for q in range(world):
        cols = [list(c) for c in partial_sums(_block(tr, q, nb), entries, ch, 3)]
        if q + 1 < world:
            mini = partial_sums(_seam(tr, q, nb, world), entries, ch, 3)
            for h in range(len(cols) - 1):
                cols[h][nb - 1] = mini[h][0]
            delta = (mini[-1][0] - mini[-1][1] - cols[-1][nb - 1]) % P     # the true last-row term minus the wrapped one
            cols[-1] = [(z + delta) % P for z in cols[-1]]
        blocks.append(cols)
    totals = [b[-1][0] for b in blocks]
    for q in range(world):
        carry = sum(totals[q + 1:]) % P
        blocks[q][-1] = [(z + carry) % P for z in blocks[q][-1]]
    got = [sum((blocks[q][k] for q in range(world)), []) for k in range(len(want))]
    assert got == want
