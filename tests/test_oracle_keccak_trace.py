"""CPU: the oracle's restatement of the Keccak table generator (oracle/keccak_trace.py) -- pinned like the
reference pins its own generator (keccak_stark.rs:657-690: last round's A''' == keccak-f(input)), and consistent
with the restated AIR (every constraint of eval_keccak vanishes on the generated rows, first/last/transition)."""
import numpy as np

from oracle import airs as oairs
from oracle import keccak_trace as kt
from oracle import stark as orc

P = 0xFFFFFFFF00000001


def _outputs(rows, base):
    r = rows[base + 23]
    out = [0] * 25
    for x in range(5):
        for y in range(5):
            c = oairs.k_reg_a_ppp(x, y)
            out[y * 5 + x] = int(r[c]) | (int(r[c + 1]) << 32)
    return out


def test_generated_rows_match_keccak_f_and_satisfy_the_air(oracle):
    rng = np.random.default_rng(42)
    inputs = [([int(v) for v in rng.integers(0, 1 << 64, size=25, dtype=np.uint64)], 100 + 7 * i) for i in range(2)]
    rows = kt.generate_trace_rows(inputs, 8)
    assert rows.shape == (64, 2431)
    for p, (inp, _) in enumerate(inputs):
        a = np.array(inp, dtype=np.uint64)
        oracle.lib.orc_keccak_f1600(a)
        assert _outputs(rows, 24 * p) == [int(v) for v in a]
    assert not rows[48:].any()
    n = rows.shape[0]

    class Cons:                      # raw constraint values with the row-position multipliers of the real consumer
        def __init__(self, i): self.i, self.bad = i, []
        def constraint(self, c): self.bad += [c % P] if c % P else []
        def constraint_transition(self, c): self.bad += [c % P] if (self.i != n - 1 and c % P) else []
        def constraint_first_row(self, c): self.bad += [c % P] if (self.i == 0 and c % P) else []
        def constraint_last_row(self, c): self.bad += [c % P] if (self.i == n - 1 and c % P) else []
    for i in range(n):
        lv = [int(v) for v in rows[i]]
        nv = [int(v) for v in rows[(i + 1) % n]]
        c = Cons(i)
        oairs.eval_keccak(lv, nv, c)
        assert not c.bad, (i, len(c.bad))
