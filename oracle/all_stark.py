"""ORACLE (test infrastructure only -- never imported by the product).

Second, independent transcription of the reference's table registry and cross-table-lookup wiring
(evm_arithmetization/src/all_stark.rs:153-417 and each table's `ctl_*` / `lookups()`), written directly
against the column *numbers* of each table's `#[repr(C)]` column struct instead of mirroring the reference's
function structure.  tests/test_all_stark_defs.py requires the product's definitions
(zk_evm_amd/all_stark.py) to encode to exactly the same programs; counts are pinned against SURVEY.md 8(a):
tuple widths 33/14/13/51/51/25/13/11/11/1, 176 Memory lookers, CTL aux columns per table
2/36/24/4/152/2/8/4/2 (two challenges).
"""
from .stark import Column, Filter, Lookup

ARITHMETIC, BYTE_PACKING, CPU, KECCAK, KECCAK_SPONGE, LOGIC, MEMORY, MEM_BEFORE, MEM_AFTER = range(9)
NUM_TABLES = 9
OPTIONAL_TABLES = (BYTE_PACKING, KECCAK, KECCAK_SPONGE, LOGIC, MEM_AFTER)      # all_stark.rs:124-131
TABLE_COLUMNS = (116, 71, 85, 2431, 438, 523, 30, 12, 12)
TABLE_AIR = (5, 4, 8, 6, 7, 2, 3, 1, 1)     # oracle/airs.py AIRS ids


def S(i): return Column([(i, 1)])
def N(i): return Column([], [(i, 1)])
def K(c): return Column([], [], c % 0xFFFFFFFF00000001)
def LC(pairs, const=0, nxt=()):
    p = 0xFFFFFFFF00000001
    return Column([(c, f % p) for c, f in pairs], [(c, f % p) for c, f in nxt], const % p)
def simple(col): return Filter([], [col])
def prod(a, b): return Filter([(a, b)], [])


class TWC:
    def __init__(self, table, columns, filt):
        self.table, self.columns, self.filter = table, columns, filt


class CTL:
    def __init__(self, looking, looked):
        self.looking_tables, self.looked_table = looking, looked


# --- Cpu column numbers (cpu/columns/mod.rs:56-97) ---
CTX, CODE_CTX, PC, STACK_LEN, KERNEL, GAS = range(6)
(BINARY_OP, TERNARY_OP, FP254_OP, EQ_ISZERO, LOGIC_OP, NOT_POP, SHIFT, JUMPDEST_KECCAK_GENERAL, JUMPS,
 PUSH_PROVER_INPUT, DUP_SWAP, CONTEXT_OP, M_OP_32BYTES, EXIT_KERNEL, M_OP_GENERAL, PC_PUSH0, SYSCALL,
 EXCEPTION) = range(6, 24)
BIT = list(range(24, 32))
GENERAL0 = 32
CLOCK = 40
def CH(k, f): return 41 + 13 * k + f          # f: 0 used, 1 is_read, 2 ctx, 3 seg, 4 virt, 5.. value
def CHV(k): return [41 + 13 * k + 5 + i for i in range(8)]
PARTIAL = 80
NCH = 5                                       # NUM_CHANNELS (cpu/membus.rs:32)


def _ts(channel=None):
    # (clock - 1) * NCH + 1 (+ channel): cpu_stark.rs:48-52,307-311
    return LC([(CLOCK, NCH)], 1 - NCH + (channel or 0))


def _opcode(): return LC([(BIT[i], 1 << i) for i in range(8)])


def cpu_keccak_sponge_cols():
    v0 = CHV(0)
    return [S(v0[2]), S(v0[1]), S(v0[0]), S(CHV(1)[0]), _ts()] + [N(c) for c in v0]


def _erigon_cpu(col):
    """Cpu column number of the eth_mainnet layout -> `cdk_erigon` layout (`poseidon` flag inserted at 14)."""
    return col + 1 if col >= 14 else col


def _remap_column(c, f):
    return Column([(f(i), k) for i, k in c.linear_combination], [(f(i), k) for i, k in c.next_row_linear_combination], c.constant)


def _remap_twc(t, f):
    filt = Filter([(_remap_column(a, f), _remap_column(b, f)) for a, b in t.filter.products],
                  [_remap_column(c, f) for c in t.filter.constants])
    return TWC(t.table, [_remap_column(c, f) for c in t.columns], filt)


def build_ctls(cdk_erigon=False):
    """cdk_erigon=True (all_stark.rs:103-172,344-366,419-441): every Cpu entry re-numbered for the 86-column table,
    the code-read filter extended by the new flag (it sums ALL operation flags), 56 Poseidon-table byte reads appended
    to the Memory CTL, and CTLs 10-12 (Poseidon simple / general input / general output)."""
    if cdk_erigon:
        from . import poseidon_table as PT
        ctls = build_ctls(False)
        for ctl in ctls:
            ctl.looking_tables = [_remap_twc(t, _erigon_cpu) if t.table == CPU else t for t in ctl.looking_tables]
            if ctl.looked_table.table == CPU:
                ctl.looked_table = _remap_twc(ctl.looked_table, _erigon_cpu)
        code_read = ctls[6].looking_tables[0]
        code_read.filter = simple(LC([(c, 1) for c in range(6, 25)]))              # COL_MAP.op.iter(): 19 flags
        ctls[6].looking_tables += [PT.ctl_looking_memory(i) for i in range(56)]
        e = lambda k: 42 + 13 * k + 5                                               # mem_channels[k].value[0], 86-col layout
        pos, bit0, clock = 14, 25, 41
        simple_cols = [LC([(e(k) + 2 * i, 1), (e(k) + 2 * i + 1, 1 << 32)]) for k in range(3) for i in range(4)] + \
                      [N(e(0) + i) for i in range(8)]
        f_simple, f_general = prod(S(pos), LC([(bit0, -1)], 1)), prod(S(pos), S(bit0))
        ctls.append(CTL([TWC(CPU, simple_cols, f_simple)], PT.ctl_looked_simple_op()))
        ctls.append(CTL([TWC(CPU, [S(e(0) + 2), S(e(0) + 1), S(e(0)), S(e(1)), LC([(clock, NCH)])], f_general)],
                        PT.ctl_looked_general_input()))
        ctls.append(CTL([TWC(CPU, [N(e(0) + i) for i in range(8)] + [LC([(clock, NCH)])], prod(S(pos), S(bit0)))],
                        PT.ctl_looked_general_output()))
        return ctls
    v0, v1, v2 = CHV(0), CHV(1), CHV(2)
    one_minus = lambda c: LC([(c, -1)], 1)
    ctls = []
    # 0 arithmetic (all_stark.rs:176-181)
    cpu_arith = TWC(CPU, [_opcode()] + [S(c) for c in v0 + v1 + v2] + [N(c) for c in v0],
                    Filter([(S(PUSH_PROVER_INPUT), S(BIT[7]))],
                           [LC([(c, 1) for c in (BINARY_OP, FP254_OP, TERNARY_OP, SHIFT, SYSCALL, EXCEPTION)])]))
    codes = [0x01, 0x02, 0x03, 0x04, 0x06, 0x08, 0x09, 0x0c, 0x0d, 0x0e, 0x0f, 0x10, 0x11, 0x1a, 0x1b, 0x1c]
    arith_cols = [LC([(i, codes[i]) for i in range(16)] + [(17, 1)])]
    for reg in (18, 34, 50, 66):
        arith_cols += [LC([(reg + 2 * i, 1), (reg + 2 * i + 1, 65536)]) for i in range(8)]
    ctls.append(CTL([cpu_arith], TWC(ARITHMETIC, arith_cols, simple(LC([(i, 1) for i in range(17)])))))
    # 1 byte_packing (all_stark.rs:185-220)
    pack = TWC(CPU, [K(1)] + cpu_keccak_sponge_cols(), prod(S(M_OP_32BYTES), S(BIT[5])))
    unpack = TWC(CPU, [K(0), S(v0[2]), S(v0[1]), S(v0[0]), LC([(v0[0], -1)], 0, [(v0[0], 1)]), _ts()] +
                 [S(c) for c in v1], prod(S(M_OP_32BYTES), one_minus(BIT[5])))
    push = TWC(CPU, [K(1), S(CODE_CTX), K(0), LC([(PC, 1)], 1), LC([(BIT[i], 1 << i) for i in range(5)], 1), _ts()] +
               [N(c) for c in v0], prod(S(GENERAL0), S(PUSH_PROVER_INPUT)))
    jumptable = TWC(CPU, [K(1), S(CH(1, 2)), S(CH(1, 3)), S(CH(1, 4)), K(3), _ts()] + [S(c) for c in v1],
                    simple(LC([(SYSCALL, 1), (EXCEPTION, 1)])))
    bp_out = [LC([(37 + 4 * i + j, 1 << (8 * j)) for j in range(4)]) for i in range(8)]
    bp_looked = TWC(BYTE_PACKING, [S(0), S(33), S(34), S(35), LC([(1 + i, i + 1) for i in range(32)]), S(36)] + bp_out,
                    simple(LC([(1 + i, 1) for i in range(32)])))
    ctls.append(CTL([pack, unpack, push, jumptable], bp_looked))
    # 2 keccak_sponge (all_stark.rs:259-271)
    ks_out = [LC([(404 + 4 * i + j, 1 << (24 - 8 * j)) for j in range(4)]) for i in (7, 6, 5, 4, 3, 2, 1, 0)]
    ks_len = LC([(5, 1)] + [(6 + i, -1) for i in range(136)], 136)
    ctls.append(CTL([TWC(CPU, cpu_keccak_sponge_cols(), prod(S(JUMPDEST_KECCAK_GENERAL), one_minus(BIT[1])))],
                    TWC(KECCAK_SPONGE, [S(1), S(2), S(3), ks_len, S(4)] + ks_out, simple(S(6 + 135)))))
    # 3, 4 keccak inputs / outputs (all_stark.rs:226-255)
    ks_active = lambda: simple(LC([(0, 1), (6 + 135, 1)]))
    def a(x, y): return 25 + (x * 5 + y) * 2
    def appp(x, y): return 2429 if x == 0 and y == 0 else 2315 + x * 10 + y * 2
    k_in = [S(a((i // 2) % 5, (i // 2) // 5) + i % 2) for i in range(50)] + [S(24)]
    k_out = [S(appp((i // 2) % 5, (i // 2) // 5) + i % 2) for i in range(50)] + [S(24)]
    ctls.append(CTL([TWC(KECCAK_SPONGE, [S(328 + i) for i in range(34)] + [S(176 + i) for i in range(16)] + [S(4)],
                         ks_active())], TWC(KECCAK, k_in, simple(S(0)))))
    digest = [LC([(404 + 4 * k + i, 1 << (8 * i)) for i in range(4)]) for k in range(8)]
    ctls.append(CTL([TWC(KECCAK_SPONGE, digest + [S(362 + i) for i in range(42)] + [S(4)], ks_active())],
                    TWC(KECCAK, k_out, simple(S(23)))))
    # 5 logic (all_stark.rs:275-292)
    lookers = [TWC(CPU, [_opcode()] + [S(c) for c in v0 + v1] + [N(c) for c in v0], simple(S(LOGIC_OP)))]
    for i in range(5):
        def pad8(cols): return (cols + [Column()] * 8)[:8]
        rate = pad8([S(142 + j) for j in range(8 * i, 34)])
        blk = pad8([LC([(192 + k + b, 1 << (8 * b)) for b in range(min(4, 136 - k))]) for k in range(32 * i, 136, 4)])
        xored = pad8([S(328 + j) for j in range(8 * i, 34)])
        lookers.append(TWC(KECCAK_SPONGE, [K(0x18)] + rate + blk + xored, ks_active()))
    lg = [LC([(0, 0x16), (1, 0x17), (2, 0x18)])]
    lg += [LC([(3 + 32 * k + b, 1 << b) for b in range(32)]) for k in range(8)]
    lg += [LC([(259 + 32 * k + b, 1 << b) for b in range(32)]) for k in range(8)]
    lg += [S(515 + k) for k in range(8)]
    ctls.append(CTL(lookers, TWC(LOGIC, lg, simple(LC([(0, 1), (1, 1), (2, 1)])))))
    # 6 memory (all_stark.rs:296-375): tuple = is_read, ctx, seg, virt, value[8], timestamp
    zeros7 = lambda: [K(0) for _ in range(7)]
    set_ctx = lambda: prod(S(CONTEXT_OP), S(BIT[0]))
    lookers = [
        TWC(CPU, [K(1), S(CODE_CTX), K(0), S(PC), _opcode()] + zeros7() + [_ts(0)],
            simple(LC([(c, 1) for c in range(BINARY_OP, EXCEPTION + 1)]))),
        TWC(CPU, [S(PARTIAL + 1), S(PARTIAL + 2), S(PARTIAL + 3), S(PARTIAL + 4)] + [S(c) for c in v0] + [_ts(4)],
            simple(S(PARTIAL))),
        TWC(CPU, [K(0), S(CTX), K(6), K(11), LC([(STACK_LEN, 1)], -1)] + zeros7() + [_ts(2)], set_ctx()),
        TWC(CPU, [K(1), S(v0[2]), K(6), K(11), N(STACK_LEN)] + zeros7() + [_ts(3)], set_ctx()),
    ]
    for k in range(3):
        lookers.append(TWC(CPU, [S(CH(k, f)) for f in (1, 2, 3, 4)] + [S(c) for c in CHV(k)] + [_ts(1 + k)],
                           simple(S(CH(k, 0)))))
    for i in range(136):
        filt = simple(S(0)) if i == 135 else simple(LC([(0, 1), (6 + 135, 1), (6 + i, -1)]))
        lookers.append(TWC(KECCAK_SPONGE, [K(1), S(1), S(2), LC([(3, 1), (5, 1)], i), S(192 + i)] +
                           [Column() for _ in range(7)] + [S(4)], filt))
    for i in range(32):
        lookers.append(TWC(BYTE_PACKING, [S(0), S(33), S(34), LC([(35, 1)] + [(1 + j, j) for j in range(32)], -i),
                                          S(37 + i)] + [Column() for _ in range(7)] + [S(36)],
                           simple(LC([(1 + j, 1) for j in range(i, 32)]))))
    lookers.append(TWC(MEM_BEFORE, [K(0), S(1), S(2), S(3)] + [S(4 + i) for i in range(8)] + [K(0)], simple(S(0))))
    ctls.append(CTL(lookers, TWC(MEMORY, [S(3), S(4), S(5), S(6)] + [S(7 + i) for i in range(8)] + [S(1)],
                                 simple(S(0)))))
    # 7, 8 mem_before / mem_after (all_stark.rs:387-417)
    mem_tuple = lambda: [S(4), S(5), S(6)] + [S(7 + i) for i in range(8)]
    mc_tuple = lambda: [S(1), S(2), S(3)] + [S(4 + i) for i in range(8)]
    ctls.append(CTL([TWC(MEMORY, mem_tuple(), Filter([(S(1), LC([(2, -1)]))], [K(1)]))],
                    TWC(MEM_BEFORE, mc_tuple(), simple(S(0)))))
    ctls.append(CTL([TWC(MEMORY, mem_tuple(), simple(S(26)))], TWC(MEM_AFTER, mc_tuple(), simple(S(0)))))
    # 9 context pruning (all_stark.rs:378-383)
    ctls.append(CTL([TWC(MEMORY, [LC([(21, 1)], -1)], Filter([], [S(22)]))],
                    TWC(CPU, [S(CTX)], prod(S(CONTEXT_OP), S(GENERAL0)))))
    # insert order of all_stark.rs:153-172: context_pruning is LAST, mem_before/after before it
    return ctls


def build_lookups(cdk_erigon=False):
    """Stark::lookups() per table (arithmetic_stark.rs:320, byte_packing_stark.rs:426, keccak_sponge_stark.rs:946,
    memory_stark.rs:858); the cdk_erigon Poseidon table has none."""
    out = [[] for _ in range(NUM_TABLES + (1 if cdk_erigon else 0))]
    out[ARITHMETIC] = [Lookup([S(18 + i) for i in range(96)], S(114), S(115), [Filter() for _ in range(96)])]
    out[BYTE_PACKING] = [Lookup([S(37 + i) for i in range(32)], S(69), S(70), [Filter() for _ in range(32)])]
    out[KECCAK_SPONGE] = [Lookup([S(192 + i) for i in range(136)], S(436), S(437), [Filter() for _ in range(136)])]
    out[MEMORY] = [Lookup([S(27), N(6)], S(28), S(29), [Filter(), simple(LC([(15, 1), (16, 1)]))]),
                   Lookup([LC([(4, 1)], 1)], S(21), S(23), [simple(S(24))])]
    return out


class Registry:
    """The per-feature-set table list: eth_mainnet (9 tables, 10 CTLs) or cdk_erigon (10 tables, 13 CTLs)."""

    def __init__(self, cdk_erigon=False):
        self.cdk_erigon = cdk_erigon
        self.NUM_TABLES = NUM_TABLES + (1 if cdk_erigon else 0)
        self.OPTIONAL_TABLES = OPTIONAL_TABLES + ((9,) if cdk_erigon else ())
        self.TABLE_COLUMNS = (116, 71, 86, 2431, 438, 523, 30, 12, 12, 322) if cdk_erigon else TABLE_COLUMNS
        self.TABLE_AIR = (5, 4, 10, 6, 7, 2, 3, 1, 1, 9) if cdk_erigon else TABLE_AIR
        self.ctls = build_ctls(cdk_erigon)
        self.lookups = build_lookups(cdk_erigon)
