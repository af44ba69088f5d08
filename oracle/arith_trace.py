"""oracle/arith_trace.py -- TEST INFRASTRUCTURE ONLY.
Restatement of the Arithmetic table's witness generators, evm_arithmetization/src/arithmetic/:
  addcy.rs:31-65 (ADD/SUB/LT/GT), mul.rs:72-121, modular.rs:211-382 (`generate_modular_op`, ADDMOD/SUBMOD/MULMOD and
  the FP254 variants), divmod.rs:24-84 (DIV/MOD), shift.rs:41-85 (SHL/SHR), byte.rs:101-200, mod.rs:253-359
  (`to_rows`, range-check rows), arithmetic_stark.rs:130-190 (`generate_trace`, `generate_range_checks`),
  polynomial helpers utils.rs:14-302.  Python ints stand in for the reference's i64 limbs and BigInts.
tests/test_oracle_tracegen.py checks every constraint of the restated AIR on the generated rows; the GPU suite proves
the generated table and has the oracle verifier accept it."""
import numpy as np

P = 0xFFFFFFFF00000001
N = 16
LIMB = 16
BASE = 1 << LIMB
M256 = (1 << 256) - 1
AUX_ABS_MAX = 1 << 20
BN_BASE = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
(IS_ADD, IS_MUL, IS_SUB, IS_DIV, IS_MOD, IS_ADDMOD, IS_MULMOD, IS_ADDFP254, IS_MULFP254, IS_SUBFP254, IS_SUBMOD, IS_LT,
 IS_GT, IS_BYTE, IS_SHL, IS_SHR, IS_RANGE_CHECK, OPCODE_COL) = range(18)
IN0, IN1, IN2, OUT, AUX0, AUX1 = 18, 34, 50, 66, 82, 98
NUM_COLS = 116


def _limbs(x, n=N):
    return [(x >> (LIMB * i)) & (BASE - 1) for i in range(n)]


def _put(row, start, x):
    row[start:start + N] = _limbs(x)


def _rd(row, start, n=N):
    return [int(v) for v in row[start:start + n]]


def _f(v):
    return v % P


def pol_remove_root_2exp(a):                       # utils.rs:278-302 (arithmetic shift = floor division)
    q = [0] * len(a)
    q[0] = -(a[0] >> LIMB)
    for d in range(1, len(a) - 1):
        q[d] = (q[d - 1] - a[d]) >> LIMB
    return q


def columns_to_int(limbs):
    return sum(c * (BASE ** i) for i, c in enumerate(limbs))


def int_to_columns(num, n):                         # bigint_to_columns: signed limbs carry the sign of num
    s = -1 if num < 0 else 1
    return [s * c for c in _limbs(abs(num), n)]


def gen_addcy(row, filt, a, b):
    _put(row, IN0, a); _put(row, IN1, b); _put(row, IN2, 0)
    if filt == IS_ADD:
        s = a + b
        _put(row, AUX0, s >> 256); _put(row, OUT, s & M256)
    elif filt == IS_SUB:
        _put(row, AUX0, 1 if a < b else 0); _put(row, OUT, (a - b) & M256)
    elif filt == IS_LT:
        _put(row, AUX0, (a - b) & M256); _put(row, OUT, 1 if a < b else 0)
    else:
        _put(row, AUX0, (b - a) & M256); _put(row, OUT, 1 if b < a else 0)


def gen_mul_limbs(row, left, right):                # mul.rs:72-109
    prod = [sum(left[i] * right[d - i] for i in range(d + 1)) for d in range(N)]
    out, cy = [0] * N, 0
    for col in range(N):
        t = prod[col] + cy
        cy = t >> LIMB
        out[col] = t & (BASE - 1)
    row[OUT:OUT + N] = out
    prod = [p - o for p, o in zip(prod, out)]
    aux = pol_remove_root_2exp(prod)
    aux[N - 1] = -cy
    aux = [c + AUX_ABS_MAX for c in aux]
    row[AUX0:AUX0 + N] = [c & 0xFFFF for c in aux]
    row[AUX1:AUX1 + N] = [(c >> 16) & 0xFFFF for c in aux]


def gen_modular_op(lv, nv, filt, pol_input, mod_start):          # modular.rs:211-341
    modulus_limbs = _rd(lv, mod_start)
    modulus = columns_to_int(modulus_limbs)
    constr = list(pol_input) + [0]
    mod_is_zero = 0
    if modulus == 0:
        if filt in (IS_DIV, IS_SHR):
            modulus = 1 << 256
        else:
            modulus = 1
            modulus_limbs[0] = 1
        mod_is_zero = 1
    inp = columns_to_int(constr)
    output = inp % modulus                                         # Python's % is already the non-negative residue
    quot = (inp - output) // modulus
    output_limbs = int_to_columns(output, N)
    quot_limbs = int_to_columns(quot, 2 * N)
    out_aux_red = int_to_columns((1 << 256) - modulus + output, N)
    constr = [c - o for c, o in zip(constr, output_limbs + [0] * N)]
    prod = [0] * (3 * N - 1)
    for i, ai in enumerate(quot_limbs):
        for j, bj in enumerate(modulus_limbs):
            prod[i + j] += ai * bj
    constr = [c - p for c, p in zip(constr, prod[:2 * N])]
    assert all(x == 0 for x in prod[2 * N:])
    aux = [c + AUX_ABS_MAX for c in pol_remove_root_2exp(constr)]
    for i in range(2 * N - 1):
        nv[35 + i] = aux[i] & 0xFFFF                               # MODULAR_AUX_INPUT_LO
        nv[66 + i] = (aux[i] >> 16) & 0xFFFF                       # MODULAR_AUX_INPUT_HI
    if filt in (IS_SUBMOD, IS_SUBFP254):
        if quot < 0:
            for i in range(N):
                quot_limbs[i] += 0xFFFF
            quot_limbs[N] = 1
        else:
            quot_limbs[N] = 0
    nv[34] = mod_is_zero                                           # MODULAR_MOD_IS_ZERO
    nv[18:18 + N] = [_f(c) for c in out_aux_red]                   # MODULAR_OUT_AUX_RED
    nv[97] = mod_is_zero * (int(lv[IS_DIV]) + int(lv[IS_SHR]))     # MODULAR_DIV_DENOM_IS_ZERO
    return [_f(c) for c in output_limbs], [_f(c) for c in quot_limbs]


def gen_modular(lv, nv, filt, a, b, m):                            # modular.rs:343-382
    _put(lv, IN0, a); _put(lv, IN1, b); _put(lv, IN2, m)
    x, y = _rd(lv, IN0), _rd(lv, IN1)
    if filt in (IS_ADDMOD, IS_ADDFP254):
        pol = [x[i] + y[i] for i in range(N)] + [0] * (N - 1)
    elif filt in (IS_SUBMOD, IS_SUBFP254):
        pol = [x[i] - y[i] for i in range(N)] + [0] * (N - 1)
    else:
        pol = [0] * (2 * N - 1)
        for i in range(N):
            for j in range(N):
                pol[i + j] += x[i] * y[j]
    out, quo = gen_modular_op(lv, nv, filt, pol, IN2)
    lv[OUT:OUT + N] = out
    lv[AUX0:AUX0 + 2 * N] = quo                                    # MODULAR_QUO_INPUT


def gen_divmod_regs(lv, nv, filt, in_start, mod_start):            # divmod.rs:24-66
    pol = _rd(lv, in_start) + [0] * (N - 1)
    out, quo = gen_modular_op(lv, nv, filt, pol, mod_start)
    assert all(q == 0 for q in quo[N:])
    lv[AUX0:AUX0 + 2 * N] = 0
    if filt in (IS_DIV, IS_SHR):
        assert [int(v) for v in lv[OUT:OUT + N]] == quo[:N]
        lv[AUX0:AUX0 + N] = out
    else:
        assert [int(v) for v in lv[OUT:OUT + N]] == out
        lv[AUX0:AUX0 + N] = quo[:N]


def gen_byte(row, idx, val):                                       # byte.rs:109-200
    _put(row, IN0, idx); _put(row, IN1, val)
    for i in range(5):
        row[AUX0 + i] = (idx >> i) & 1
    row[AUX0 + 5] = ((idx & 0xFFFF) >> 5)
    hi_sum = (int(row[AUX0 + 5]) + sum(int(v) for v in row[IN0 + 1:IN0 + N])) % P
    inv = pow(hi_sum, P - 2, P) if hi_sum else 1
    for k in range(4):
        row[91 + k] = (inv >> (16 * k)) & 0xFFFF
    row[90] = 1 if hi_sum else 0
    i, src, dest = 3, IN1, AUX1
    while True:
        lvl = 1 << i
        src += (0 if (idx >> (i + 1)) & 1 else 1) * lvl
        row[dest:dest + lvl] = row[src:src + lvl].copy()
        if i == 0:
            break
        src, dest, i = dest, dest + lvl, i - 1
    t = int(row[dest])
    lo, hi = t & 0xFF, t >> 8
    row[88], row[89] = lo << 8, hi
    out = lo if idx & 1 else hi
    row[AUX1 + 15] = out
    _put(row, OUT, out if idx < 32 else 0)


def to_rows(op):
    """op = (kind, args...) -> list of 1 or 2 rows (numpy uint64[116])."""
    kind = op[0]
    r1, r2 = np.zeros(NUM_COLS, dtype=np.uint64), np.zeros(NUM_COLS, dtype=np.uint64)
    if kind == "range_check":
        _, a, b, c, opcode, res = op
        r1[IS_RANGE_CHECK], r1[OPCODE_COL] = 1, opcode
        _put(r1, IN0, a); _put(r1, IN1, b); _put(r1, IN2, c); _put(r1, OUT, res)
        return [r1]
    filt = op[1]
    r1[filt] = 1
    if filt in (IS_ADD, IS_SUB, IS_LT, IS_GT):
        gen_addcy(r1, filt, op[2], op[3])
        return [r1]
    if filt == IS_MUL:
        _put(r1, IN0, op[2]); _put(r1, IN1, op[3]); _put(r1, IN2, 0)
        gen_mul_limbs(r1, _rd(r1, IN0), _rd(r1, IN1))
        return [r1]
    if filt in (IS_DIV, IS_MOD):
        a, b = op[2], op[3]
        res = (a // b if b else 0) if filt == IS_DIV else (a % b if b else 0)
        _put(r1, IN0, a); _put(r1, IN1, b); _put(r1, OUT, res)
        gen_divmod_regs(r1, r2, filt, IN0, IN1)
        return [r1, r2]
    if filt in (IS_SHL, IS_SHR):
        shift, inp = op[2], op[3]
        res = ((inp << shift) & M256 if shift < 256 else 0) if filt == IS_SHL else (inp >> shift if shift < 256 else 0)
        _put(r1, IN0, shift); _put(r1, IN1, inp); _put(r1, OUT, res)
        _put(r1, IN2, 0 if shift > 255 else 1 << shift)
        if filt == IS_SHL:
            gen_mul_limbs(r1, _rd(r1, IN1), _rd(r1, IN2))
            return [r1]
        gen_divmod_regs(r1, r2, IS_SHR, IN1, IN2)
        return [r1, r2]
    if filt == IS_BYTE:
        gen_byte(r1, op[2], op[3])
        return [r1]
    m = BN_BASE if filt in (IS_ADDFP254, IS_MULFP254, IS_SUBFP254) else op[4]
    gen_modular(r1, r2, filt, op[2], op[3], m)
    return [r1, r2]


def generate_trace(operations):
    """arithmetic_stark.rs:158-190 -> (116, n) uint64, n = max(next_power_of_two(rows), 2^16)."""
    rows = []
    for op in operations:
        rows += to_rows(op)
    n = max(1 << max(len(rows) - 1, 0).bit_length(), 1 << 16)
    t = np.zeros((NUM_COLS, n), dtype=np.uint64)
    if rows:
        t[:, :len(rows)] = np.array(rows, dtype=np.uint64).T
    t[114] = np.minimum(np.arange(n, dtype=np.uint64), np.uint64(65535))
    t[115] = 0
    t[115, :1 << 16] = np.bincount(t[18:114].astype(np.int64).reshape(-1), minlength=1 << 16).astype(np.uint64)
    return t, len(rows)
