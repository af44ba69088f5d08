"""oracle/airs.py -- TEST INFRASTRUCTURE ONLY.
Pure-Python restatements of the reference tables' `eval_packed_generic` (constraints in the
reference's yield order), each citing the reference file it follows.  lv / nv are lists of ints."""
P = 0xFFFFFFFF00000001


def eval_none(lv, nv, c):
    pass


def eval_mem_continuation(lv, nv, c):
    # evm_arithmetization/src/memory_continuation/memory_continuation_stark.rs:110-122
    f = lv[0]
    c.constraint(f * (f - 1))


def eval_logic(lv, nv, c):
    # evm_arithmetization/src/logic.rs:249-303 (columns logic.rs:46-71)
    is_and, is_or, is_xor = lv[0], lv[1], lv[2]
    for flag in (is_and, is_or, is_xor):
        c.constraint(flag * (flag - 1))
    all_flags = is_and + is_or + is_xor
    c.constraint(all_flags * (all_flags - 1))
    sum_coeff = is_or + is_xor
    and_coeff = is_and - is_or - 2 * is_xor
    in0, in1, res = lv[3:259], lv[259:515], lv[515:523]
    for bits in (in0, in1):
        for b in bits:
            c.constraint(b * (b - 1))
    for limb in range(8):
        xb, yb = in0[32 * limb:32 * limb + 32], in1[32 * limb:32 * limb + 32]
        x = sum(b * (1 << i) for i, b in enumerate(xb))
        y = sum(b * (1 << i) for i, b in enumerate(yb))
        x_land_y = sum(a * b * (1 << i) for i, (a, b) in enumerate(zip(xb, yb)))
        c.constraint(res[limb] - (sum_coeff * (x + y) + and_coeff * x_land_y))


AIRS = {0: (eval_none, None), 1: (eval_mem_continuation, 12), 2: (eval_logic, 523)}


def eval_memory(lv, nv, c):
    # evm_arithmetization/src/memory/memory_stark.rs:474-626; columns memory/columns.rs:13-94
    SEG_CODE, SEG_TRIE_DATA, SEG_ACC, SEG_STO = 0, 12, 34, 35   # memory/segments.rs (unscaled)
    filt, timestamp, timestamp_inv, is_read = lv[0], lv[1], lv[2], lv[3]
    ctx, seg, virt = lv[4], lv[5], lv[6]
    vals, nvals = lv[7:15], nv[7:15]
    cfc, sfc, vfc = lv[15], lv[16], lv[17]
    initialize_aux, preinit, preinit_aux = lv[18], lv[19], lv[20]
    is_stale, maybe_in_mem_after, mem_after_filter = lv[24], lv[25], lv[26]
    range_check = lv[27]
    n_ts, n_is_read, n_ctx, n_seg, n_virt = nv[1], nv[3], nv[4], nv[5], nv[6]
    c.constraint(filt * (filt - 1))
    c.constraint((1 - filt) * (1 - is_read))
    au = 1 - cfc - sfc - vfc
    not_au = 1 - au
    c.constraint(cfc * (1 - cfc))
    c.constraint(sfc * (1 - sfc))
    c.constraint(vfc * (1 - vfc))
    c.constraint(au * not_au)
    c.constraint_transition(sfc * (n_ctx - ctx))
    c.constraint_transition(vfc * (n_ctx - ctx))
    c.constraint_transition(vfc * (n_seg - seg))
    c.constraint_transition(au * (n_ctx - ctx))
    c.constraint_transition(au * (n_seg - seg))
    c.constraint_transition(au * (n_virt - virt))
    crc = cfc * (n_ctx - ctx - 1) + sfc * (n_seg - seg - 1) + vfc * (n_virt - virt - 1) + au * (n_ts - timestamp)
    c.constraint_transition(range_check - crc)
    c.constraint_transition(preinit_aux - (n_seg - SEG_ACC) * (n_seg - SEG_STO))
    c.constraint_transition(preinit - (n_seg - SEG_CODE) * (n_seg - SEG_TRIE_DATA) * preinit_aux)
    c.constraint_transition(initialize_aux - preinit * not_au * n_is_read)
    for i in range(8):
        c.constraint_transition(n_is_read * au * (nvals[i] - vals[i]))
        c.constraint_transition(initialize_aux * nvals[i])
    c.constraint_transition(maybe_in_mem_after + filt * not_au * (is_stale - 1))
    c.constraint(mem_after_filter * (mem_after_filter - 1))
    for i in range(8):
        c.constraint((mem_after_filter - maybe_in_mem_after) * preinit * vals[i])
    c.constraint(timestamp * (timestamp * timestamp_inv - 1))
    c.constraint_first_row(lv[28])
    c.constraint_transition(nv[28] - lv[28] - 1)


def eval_byte_packing(lv, nv, c):
    # evm_arithmetization/src/byte_packing/byte_packing_stark.rs:296-352; columns byte_packing/columns.rs:12-40
    NB, IDX, VAL = 32, 1, 37
    rc1, rc2 = lv[69], nv[69]
    c.constraint_first_row(rc1)
    incr = rc2 - rc1
    c.constraint_transition(incr * incr - incr)
    c.constraint_last_row(rc1 - 255)
    cur = sum(lv[IDX:IDX + NB])
    c.constraint(cur * (cur - 1))
    c.constraint_first_row(cur - 1)
    c.constraint(lv[0] * (lv[0] - 1))
    for i in range(NB):
        c.constraint(lv[IDX + i] * (lv[IDX + i] - 1))
    nxt = sum(nv[IDX:IDX + NB])
    c.constraint_transition(nxt * (nxt - cur))
    for i in range(NB - 1):
        for j in range(i + 1, NB):
            c.constraint(lv[IDX + i] * lv[VAL + j])


AIRS.update({3: (eval_memory, 30), 4: (eval_byte_packing, 71)})


# ---- ArithmeticStark -----------------------------------------------------------------------------
# evm_arithmetization/src/arithmetic/{arithmetic_stark.rs:203-252, mul.rs:123-185, addcy.rs:98-172,
# divmod.rs:86-145, modular.rs:382-612, byte.rs:201-296, shift.rs:85-128, utils.rs, columns.rs}
N_LIMBS = 16
A_IS = dict(ADD=0, MUL=1, SUB=2, DIV=3, MOD=4, ADDMOD=5, MULMOD=6, ADDFP254=7, MULFP254=8, SUBFP254=9,
            SUBMOD=10, LT=11, GT=12, BYTE=13, SHL=14, SHR=15, RANGE_CHECK=16)
A_OPCODE = 17
A_IN0, A_IN1, A_IN2, A_OUT, A_AUX0, A_AUX1 = 18, 34, 50, 66, 82, 98
A_RANGE_COUNTER, A_RC_FREQ = 114, 115
A_BASE = 1 << 16
A_OFFSET = 1 << 20          # AUX_COEFF_ABS_MAX
A_OVERFLOW_INV = 18446462594437939201   # GOLDILOCKS_INVERSE_65536 (addcy.rs:67)
BN_BASE = [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029]
BN254_LIMBS = [(BN_BASE[i // 4] >> (16 * (i % 4))) & 0xFFFF for i in range(16)]


def _rd(v, start, n=N_LIMBS):
    return list(v[start:start + n])


def _pol_adjoin_root(a, root):
    res = [(-root) * a[0]]
    for d in range(1, len(a)):
        res.append(a[d - 1] - root * a[d])
    return res


def _arith_addcy(c, filt, x, y, z, given_cy, two_row):
    cy = 0
    for xi, yi, zi in zip(x, y, z):
        t = cy + xi + yi - zi
        (c.constraint_transition if two_row else c.constraint)(filt * t * (A_BASE - t))
        cy = t * A_OVERFLOW_INV
    if two_row:
        c.constraint_transition(filt * (cy - given_cy[0]))
        for i in range(1, N_LIMBS):
            c.constraint_transition(filt * given_cy[i])
    else:
        c.constraint(filt * given_cy[0] * (given_cy[0] - 1))
        c.constraint(filt * (cy - given_cy[0]))
        for i in range(1, N_LIMBS):
            c.constraint(filt * given_cy[i])


def _arith_mul(lv, c, filt, left, right):
    out = _rd(lv, A_OUT)
    aux = [lv[A_AUX0 + i] + lv[A_AUX1 + i] * A_BASE - A_OFFSET for i in range(N_LIMBS)]
    cp = [sum(left[i] * right[d - i] for i in range(d + 1)) for d in range(N_LIMBS)]   # pol_mul_lo
    cp = [a - b for a, b in zip(cp, out)]
    adj = _pol_adjoin_root(aux, A_BASE)
    cp = [a - b for a, b in zip(cp, adj)]
    for x in cp:
        c.constraint(filt * x)


def _modular_constr_poly(lv, nv, c, filt, output, modulus, quot):
    output, modulus = list(output), list(modulus)
    mod_is_zero = nv[34]
    c.constraint_transition(filt * (mod_is_zero * mod_is_zero - mod_is_zero))
    limb_sum = sum(modulus)
    c.constraint_transition(filt * limb_sum * mod_is_zero)
    modulus[0] = modulus[0] + mod_is_zero
    div_denom_is_zero = nv[97]
    c.constraint_transition(filt * (mod_is_zero * (lv[A_IS["DIV"]] + lv[A_IS["SHR"]]) - div_denom_is_zero))
    output[0] = output[0] + div_denom_is_zero
    # check_reduced
    is_less_than = [0] * N_LIMBS
    is_less_than[0] = 1 - mod_is_zero * (lv[A_IS["DIV"]] + lv[A_IS["SHR"]])
    _arith_addcy(c, filt, modulus, _rd(nv, 18), output, is_less_than, True)
    output[0] = output[0] - div_denom_is_zero
    prod = [0] * (3 * N_LIMBS - 1)                           # pol_mul_wide2(quot, modulus)
    for i, ai in enumerate(quot):
        for j, bj in enumerate(modulus):
            prod[i + j] = prod[i + j] + ai * bj
    for x in prod[2 * N_LIMBS:]:
        c.constraint_transition(filt * x)
    cp = prod[:2 * N_LIMBS]
    for i in range(N_LIMBS):
        cp[i] = cp[i] + output[i]
    aux = [0] * (2 * N_LIMBS)
    for i in range(2 * N_LIMBS - 1):
        aux[i] = nv[35 + i] - A_OFFSET                       # MODULAR_AUX_INPUT_LO = 35..66
    for i in range(2 * N_LIMBS - 1):
        aux[i] = aux[i] + A_BASE * nv[66 + i]                # MODULAR_AUX_INPUT_HI = 66..97
    adj = _pol_adjoin_root(aux, A_BASE)
    return [a + b for a, b in zip(cp, adj)]


def _submod_constr_poly(lv, nv, c, filt, output, modulus, quot):
    quot = list(quot)
    sign = quot[N_LIMBS]
    c.constraint(filt * sign * (sign - 1))
    for i in range(N_LIMBS):
        quot[i] = quot[i] - 0xFFFF * sign
    quot[N_LIMBS] = 0
    for d in quot[N_LIMBS:]:
        c.constraint(filt * d)
    return _modular_constr_poly(lv, nv, c, filt, output, modulus, quot)


def _arith_divmod_helper(lv, nv, c, filt, num_s, den_s, quo_s, rem_s):
    c.constraint_last_row(filt)
    num = _rd(lv, num_s)
    den = _rd(lv, den_s)
    quo = _rd(lv, quo_s) + [0] * N_LIMBS
    rem = _rd(lv, rem_s)
    cp = _modular_constr_poly(lv, nv, c, filt, rem, den, quo)
    for i in range(N_LIMBS):
        cp[i] = cp[i] - num[i]
    for x in cp:
        c.constraint_transition(filt * x)


def eval_arithmetic(lv, nv, c):
    # arithmetic_stark.rs:203-252
    for f in range(17):
        c.constraint(lv[f] * (lv[f] - 1))
    all_flags = sum(lv[0:17])
    c.constraint(all_flags * (all_flags - 1))
    c.constraint((1 - lv[A_IS["RANGE_CHECK"]]) * lv[A_OPCODE])
    rc1, rc2 = lv[A_RANGE_COUNTER], nv[A_RANGE_COUNTER]
    c.constraint_first_row(rc1)
    incr = rc2 - rc1
    c.constraint_transition(incr * incr - incr)
    c.constraint_last_row(rc1 - 65535)
    # mul.rs:177-185
    _arith_mul(lv, c, lv[A_IS["MUL"]], _rd(lv, A_IN0), _rd(lv, A_IN1))
    # addcy.rs:153-172
    in0, in1, out, aux = _rd(lv, A_IN0), _rd(lv, A_IN1), _rd(lv, A_OUT), _rd(lv, A_AUX0)
    _arith_addcy(c, lv[A_IS["ADD"]], in0, in1, out, aux, False)
    _arith_addcy(c, lv[A_IS["SUB"]], in1, out, in0, aux, False)
    _arith_addcy(c, lv[A_IS["LT"]], in1, aux, in0, out, False)
    _arith_addcy(c, lv[A_IS["GT"]], in0, aux, in1, out, False)
    # divmod.rs:118-145
    _arith_divmod_helper(lv, nv, c, lv[A_IS["DIV"]], A_IN0, A_IN1, A_OUT, A_AUX0)
    _arith_divmod_helper(lv, nv, c, lv[A_IS["MOD"]], A_IN0, A_IN1, A_AUX0, A_OUT)
    # modular.rs:542-612
    bn = lv[A_IS["ADDFP254"]] + lv[A_IS["MULFP254"]] + lv[A_IS["SUBFP254"]]
    filt = lv[A_IS["ADDMOD"]] + lv[A_IS["SUBMOD"]] + lv[A_IS["MULMOD"]] + bn
    c.constraint_last_row(filt)
    modulus = _rd(lv, A_IN2)
    for mi, bi in zip(modulus, BN254_LIMBS):
        c.constraint_transition(bn * (mi - bi))
    output = _rd(lv, A_OUT)
    quo_input = _rd(lv, A_AUX0, 2 * N_LIMBS)
    add_f = lv[A_IS["ADDMOD"]] + lv[A_IS["ADDFP254"]]
    sub_f = lv[A_IS["SUBMOD"]] + lv[A_IS["SUBFP254"]]
    mul_f = lv[A_IS["MULMOD"]] + lv[A_IS["MULFP254"]]
    sub_cp = _submod_constr_poly(lv, nv, c, sub_f, output, modulus, quo_input)
    mod_cp = _modular_constr_poly(lv, nv, c, add_f + mul_f, output, modulus, quo_input)
    i0, i1 = _rd(lv, A_IN0), _rd(lv, A_IN1)
    add_in = [a + b for a, b in zip(i0, i1)] + [0] * (N_LIMBS - 1)
    sub_in = [a - b for a, b in zip(i0, i1)] + [0] * (N_LIMBS - 1)
    mul_in = [0] * (2 * N_LIMBS - 1)
    for i, ai in enumerate(i0):
        for j, bj in enumerate(i1):
            mul_in[i + j] = mul_in[i + j] + ai * bj
    for inp, f, cp in ((add_in, add_f, mod_cp), (sub_in, sub_f, sub_cp), (mul_in, mul_f, mod_cp)):
        cpc = list(cp)
        for i in range(2 * N_LIMBS - 1):
            cpc[i] = cpc[i] - inp[i]
        for x in cpc:
            c.constraint_transition(f * x)
    # byte.rs:201-296
    is_byte = lv[A_IS["BYTE"]]
    idx, val, outb = _rd(lv, A_IN0), _rd(lv, A_IN1), _rd(lv, A_OUT)
    dec, tree = _rd(lv, A_AUX0), _rd(lv, A_AUX1)
    idx0_lo5 = 0
    for i in range(5):
        bit = dec[i]
        c.constraint(is_byte * (bit * bit - bit))
        idx0_lo5 = idx0_lo5 + bit * (1 << i)
    idx0_hi = dec[5] * 32
    c.constraint(is_byte * (idx[0] - (idx0_lo5 + idx0_hi)))
    bit = dec[4]
    for i in range(8):
        c.constraint(is_byte * (tree[i] - (bit * val[i] + (1 - bit) * val[i + 8])))
    bit = dec[3]
    for i in range(4):
        c.constraint(is_byte * (tree[i + 8] - (bit * tree[i] + (1 - bit) * tree[i + 4])))
    bit = dec[2]
    for i in range(2):
        c.constraint(is_byte * (tree[i + 12] - (bit * tree[i + 8] + (1 - bit) * tree[i + 10])))
    bit = dec[1]
    limb = bit * tree[12] + (1 - bit) * tree[13]
    c.constraint(is_byte * (tree[14] - limb))
    base8 = 256
    lo_byte, hi_byte = lv[88], lv[89]
    c.constraint(is_byte * (lo_byte + base8 * (base8 * hi_byte - limb)))
    bit = dec[0]
    t = bit * lo_byte + (1 - bit) * base8 * hi_byte
    c.constraint(is_byte * (base8 * tree[15] - t))
    expected = tree[15]
    hi_limb_sum = lv[87] + sum(idx[1:])
    idx_is_large = lv[90]
    c.constraint(is_byte * (idx_is_large * idx_is_large - idx_is_large))
    c.constraint(is_byte * hi_limb_sum * (idx_is_large - 1))
    hi_inv = lv[91] + lv[92] * (1 << 16) + lv[93] * (1 << 32) + lv[94] * (1 << 48)
    c.constraint(is_byte * (hi_limb_sum * hi_inv - idx_is_large))
    c.constraint(is_byte * (outb[0] - (1 - idx_is_large) * expected))
    for i in range(1, N_LIMBS):
        c.constraint(is_byte * outb[i])
    # shift.rs:85-128
    _arith_mul(lv, c, lv[A_IS["SHL"]], _rd(lv, A_IN1), _rd(lv, A_IN2))
    _arith_divmod_helper(lv, nv, c, lv[A_IS["SHR"]], A_IN1, A_IN2, A_OUT, A_AUX0)


AIRS.update({5: (eval_arithmetic, 116)})


# ---- KeccakStark -----------------------------------------------------------------------------------
# evm_arithmetization/src/keccak/{keccak_stark.rs:266-426, round_flags.rs:14-60, logic.rs:15-53,
# columns.rs:7-134, constants.rs}
K_ROUNDS = 24
K_TIMESTAMP = 24
K_R = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
K_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
        0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
        0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
        0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
        0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]


def k_reg_a(x, y): return 25 + (x * 5 + y) * 2
def k_reg_c(x, z): return 75 + x * 64 + z
def k_reg_c_prime(x, z): return 395 + x * 64 + z
def k_reg_a_prime(x, y, z): return 715 + x * 320 + y * 64 + z
def k_reg_b(x, y, z):
    a, b = (x + 3 * y) % 5, x
    return k_reg_a_prime(a, b, (z + 64 - K_R[a][b]) % 64)
def k_reg_a_pp(x, y): return 2315 + x * 10 + y * 2
def k_reg_a_pp_00_bit(i): return 2365 + i
def k_reg_a_ppp(x, y): return 2429 if (x == 0 and y == 0) else k_reg_a_pp(x, y)


def _xor_gen(x, y): return x + y - x * (y + y)
def _xor3_gen(x, y, z): return _xor_gen(x, _xor_gen(y, z))
def _andn_gen(x, y): return (1 - x) * y


def _fold_bits(get_bit, lo, hi):
    acc = 0
    for z in range(hi - 1, lo - 1, -1):
        acc = acc + acc + get_bit(z)
    return acc


def eval_keccak(lv, nv, c):
    # round_flags.rs:14-60
    for i in range(K_ROUNDS):
        c.constraint(lv[i] * (lv[i] - 1))
    local_any = sum(lv[0:K_ROUNDS])
    c.constraint_first_row(local_any * (lv[0] - 1))
    for i in range(1, K_ROUNDS):
        c.constraint_first_row(local_any * lv[i])
    cur_any = local_any
    next_any = sum(nv[0:K_ROUNDS])
    last_round_flag = lv[K_ROUNDS - 1]
    padding = (next_any - 1) * cur_any * (last_round_flag - 1)
    for i in range(K_ROUNDS):
        c.constraint_transition(next_any * (nv[(i + 1) % K_ROUNDS] - lv[i]) + padding)
    c.constraint_transition(next_any * (cur_any - 1))
    # keccak_stark.rs:281-425
    not_final_step = 1 - lv[K_ROUNDS - 1]
    c.constraint(local_any * not_final_step * (nv[K_TIMESTAMP] - lv[K_TIMESTAMP]))
    for x in range(5):
        for z in range(64):
            xor = _xor3_gen(lv[k_reg_c(x, z)], lv[k_reg_c((x + 4) % 5, z)], lv[k_reg_c((x + 1) % 5, (z + 63) % 64)])
            c.constraint(lv[k_reg_c_prime(x, z)] - xor)
    for x in range(5):
        for y in range(5):
            gb = lambda z, x=x, y=y: _xor3_gen(lv[k_reg_a_prime(x, y, z)], lv[k_reg_c(x, z)], lv[k_reg_c_prime(x, z)])
            c.constraint(_fold_bits(gb, 0, 32) - lv[k_reg_a(x, y)])
            c.constraint(_fold_bits(gb, 32, 64) - lv[k_reg_a(x, y) + 1])
    for x in range(5):
        for z in range(64):
            s = sum(lv[k_reg_a_prime(x, i, z)] for i in range(5))
            diff = s - lv[k_reg_c_prime(x, z)]
            c.constraint(diff * (diff - 2) * (diff - 4))
    for x in range(5):
        for y in range(5):
            gb = lambda z, x=x, y=y: _xor_gen(lv[k_reg_b(x, y, z)],
                                              _andn_gen(lv[k_reg_b((x + 1) % 5, y, z)], lv[k_reg_b((x + 2) % 5, y, z)]))
            c.constraint(_fold_bits(gb, 0, 32) - lv[k_reg_a_pp(x, y)])
            c.constraint(_fold_bits(gb, 32, 64) - lv[k_reg_a_pp(x, y) + 1])
    bits = [lv[k_reg_a_pp_00_bit(i)] for i in range(64)]
    c.constraint(_fold_bits(lambda z: bits[z], 0, 32) - lv[k_reg_a_pp(0, 0)])
    c.constraint(_fold_bits(lambda z: bits[z], 32, 64) - lv[k_reg_a_pp(0, 0) + 1])

    def xored_bit(i):
        rc_bit = 0
        for r in range(K_ROUNDS):
            rc_bit = rc_bit + lv[r] * ((K_RC[r] >> i) & 1)
        return _xor_gen(bits[i], rc_bit)
    c.constraint(_fold_bits(xored_bit, 0, 32) - lv[k_reg_a_ppp(0, 0)])
    c.constraint(_fold_bits(xored_bit, 32, 64) - lv[k_reg_a_ppp(0, 0) + 1])
    is_last_round = lv[K_ROUNDS - 1]
    not_last_round = 1 - is_last_round
    for x in range(5):
        for y in range(5):
            c.constraint_transition(not_last_round * (lv[k_reg_a_ppp(x, y)] - nv[k_reg_a(x, y)]))
            c.constraint_transition(not_last_round * (lv[k_reg_a_ppp(x, y) + 1] - nv[k_reg_a(x, y) + 1]))


AIRS.update({6: (eval_keccak, 2431)})


def eval_keccak_sponge(lv, nv, c):
    # evm_arithmetization/src/keccak_sponge/keccak_sponge_stark.rs:546-715; columns keccak_sponge/columns.rs:31-95
    RATE, RATE_U32, CAP_U32, DIG_U32 = 136, 34, 16, 8
    PAD, ORATE, OCAP, BLOCK, PARTIAL, DIGEST, RC = 6, 142, 176, 192, 362, 404, 436
    rc1, rc2 = lv[RC], nv[RC]
    c.constraint_first_row(rc1)
    incr = rc2 - rc1
    c.constraint_transition(incr * incr - incr)
    c.constraint_last_row(rc1 - 255)
    full = lv[0]
    c.constraint(full * (full - 1))
    for i in range(RATE):
        c.constraint(lv[PAD + i] * (lv[PAD + i] - 1))
    is_final = lv[PAD + RATE - 1]
    for i in range(1, RATE):
        c.constraint(lv[PAD + i - 1] * (lv[PAD + i] - 1))
    c.constraint(is_final * full)
    absorbed = lv[5]
    c.constraint_first_row(absorbed)
    for i in range(RATE_U32):
        c.constraint_first_row(lv[ORATE + i])
    for i in range(CAP_U32):
        c.constraint_first_row(lv[OCAP + i])
    c.constraint_transition(is_final * nv[5])
    for i in range(RATE_U32):
        c.constraint_transition(is_final * nv[ORATE + i])
    for i in range(CAP_U32):
        c.constraint_transition(is_final * nv[OCAP + i])
    c.constraint_transition(full * (lv[1] - nv[1]))
    c.constraint_transition(full * (lv[2] - nv[2]))
    c.constraint_transition(full * (lv[3] - nv[3]))
    c.constraint_transition(full * (lv[4] - nv[4]))
    for k in range(DIG_U32):
        cur = lv[DIGEST + 4 * k]
        for i in range(1, 4):
            cur = cur + lv[DIGEST + 4 * k + i] * (1 << (8 * i))
        c.constraint_transition(full * (nv[ORATE + k] - cur))
    for k in range(RATE_U32 - DIG_U32):            # zip(partial[0..42], next.original_rate[8..34]) -> 26 pairs
        c.constraint_transition(full * (nv[ORATE + DIG_U32 + k] - lv[PARTIAL + k]))
    for k in range(CAP_U32):                        # partial.skip(26) zip next.original_capacity (16)
        c.constraint_transition(full * (nv[OCAP + k] - lv[PARTIAL + (RATE_U32 - DIG_U32) + k]))
    c.constraint_transition(full * (absorbed + RATE - nv[5]))
    single = lv[PAD + RATE - 1] - lv[PAD + RATE - 2]
    c.constraint_transition(single * (lv[BLOCK + RATE - 1] - 0b10000001))
    for i in range(RATE - 1):
        first = lv[PAD + i] - lv[PAD + i - 1] if i > 0 else lv[PAD + i]
        c.constraint_transition(first * (lv[BLOCK + i] - 1))
        c.constraint_transition(lv[PAD + i] * (first - 1) * lv[BLOCK + i])
    c.constraint_transition(is_final * (single - 1) * (lv[BLOCK + RATE - 1] - 0b10000000))
    is_dummy = 1 - full - is_final
    c.constraint_transition(is_dummy * (nv[0] + nv[PAD + RATE - 1]))


AIRS.update({7: (eval_keccak_sponge, 438)})


# ---- CpuStark ------------------------------------------------------------------------------------------
# evm_arithmetization/src/cpu/cpu_stark.rs:594-626 and its 18 modules, in call order.  Columns
# cpu/columns/{mod.rs:56-97, ops.rs:6-47, general.rs}: context 0, code_context 1, program_counter 2,
# stack_len 3, is_kernel_mode 4, gas 5, op flags 6..23, opcode_bits 24..31, general (union) 32..39,
# clock 40, mem_channels[3] 41..79 (13 each), partial_channel 80..84.
C_CTX, C_CODE_CTX, C_PC, C_STACK_LEN, C_KERNEL, C_GAS = 0, 1, 2, 3, 4, 5
C_OPS = ["binary_op", "ternary_op", "fp254_op", "eq_iszero", "logic_op", "not_pop", "shift",
         "jumpdest_keccak_general", "jumps", "push_prover_input", "dup_swap", "context_op", "m_op_32bytes",
         "exit_kernel", "m_op_general", "pc_push0", "syscall", "exception"]
C_OP = {name: 6 + i for i, name in enumerate(C_OPS)}
C_BITS, C_GEN, C_CLOCK = 24, 32, 40
# `cdk_erigon`: one more flag after jumpdest_keccak_general (cpu/columns/ops.rs:22-25); every later column moves by one
C_OPS_ERIGON = C_OPS[:8] + ["poseidon"] + C_OPS[8:]
SEG_STACK, SEG_SHIFT_TABLE, SEG_JUMPDEST_BITS, SEG_CODE_ = 1, 13, 14, 0


class _Chan:
    def __init__(self, v, base, partial=False):
        self.used, self.is_read, self.addr_context, self.addr_segment, self.addr_virtual = v[base:base + 5]
        self.value = None if partial else list(v[base + 5:base + 13])


class _CpuRow:
    def __init__(self, v, erigon=False):
        self.v = v
        names = C_OPS_ERIGON if erigon else C_OPS
        x = 1 if erigon else 0
        self.context, self.code_context, self.program_counter = v[0], v[1], v[2]
        self.stack_len, self.is_kernel_mode, self.gas = v[3], v[4], v[5]
        self.op = {name: v[6 + i] for i, name in enumerate(names)}
        self.ops = [v[6 + i] for i in range(len(names))]  # struct field order
        self.opcode_bits = list(v[24 + x:32 + x])
        self.g = list(v[32 + x:40 + x])
        self.clock = v[40 + x]
        self.mem_channels = [_Chan(v, 41 + x + 13 * k) for k in range(3)]
        self.partial_channel = _Chan(v, 80 + x, True)
    # general (union) views
    @property
    def exc_code_bits(self): return self.g[0:3]
    @property
    def diff_pinv(self): return self.g
    @property
    def should_jump(self): return self.g[0]
    @property
    def cond_sum_pinv(self): return self.g[1]
    @property
    def high_limb_sum_inv(self): return self.g[0]
    @property
    def stack_inv(self): return self.g[4]
    @property
    def stack_inv_aux(self): return self.g[5]
    @property
    def stack_inv_aux_2(self): return self.g[6]
    @property
    def stack_len_bounds_aux(self): return self.g[7]
    @property
    def is_not_kernel(self): return self.g[0]
    @property
    def pruning_flag(self): return self.g[0]


# stack.rs:42-171: (num_pops, pushes, disable_other_channels), struct field order
_SB = {"binary_op": (2, True, True), "ternary_op": (3, True, True), "fp254_op": (2, True, True),
       "eq_iszero": None, "logic_op": (2, True, True), "not_pop": None, "shift": (2, True, False),
       "jumpdest_keccak_general": None, "jumps": None, "push_prover_input": (0, True, True),
       "dup_swap": None, "context_op": None, "m_op_32bytes": (2, True, False), "exit_kernel": (1, False, True),
       "m_op_general": None, "pc_push0": (0, True, True), "syscall": (0, True, False), "exception": (0, True, False)}
_MIGHT_OVERFLOW = {"push_prover_input", "pc_push0", "dup_swap", "exit_kernel"}
# gas.rs:24-47 (None = handled manually)
_GAS = {"fp254_op": 0, "eq_iszero": 3, "logic_op": 3, "shift": 3, "pc_push0": 2, "dup_swap": 3, "context_op": 0,
        "m_op_32bytes": 0, "m_op_general": 0}


def _stack_eval_one(lv, nv, filt, sb, c):
    # stack.rs:173-282
    num_pops, pushes, disable = sb
    if num_pops > 0:
        for i in range(1, num_pops):
            ch = lv.mem_channels[i]
            c.constraint(filt * (ch.used - 1))
            c.constraint(filt * (ch.is_read - 1))
            c.constraint(filt * (ch.addr_context - lv.context))
            c.constraint(filt * (ch.addr_segment - SEG_STACK))
            c.constraint(filt * (ch.addr_virtual - (lv.stack_len - (i + 1))))
        c.constraint(filt * lv.partial_channel.used)
        if not pushes:
            len_diff = lv.stack_len - num_pops
            nf = len_diff * filt
            ch = nv.mem_channels[0]
            c.constraint_transition(nf * (ch.used - 1))
            c.constraint_transition(nf * (ch.is_read - 1))
            c.constraint_transition(nf * (ch.addr_context - nv.context))
            c.constraint_transition(nf * (ch.addr_segment - SEG_STACK))
            c.constraint_transition(nf * (ch.addr_virtual - (nv.stack_len - 1)))
            c.constraint(filt * (len_diff * lv.stack_inv - lv.stack_inv_aux))
            c.constraint_transition(filt * (lv.stack_inv_aux - 1) * ch.used)
    elif pushes:
        nf = lv.stack_len * filt
        ch = lv.partial_channel
        c.constraint(nf * (ch.used - 1))
        c.constraint(nf * ch.is_read)
        c.constraint(nf * (ch.addr_context - lv.context))
        c.constraint(nf * (ch.addr_segment - SEG_STACK))
        c.constraint(nf * (ch.addr_virtual - (lv.stack_len - 1)))
        c.constraint(filt * (lv.stack_len * lv.stack_inv - lv.stack_inv_aux))
        c.constraint(filt * (lv.stack_inv_aux - 1) * ch.used)
    else:
        c.constraint(filt * nv.mem_channels[0].used)
        for a, b in zip(lv.mem_channels[0].value, nv.mem_channels[0].value):
            c.constraint(filt * (a - b))
        c.constraint(filt * lv.partial_channel.used)
    if disable:
        for i in range(max(1, num_pops), 3 - (1 if pushes else 0)):
            c.constraint(filt * lv.mem_channels[i].used)
    c.constraint_transition(filt * (nv.stack_len - (lv.stack_len - num_pops + (1 if pushes else 0))))


def make_eval_cpu(halt_pc, start_pc, syscall_jumptable, exception_jumptable, cdk_erigon=False):
    """cdk_erigon=True: the 86-column variant (`poseidon` flag): contextops.rs:26-27, control_flow.rs:11-23,
    decode.rs:10,41-42, gas.rs:30-31, jumps.rs:124-150 (no JUMPDEST-bit read), stack.rs:106-119,353-369."""
    C_OPS = C_OPS_ERIGON if cdk_erigon else globals()["C_OPS"]
    _SB = dict(globals()["_SB"], poseidon=None)
    _GAS = dict(globals()["_GAS"], **({"poseidon": 0} if cdk_erigon else {}))

    def eval_cpu(lv_raw, nv_raw, c):
        lv, nv = _CpuRow(lv_raw, cdk_erigon), _CpuRow(nv_raw, cdk_erigon)
        b = lv.opcode_bits
        # byte_unpacking.rs
        filt = lv.op["m_op_32bytes"] * (b[5] - 1)
        new_addr, written = nv.mem_channels[0].value, lv.mem_channels[0].value
        length = sum(b[i] * (1 << i) for i in range(5)) + 1
        c.constraint(filt * (new_addr[0] - written[0] - length))
        c.constraint(filt * (new_addr[1] - written[1]))
        c.constraint(filt * (new_addr[2] - written[2]))
        for limb in new_addr[3:]:
            c.constraint(filt * limb)
        # clock.rs
        c.constraint_first_row(lv.clock - 1)
        c.constraint_transition(nv.clock - lv.clock - 1)
        # contextops.rs: keep
        for name in C_OPS:
            if name != "context_op":
                c.constraint_transition(lv.op[name] * (nv.context - lv.context))
        is_get = lv.op["context_op"] * (b[0] - 1)
        c.constraint_transition(is_get * (nv.context - lv.context))
        # get
        filt = lv.op["context_op"] * (1 - b[0])
        nst = nv.mem_channels[0].value
        c.constraint(filt * (nst[2] - lv.context))
        for i, limb in enumerate(nst):
            if i != 2:
                c.constraint(filt * limb)
        c.constraint(filt * lv.pruning_flag)
        c.constraint(filt * (nv.stack_len - (lv.stack_len + 1)))
        c.constraint(filt * lv.mem_channels[1].used)
        c.constraint(filt * nv.mem_channels[0].used)
        # set
        filt = lv.op["context_op"] * b[0]
        st = lv.mem_channels[0].value
        c.constraint(filt * (st[2] - nv.context))
        for i, limb in enumerate(st[1:]):
            if i != 1:
                c.constraint(filt * limb)
        c.constraint(lv.op["context_op"] * lv.pruning_flag * (lv.pruning_flag - 1))
        c.constraint(filt * (lv.pruning_flag - st[0]))
        ntc = nv.mem_channels[0]
        c.constraint(lv.op["context_op"] * (lv.stack_inv_aux * b[0] - lv.stack_inv_aux_2))
        for ln, lr in zip(ntc.value, lv.mem_channels[2].value):
            c.constraint(lv.op["context_op"] * lv.stack_inv_aux_2 * (ln - lr))
        c.constraint(filt * lv.mem_channels[1].used)
        c.constraint(filt * ntc.used)
        # contextops eval_packed tail
        filt = lv.op["context_op"]
        ch = lv.mem_channels[2]
        stack_len = nv.stack_len - (1 - b[0])
        c.constraint(filt * (stack_len * lv.stack_inv - lv.stack_inv_aux))
        c.constraint(filt * (lv.stack_inv_aux - ch.used))
        nf = filt * lv.stack_inv_aux
        c.constraint(nf * (ch.is_read - b[0]))
        c.constraint(nf * (ch.addr_context - nv.context))
        c.constraint(nf * (ch.addr_segment - SEG_STACK))
        c.constraint(nf * (ch.addr_virtual - (stack_len - 1)))
        # control_flow.rs
        is_cpu = sum(lv.ops)
        is_cpu_next = sum(nv.ops)
        next_halt = 1 - is_cpu_next
        c.constraint_transition(is_cpu * (is_cpu_next + next_halt - 1))
        native = sum(lv.op[k] for k in ("binary_op", "ternary_op", "fp254_op", "eq_iszero", "logic_op", "not_pop",
                                        "shift", "jumpdest_keccak_general") + (("poseidon",) if cdk_erigon else ()) +
                     ("pc_push0", "dup_swap", "context_op", "m_op_general"))
        c.constraint_transition(native * (lv.program_counter - nv.program_counter + 1))
        c.constraint_transition(native * (lv.is_kernel_mode - nv.is_kernel_mode))
        is_pi = lv.op["push_prover_input"] * b[7]
        c.constraint_transition(is_pi * (lv.program_counter - nv.program_counter + 1))
        c.constraint_transition(is_pi * (lv.is_kernel_mode - nv.is_kernel_mode))
        c.constraint(lv.op["push_prover_input"] * ((lv.is_kernel_mode + lv.is_not_kernel) - 1))
        last_noncpu = (is_cpu - 1) * is_cpu_next
        c.constraint_transition(last_noncpu * (nv.program_counter - start_pc))
        c.constraint_transition(last_noncpu * (nv.is_kernel_mode - 1))
        c.constraint_transition(last_noncpu * nv.stack_len)
        # decode.rs
        km = lv.is_kernel_mode
        c.constraint(km * (km - 1))
        for bit in b:
            c.constraint(bit * (bit - 1))
        OPCODES = [(0x14, 1, False, "eq_iszero")] + ([(0x22, 1, True, "poseidon")] if cdk_erigon else []) + \
                  [(0x56, 1, False, "jumps"), (0x80, 5, False, "dup_swap"),
                   (0xf6, 1, True, "context_op"), (0xf9, 0, True, "exit_kernel")]
        COMBINED = ["logic_op", "fp254_op", "binary_op", "ternary_op", "shift", "m_op_general",
                    "jumpdest_keccak_general", "not_pop", "pc_push0", "m_op_32bytes", "push_prover_input"]
        for _, _, _, name in OPCODES:
            c.constraint(lv.op[name] * (lv.op[name] - 1))
        for name in COMBINED:
            c.constraint(lv.op[name] * (lv.op[name] - 1))
        flag_sum = sum(lv.op[n] for _, _, _, n in OPCODES) + sum(lv.op[n] for n in COMBINED)
        c.constraint(flag_sum * (flag_sum - 1))
        for oc, block_len, kernel_only, name in OPCODES:
            unavailable = (1 - km) if kernel_only else 0
            mismatch = 0
            for i in range(7, block_len - 1, -1):          # .rev().take(8 - block_length)
                mismatch = mismatch + ((1 - b[i]) if (oc >> i) & 1 else b[i])
            c.constraint(lv.op[name] * (unavailable + mismatch))

        def high_bits(k):
            return sum(b[i] * (1 << i) for i in range(7, 7 - k, -1))
        c.constraint((km - 1) * lv.op["fp254_op"])
        c.constraint(lv.op["ternary_op"] * b[1] * (km - 1))
        opcode = high_bits(8)
        c.constraint((km - 1) * lv.op["m_op_general"])
        c.constraint((opcode - 0xfb) * (opcode - 0xfc) * lv.op["m_op_general"])
        c.constraint((km - 1) * lv.op["jumpdest_keccak_general"] * (1 - b[1]))
        c.constraint((opcode - 0x21) * (opcode - 0x5b) * lv.op["jumpdest_keccak_general"])
        c.constraint((opcode - 0x58) * (opcode - 0x5f) * lv.op["pc_push0"])
        c.constraint((opcode - 0x19) * (opcode - 0x50) * lv.op["not_pop"])
        c.constraint((km - 1) * lv.op["m_op_32bytes"])
        high3 = high_bits(3)
        c.constraint((high3 - 0xc0) * (opcode - 0xf8) * lv.op["m_op_32bytes"])
        c.constraint((opcode - 0xee) * (high3 - 0x60) * lv.op["push_prover_input"])
        c.constraint(lv.op["push_prover_input"] * b[7] * (km - 1))
        # dup_swap.rs
        n = b[0] + b[1] * 2 + b[2] * 4 + b[3] * 8

        def chan_eq(f, a, bb):
            for x, y in zip(a.value, bb.value):
                c.constraint(f * (x - y))

        def constrain_chan(is_read, f, offset, ch):
            c.constraint(f * (ch.used - 1))
            c.constraint(f * (ch.is_read - (1 if is_read else 0)))
            c.constraint(f * (ch.addr_context - lv.context))
            c.constraint(f * (ch.addr_segment - SEG_STACK))
            c.constraint(f * (ch.addr_virtual - (lv.stack_len - 1 - offset)))
        f = lv.op["dup_swap"] * (1 - b[4])
        chan_eq(f, lv.mem_channels[1], lv.mem_channels[0])
        constrain_chan(False, f, 0, lv.mem_channels[1])
        chan_eq(f, lv.mem_channels[2], nv.mem_channels[0])
        constrain_chan(True, f, n, lv.mem_channels[2])
        c.constraint_transition(f * (nv.stack_len - lv.stack_len - 1))
        c.constraint(f * nv.mem_channels[0].used)
        f = lv.op["dup_swap"] * b[4]
        chan_eq(f, lv.mem_channels[0], lv.mem_channels[2])
        constrain_chan(False, f, n + 1, lv.mem_channels[2])
        chan_eq(f, lv.mem_channels[1], nv.mem_channels[0])
        constrain_chan(True, f, n + 1, lv.mem_channels[1])
        c.constraint(f * (nv.stack_len - lv.stack_len))
        c.constraint(f * nv.mem_channels[0].used)
        c.constraint(lv.op["dup_swap"] * lv.partial_channel.used)
        # gas.rs
        gfilt = sum(lv.op[k] for k in C_OPS if k in _GAS)
        gas_used = sum(_GAS[k] * lv.op[k] for k in C_OPS if k in _GAS)
        c.constraint_transition(gfilt * (nv.gas - (lv.gas + gas_used)))
        gas_diff = nv.gas - lv.gas
        for k in C_OPS:
            if k in _GAS:
                c.constraint_transition(lv.op[k] * (gas_diff - _GAS[k]))
        c.constraint_transition(lv.op["jumps"] * (gas_diff - (8 + b[0] * 2)))
        cost_filter = b[0] + b[4] - b[0] * b[4]
        c.constraint_transition(lv.op["binary_op"] * (gas_diff - (5 + cost_filter * (3 - 5))))
        c.constraint_transition(lv.op["ternary_op"] * (gas_diff - (8 - b[1] * 8)))
        c.constraint_transition(lv.op["not_pop"] * (gas_diff - ((1 - b[0]) * 2 + b[0] * 3)))
        c.constraint_transition(lv.op["jumpdest_keccak_general"] * (gas_diff - (b[1] * 1 + (1 - b[1]) * 0)))
        c.constraint_transition(lv.op["push_prover_input"] * (gas_diff - ((1 - b[7]) * 3 + b[7] * 0)))
        c.constraint_transition((is_cpu - 1) * is_cpu_next * nv.gas)
        # halt.rs
        halt_state = 1 - is_cpu
        c.constraint(halt_state * (halt_state - 1))
        c.constraint_transition(halt_state * (next_halt - 1))
        c.constraint(halt_state * (lv.is_kernel_mode - 1))
        for i in range(3):
            c.constraint(halt_state * lv.mem_channels[i].used)
        c.constraint_last_row(halt_state - 1)
        c.constraint(halt_state * (lv.program_counter - halt_pc))
        # jumps.rs: exit_kernel
        inp = lv.mem_channels[0].value
        f = lv.op["exit_kernel"]
        c.constraint_transition(f * (inp[0] - nv.program_counter))
        c.constraint_transition(f * (inp[1] - nv.is_kernel_mode))
        c.constraint_transition(f * (inp[6] - nv.gas))
        c.constraint(f * inp[7])
        # jump / jumpi
        dst, cond = lv.mem_channels[0].value, lv.mem_channels[1].value
        f = lv.op["jumps"]
        is_jump, is_jumpi = f * (1 - b[0]), f * b[0]
        len_diff = lv.stack_len - 1 - b[0]
        nf = len_diff * f
        ch = nv.mem_channels[0]
        c.constraint_transition(nf * (ch.used - 1))
        c.constraint_transition(nf * (ch.is_read - 1))
        c.constraint_transition(nf * (ch.addr_context - nv.context))
        c.constraint_transition(nf * (ch.addr_segment - SEG_STACK))
        c.constraint_transition(nf * (ch.addr_virtual - (nv.stack_len - 1)))
        c.constraint(f * (len_diff * lv.stack_inv - lv.stack_inv_aux))
        c.constraint_transition(f * (lv.stack_inv_aux - 1) * ch.used)
        c.constraint(is_jump * (cond[0] - 1))
        for limb in cond[1:]:
            c.constraint(is_jump * limb)
        sj = lv.should_jump
        c.constraint(f * sj * (sj - 1))
        cond_sum = sum(cond)
        c.constraint(f * (sj - 1) * cond_sum)
        c.constraint(f * (lv.cond_sum_pinv * cond_sum - sj))
        c.constraint(f * sj * sum(dst[1:]))
        if not cdk_erigon:                                   # "We skip jump destinations verification with cdk_erigon"
            jd = lv.mem_channels[2]
            c.constraint(f * (jd.value[0] - 1))
            c.constraint(f * (jd.used - sj * (1 - lv.is_kernel_mode)))
            c.constraint(f * (jd.is_read - 1))
            c.constraint(f * (jd.addr_context - lv.context))
            c.constraint(f * (jd.addr_segment - SEG_JUMPDEST_BITS))
            c.constraint(f * (jd.addr_virtual - dst[0]))
        c.constraint(f * lv.partial_channel.used)
        c.constraint(is_jump * lv.mem_channels[1].used)
        c.constraint_transition(is_jump * (nv.stack_len - lv.stack_len + 1))
        c.constraint_transition(is_jumpi * (nv.stack_len - lv.stack_len + 2))
        c.constraint_transition(f * (sj - 1) * (nv.program_counter - (lv.program_counter + 1)))
        c.constraint_transition(f * sj * (nv.program_counter - dst[0]))
        # membus.rs
        c.constraint(lv.code_context - (1 - lv.is_kernel_mode) * lv.context)
        for ch in lv.mem_channels:
            c.constraint(ch.used * (ch.used - 1))
        c.constraint(lv.partial_channel.used * (lv.partial_channel.used - 1))
        # memio.rs: load
        f = lv.op["m_op_general"] * b[0]
        a = lv.mem_channels[0].value
        lc = lv.mem_channels[1]
        c.constraint(f * (lc.used - 1))
        c.constraint(f * (lc.is_read - 1))
        c.constraint(f * (lc.addr_context - a[2]))
        c.constraint(f * (lc.addr_segment - a[1]))
        c.constraint(f * (lc.addr_virtual - a[0]))
        for x, y in zip(lc.value, nv.mem_channels[0].value):
            c.constraint(f * (x - y))
        c.constraint(f * lv.mem_channels[2].used)
        c.constraint(f * lv.partial_channel.used)
        _stack_eval_one(lv, nv, f, (1, True, False), c)
        # store
        f = lv.op["m_op_general"] * (b[0] - 1)
        a = lv.mem_channels[1].value
        sc = lv.partial_channel
        c.constraint(f * (sc.used - 1))
        c.constraint(f * sc.is_read)
        c.constraint(f * (sc.addr_context - a[2]))
        c.constraint(f * (sc.addr_segment - a[1]))
        c.constraint(f * (sc.addr_virtual - a[0]))
        c.constraint(f * lv.mem_channels[2].used)
        ch = lv.mem_channels[1]
        c.constraint(f * (ch.used - 1))
        c.constraint(f * (ch.is_read - 1))
        c.constraint(f * (ch.addr_context - lv.context))
        c.constraint(f * (ch.addr_segment - SEG_STACK))
        c.constraint(f * (ch.addr_virtual - (lv.stack_len - 2)))
        len_diff = lv.stack_len - 2
        mg = lv.op["m_op_general"]
        c.constraint(mg * (len_diff * lv.stack_inv - lv.stack_inv_aux))
        trc = nv.mem_channels[0]
        is_top_read = lv.stack_inv_aux * (1 - b[0])
        c.constraint(mg * (lv.stack_inv_aux_2 - is_top_read))
        nf = mg * lv.stack_inv_aux_2
        c.constraint_transition(nf * (trc.used - 1))
        c.constraint_transition(nf * (trc.is_read - 1))
        c.constraint_transition(nf * (trc.addr_context - nv.context))
        c.constraint_transition(nf * (trc.addr_segment - SEG_STACK))
        c.constraint_transition(nf * (trc.addr_virtual - (nv.stack_len - 1)))
        c.constraint(mg * (lv.stack_inv_aux - 1) * trc.used)
        c.constraint(mg * b[0] * trc.used)
        # modfp254.rs
        P_LIMBS = [0xd87cfd47, 0x3c208c16, 0x6871ca8d, 0x97816a91, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72]
        for limb, pl in zip(lv.mem_channels[2].value, P_LIMBS):
            c.constraint(lv.op["fp254_op"] * (limb - pl))
        # pc.rs
        f = lv.op["pc_push0"] * (1 - b[0])
        nst = nv.mem_channels[0].value
        c.constraint(f * (nst[0] - lv.program_counter))
        for limb in nst[1:]:
            c.constraint(f * limb)
        # push0.rs
        f = lv.op["pc_push0"] * b[0]
        for limb in nv.mem_channels[0].value:
            c.constraint(f * limb)
        # shift.rs
        sh = lv.op["shift"]
        disp, two_exp = lv.mem_channels[0], lv.mem_channels[2]
        hz = two_exp.used
        c.constraint(sh * hz * (two_exp.is_read - 1))
        hsum = sum(disp.value[1:])
        c.constraint(sh * (hsum * lv.high_limb_sum_inv - (1 - hz)))
        c.constraint(sh * hsum * hz)
        c.constraint(sh * two_exp.addr_context)
        c.constraint(sh * (two_exp.addr_segment - SEG_SHIFT_TABLE))
        c.constraint(sh * (two_exp.addr_virtual - disp.value[0]))
        # simple_logic: not.rs
        f = lv.op["not_pop"] * b[0]
        for x, y in zip(lv.mem_channels[0].value, nv.mem_channels[0].value):
            c.constraint(f * (y + x - 0xFFFFFFFF))
        _stack_eval_one(lv, nv, f, (1, True, True), c)
        # eq_iszero.rs
        in0, in1, out = lv.mem_channels[0].value, lv.mem_channels[1].value, nv.mem_channels[0].value
        eqf = lv.op["eq_iszero"] * (1 - b[0])
        izf = lv.op["eq_iszero"] * b[0]
        ef = lv.op["eq_iszero"]
        equal = out[0]
        unequal = 1 - equal
        c.constraint(ef * equal * unequal)
        for limb in out[1:]:
            c.constraint(ef * limb)
        for limb in in1:
            c.constraint(izf * limb)
        for x, y in zip(in0, in1):
            c.constraint(ef * equal * (x - y))
        dot = sum((x - y) * d for x, y, d in zip(in0, in1, lv.diff_pinv))
        c.constraint(ef * (dot - unequal))
        _stack_eval_one(lv, nv, eqf, (2, True, True), c)
        _stack_eval_one(lv, nv, izf, (1, True, True), c)
        # stack.rs eval_packed
        for name in C_OPS:
            if _SB[name] is not None:
                _stack_eval_one(lv, nv, lv.op[name], _SB[name], c)
            if name in _MIGHT_OVERFLOW:
                diff = nv.stack_len - 1025
                c.constraint_transition(lv.op[name] * (diff * lv.stack_len_bounds_aux - (1 - nv.is_kernel_mode)))
        _stack_eval_one(lv, nv, lv.op["jumpdest_keccak_general"] * b[1], (0, False, True), c)
        _stack_eval_one(lv, nv, lv.op["jumpdest_keccak_general"] * (1 - b[1]), (2, True, True), c)
        if cdk_erigon:                                           # POSEIDON (3 pops) / POSEIDON_GENERAL (2 pops)
            _stack_eval_one(lv, nv, lv.op["poseidon"] * (1 - b[0]), (3, True, True), c)
            _stack_eval_one(lv, nv, lv.op["poseidon"] * b[0], (2, True, True), c)
        npop = lv.op["not_pop"]
        c.constraint(npop * ((lv.stack_len - 1) * lv.stack_inv - lv.stack_inv_aux))
        trc = nv.mem_channels[0]
        is_top_read = lv.stack_inv_aux * (1 - b[0])
        c.constraint(npop * (lv.stack_inv_aux_2 - is_top_read))
        nf = npop * lv.stack_inv_aux_2
        c.constraint_transition(nf * (trc.used - 1))
        c.constraint_transition(nf * (trc.is_read - 1))
        c.constraint_transition(nf * (trc.addr_context - nv.context))
        c.constraint_transition(nf * (trc.addr_segment - SEG_STACK))
        c.constraint_transition(nf * (trc.addr_virtual - (nv.stack_len - 1)))
        c.constraint(npop * (lv.stack_inv_aux_2 - 1) * trc.used)
        for ch in lv.mem_channels[1:]:
            c.constraint(npop * (b[0] - 1) * ch.used)
        c.constraint(npop * (b[0] - 1) * lv.partial_channel.used)
        c.constraint_transition(npop * (b[0] - 1) * (nv.stack_len - lv.stack_len + 1))
        # syscalls_exceptions.rs
        fs, fe = lv.op["syscall"], lv.op["exception"]
        tf = fs + fe
        c.constraint(fs * (fs - 1))
        c.constraint(fe * (fe - 1))
        ecb = lv.exc_code_bits
        exc_code = sum(bit * (1 << i) for i, bit in enumerate(ecb))
        c.constraint(fe * (exc_code - 6) * lv.is_kernel_mode)
        for bit in ecb:
            c.constraint(fe * bit * (bit - 1))
        opc = sum(bit * (1 << i) for i, bit in enumerate(b))
        op_handler = syscall_jumptable + opc * 3
        exc_handler = exception_jumptable + exc_code * 3
        jc = lv.mem_channels[1]
        c.constraint(tf * jc.used)
        c.constraint(tf * (jc.is_read - 1))
        c.constraint(tf * jc.addr_context)
        c.constraint(tf * (jc.addr_segment - SEG_CODE_))
        c.constraint(fs * (jc.addr_virtual - op_handler))
        c.constraint(fe * (jc.addr_virtual - exc_handler))
        for limb in jc.value[1:]:
            c.constraint(tf * limb)
        c.constraint(tf * lv.mem_channels[2].used)
        c.constraint_transition(tf * (nv.program_counter - jc.value[0]))
        c.constraint_transition(tf * (nv.is_kernel_mode - 1))
        c.constraint_transition(tf * nv.gas)
        out = nv.mem_channels[0].value
        c.constraint(fs * (out[0] - (lv.program_counter + 1)))
        c.constraint(fe * (out[0] - lv.program_counter))
        c.constraint(fs * (out[1] - lv.is_kernel_mode))
        c.constraint(tf * (out[6] - lv.gas))
        c.constraint(tf * out[7])
        c.constraint(fe * (exc_code - 6) * out[1])
        for limb in out[2:6]:
            c.constraint(tf * limb)
    return eval_cpu


CPU_TEST_CONSTS = (31337, 4242, 777777, 888888)
AIRS.update({8: (make_eval_cpu(*CPU_TEST_CONSTS), 85), 10: (make_eval_cpu(*CPU_TEST_CONSTS, cdk_erigon=True), 86)})


# ---- cdk_erigon Poseidon table (oracle/poseidon_table.py) ----
from .poseidon_table import eval_poseidon  # noqa: E402
AIRS.update({9: (eval_poseidon, 322)})
