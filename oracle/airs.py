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
