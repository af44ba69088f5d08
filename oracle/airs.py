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
