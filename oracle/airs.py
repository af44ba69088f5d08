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
