"""oracle/mem_trace.py -- TEST INFRASTRUCTURE ONLY.
Restatement of the Memory table's witness generator, evm_arithmetization/src/memory/memory_stark.rs:104-455
(`MemoryOp::into_row`, `generate_first_change_flags_and_rc`, `generate_trace_row_major`, `fill_gaps`, `pad_memory_ops`,
`insert_stale_contexts`, `generate_trace_col_major`, `generate_trace`), columns memory/columns.rs:13-94.
A MemoryOp is a dict(filter, timestamp, ctx, seg, virt, is_read, value).  Checked against the restated AIR
(tests/test_oracle_tracegen.py) and, through a GPU proof, by the oracle verifier."""
import numpy as np

P = 0xFFFFFFFF00000001
SEG_CODE, SEG_TRIE_DATA, SEG_ACCOUNTS_LL, SEG_STORAGE_LL = 0, 12, 34, 35        # memory/segments.rs (unscaled)
PREINITIALIZED = (SEG_CODE, SEG_TRIE_DATA, SEG_ACCOUNTS_LL, SEG_STORAGE_LL)
(FILTER, TIMESTAMP, TIMESTAMP_INV, IS_READ, CTX, SEG, VIRT) = range(7)
VALUE = 7
(CTX_FIRST, SEG_FIRST, VIRT_FIRST, INIT_AUX, PREINIT, PREINIT_AUX, STALE_CONTEXTS, IS_PRUNED, STALE_FREQ, IS_STALE,
 MAYBE_AFTER, AFTER_FILTER, RANGE_CHECK, COUNTER, FREQUENCIES) = range(15, 30)


def _key(op):
    return (op["ctx"], op["seg"], op["virt"], op["timestamp"])


def _dummy_read(ctx, seg, virt, ts, value):
    return dict(filter=False, timestamp=ts, ctx=ctx, seg=seg, virt=virt, is_read=True, value=value)


def fill_gaps(ops):                                                     # memory_stark.rs:296-355
    if ops[0]["virt"] != 0:
        ops.insert(0, _dummy_read(0, 0, 0, 1, 0))
    max_rc = (1 << max(len(ops) - 1, 0).bit_length()) - 1
    snapshot = [dict(o) for o in ops]
    for a, b in zip(snapshot, snapshot[1:]):
        curr, nxt = dict(a), dict(b)
        if curr["ctx"] != nxt["ctx"] or curr["seg"] != nxt["seg"]:
            while nxt["virt"] > max_rc:
                d = _dummy_read(nxt["ctx"], nxt["seg"], nxt["virt"] - max_rc, curr["timestamp"] + 1, 0)
                ops.append(d)
                nxt = d
        elif curr["virt"] != nxt["virt"]:
            while nxt["virt"] - curr["virt"] - 1 > max_rc:
                d = _dummy_read(curr["ctx"], curr["seg"], curr["virt"] + max_rc + 1, curr["timestamp"] + 1, 0)
                ops.append(d)
                curr = d
        else:
            while nxt["timestamp"] - curr["timestamp"] > max_rc:
                d = _dummy_read(curr["ctx"], curr["seg"], curr["virt"], curr["timestamp"] + max_rc, curr["value"])
                ops.append(d)
                curr = d


def pad_memory_ops(ops):                                                # memory_stark.rs:357-383
    last = ops[-1]
    pad = dict(filter=False, timestamp=last["timestamp"] + 1, ctx=last["ctx"], seg=last["seg"], virt=last["virt"] + 1,
               is_read=True, value=0)
    n = len(ops)
    target = 1 << n.bit_length()                                        # (n + 1).next_power_of_two()
    ops += [dict(pad) for _ in range(target - n)]


def generate_trace(memory_ops, mem_before_values=(), stale_contexts=()):
    """-> ((30, n) uint64 column-major, mem_after rows)."""
    ops = [dict(o) for o in memory_ops]
    for (ctx, seg, virt), value in mem_before_values:
        ops.append(dict(filter=True, timestamp=0, ctx=ctx, seg=seg, virt=virt, is_read=False, value=value))
    ops.sort(key=_key)
    fill_gaps(ops)
    ops.sort(key=_key)
    pad_memory_ops(ops)
    ops.sort(key=_key)
    n = len(ops)
    t = np.zeros((30, n), dtype=np.uint64)
    for i, o in enumerate(ops):                                         # MemoryOp::into_row
        t[FILTER, i] = 1 if o["filter"] else 0
        t[TIMESTAMP, i] = o["timestamp"]
        t[TIMESTAMP_INV, i] = pow(o["timestamp"], P - 2, P) if o["timestamp"] else 0
        t[IS_READ, i] = 1 if o["is_read"] else 0
        t[CTX, i], t[SEG, i], t[VIRT, i] = o["ctx"], o["seg"], o["virt"]
        for j in range(8):
            t[VALUE + j, i] = (o["value"] >> (32 * j)) & 0xFFFFFFFF
    for i in range(n):                                                  # generate_first_change_flags_and_rc
        j = 0 if i == n - 1 else i + 1
        c, s, v, ts = (int(t[k, i]) for k in (CTX, SEG, VIRT, TIMESTAMP))
        nc, ns, nv, nts = (int(t[k, j]) for k in (CTX, SEG, VIRT, TIMESTAMP))
        cf = c != nc
        sf = s != ns and not cf
        vf = v != nv and not sf and not cf
        t[CTX_FIRST, i], t[SEG_FIRST, i], t[VIRT_FIRST, i] = int(cf), int(sf), int(vf)
        if i == n - 1:
            rc = 0
        elif cf:
            rc = nc - c - 1
        elif sf:
            rc = ns - s - 1
        elif vf:
            rc = nv - v - 1
        else:
            rc = nts - ts
        rc %= P
        assert rc < n, "Range check too large. Bug in fill_gaps?"
        t[RANGE_CHECK, i] = rc
        aux = ((ns - SEG_ACCOUNTS_LL) * (ns - SEG_STORAGE_LL)) % P
        pre = ((ns - SEG_CODE) * (ns - SEG_TRIE_DATA) * aux) % P
        t[PREINIT_AUX, i], t[PREINIT, i] = aux, pre
        t[INIT_AUX, i] = (pre * (int(cf) + int(sf) + int(vf)) * int(t[IS_READ, j])) % P
    for ctx in stale_contexts:                                          # insert_stale_contexts
        t[STALE_CONTEXTS, ctx] = ctx + 1
        t[IS_PRUNED, ctx] = 1
    t[COUNTER] = np.arange(n, dtype=np.uint64)                          # generate_trace_col_major
    for i in range(n):
        t[FREQUENCIES, int(t[RANGE_CHECK, i])] += 1
        if t[CTX_FIRST, i] == 1 or t[SEG_FIRST, i] == 1:
            t[FREQUENCIES, int(t[VIRT, i + 1]) if i < n - 1 else 0] += 1
        ctx = int(t[CTX, i])
        if ctx + 1 == int(t[STALE_CONTEXTS, ctx]):
            t[IS_STALE, i] = 1
            t[STALE_FREQ, ctx] += 1
        elif t[FILTER, i] == 1 and (t[CTX_FIRST, i] == 1 or t[SEG_FIRST, i] == 1 or t[VIRT_FIRST, i] == 1):
            t[MAYBE_AFTER, i] = 1
            if any(t[VALUE + j, i] != 0 for j in range(8)) or int(t[SEG, i]) in PREINITIALIZED:
                t[AFTER_FILTER, i] = 1
    mem_after = [[1] + [int(t[k, i]) for k in range(CTX, CTX_FIRST)] for i in range(n) if t[AFTER_FILTER, i] == 1]
    return t, mem_after
