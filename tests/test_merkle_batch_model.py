"""CPU: the digest offsets of `merkle_levels_batched` (zk_evm_amd/csrc/merkle_host.inc, r05: the small levels of several trees in one
launch per level) against the walk `merkle_levels` does one tree at a time (child += 2 * count digests per level), and the set of
levels each of the two builds when a tree is split between them at `top` -- every level exactly once, whatever the tree's height."""


def sequential(log_leaves, cap_height):
    """(level l -> (child digest offset, parent digest offset)) as merkle_levels walks them"""
    out, child = {}, 0
    for l in range(log_leaves - 1, cap_height - 1, -1):
        cnt = 1 << l
        parent = child + 2 * cnt
        out[l] = (child, parent)
        child = parent
    return out


def batched_offsets(log_leaves, l):
    total = 2 << log_leaves
    return total - (4 << l), total - (2 << l)


def test_offsets_and_level_split():
    for cap in (0, 2, 4):
        for L in range(cap, 24):
            seq = sequential(L, cap)
            for l, (c, p) in seq.items():
                assert batched_offsets(L, l) == (c, p), (L, l)
            for top in (4, 9, 14, 17, 30):
                lane = [l for l in range(L - 1, cap - 1, -1) if not l <= top]          # merkle_levels(stop_at = top)
                start = min(L - 1, top) if L > 0 else -1
                batch = [l for l in range(start, cap - 1, -1)] if L > cap else []        # merkle_levels_batched, this tree
                batch = [l for l in batch if L > l]
                assert sorted(lane + batch) == sorted(seq), (cap, L, top)
                assert not set(lane) & set(batch)
    # the cap = the last 2^cap_height digests of the array
    for cap, L in ((4, 21), (4, 4), (0, 7)):
        n_digests = sum(1 << l for l in range(cap, L + 1))
        last_parent = sequential(L, cap)[cap][1] if L > cap else 0
        assert last_parent == n_digests - (1 << cap)
