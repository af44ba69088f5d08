"""CPU model of the lane-swap strided NTT pass (zk_evm_amd/csrc/ntt.cuh ntt_strided_swap_kernel): the kernel's register /
lane / wave bookkeeping -- which row of the 2^9-row tile every register of every lane holds after each v_permlane16/32_swap,
which twiddle entry each butterfly reads, the LDS exchange -- restated expression by expression in Python and run against the
plain definition of the pass (the stages of ntt_step, one butterfly at a time).  The twiddle TABLES hold random field elements:
what is checked is the index algebra, and a wrong index cannot cancel.  The arithmetic itself is the GPU tests' business
(tests/test_gpu_commit.py pins the kernel to ntt_pass_kernel bit for bit)."""
import random

P = 0xFFFFFFFF00000001
R, LOG_T = 9, 4


def bfly(a, b, w):
    t = b * w % P
    return (a + t) % P, (a - t) % P


def reference_pass(data, log_n, log_d, tw, dit):
    out = list(data)
    ks = range(R) if dit else range(R - 1, -1, -1)
    for k in ks:
        log_D = log_d + k
        D = 1 << log_D
        for x in range(1 << log_n):
            if x & D:
                continue
            if dit:
                w = tw[D - 1 + (x & (D - 1))]
            else:
                w = tw[(1 << (log_n - 1 - log_D)) - 1 + (x >> (log_D + 1))]
            out[x], out[x + D] = bfly(out[x], out[x + D], w)
    return out


def lane_swap(v, lanebit, regbit):
    """(register a, lane bit = 1) <-> (register b, lane bit = 0) for every pair a = m, b = m | 1 << regbit; v[lane][m]"""
    for lane in range(64):
        if (lane >> lanebit) & 1:
            continue
        partner = lane | (1 << lanebit)
        for m in range(16):
            if m & (1 << regbit):
                continue
            a, b = m, m | (1 << regbit)
            v[partner][a], v[lane][b] = v[lane][b], v[partner][a]


def model_tile(src, dst, log_n, log_d, tw, dit, tile_id, last_pass=False):
    log_lo_tiles = log_d - LOG_T
    hi_idx, lo_tile = tile_id >> log_lo_tiles, tile_id & ((1 << log_lo_tiles) - 1)
    base = (hi_idx << (log_d + R)) + (lo_tile << LOG_T)
    V = [[[None] * 16 for _ in range(64)] for _ in range(8)]           # V[wave][lane][m]
    lds = {}

    def lanes():
        for wv in range(8):
            for lane in range(64):
                yield wv, lane, lane & 15, (lane >> 4) & 1, lane >> 5

    def tw_load(lane_byte_off, uniform_index):
        assert lane_byte_off % 8 == 0
        return tw[uniform_index + lane_byte_off // 8]

    if not dit:
        for wv, lane, u, l4, l5 in lanes():
            s = base + u + (((l5 << 8) | (l4 << 7) | wv) << log_d)
            for m in range(16):
                V[wv][lane][m] = src[s + (m << (3 + log_d))]
        s8 = log_n - 1 - log_d - 8
        lvl = lambda k: ((1 << (s8 + 8 - k)) - 1) + (hi_idx << (8 - k))
        for wv in range(8):
            v = V[wv]
            lane_swap(v, 5, 3)
            for lane in range(64):
                for m in range(8):
                    v[lane][m], v[lane][m | 8] = bfly(v[lane][m], v[lane][m | 8], tw[lvl(8)])
            lane_swap(v, 4, 2)
            for lane in range(64):
                for m in range(16):
                    if not m & 4:
                        v[lane][m], v[lane][m | 4] = bfly(v[lane][m], v[lane][m | 4], tw[lvl(7) + (m >> 3)])
            lane_swap(v, 5, 1)
            for lane in range(64):
                for m in range(16):
                    if not m & 2:
                        v[lane][m], v[lane][m | 2] = bfly(v[lane][m], v[lane][m | 2], tw[lvl(6) + (m >> 2)])
            lane_swap(v, 4, 0)
            for lane in range(64):
                for m in range(16):
                    if not m & 1:
                        v[lane][m], v[lane][m | 1] = bfly(v[lane][m], v[lane][m | 1], tw[lvl(5) + (m >> 1)])
            lane_swap(v, 5, 3)
            for lane in range(64):
                l5 = lane >> 5
                for m in range(8):
                    w = tw_load(l5 * 64, lvl(4) + m)
                    v[lane][m], v[lane][m | 8] = bfly(v[lane][m], v[lane][m | 8], w)
            lane_swap(v, 4, 2)
            for lane in range(64):
                l4, l5 = (lane >> 4) & 1, lane >> 5
                for m in range(16):
                    if not m & 4:
                        i = ((m >> 1) & 1) * 4 + (m & 1) * 2 + (m >> 3)
                        w = tw_load((l5 * 16 + l4 * 8) * 8, lvl(3) + i)
                        v[lane][m], v[lane][m | 4] = bfly(v[lane][m], v[lane][m | 4], w)
        for wv, lane, u, l4, l5 in lanes():
            tb = (l5 << 8) | (l4 << 7) | wv
            for m in range(16):
                t = (tb | ((m & 2) << 5) | ((m & 1) << 5) | ((m & 8) << 1) | ((m & 4) << 1)) ^ (((m >> 3) & 1) ^ l4)
                assert ((t << 4) + u) not in lds
                lds[(t << 4) + u] = V[wv][lane][m]
        bank_check = []
        for wv, lane, u, l4, l5 in lanes():
            tb = (wv << 6) | (l5 << 5) | (l4 << 4)
            par = l4 ^ ((wv >> 1) & 1)
            for m in range(16):
                V[wv][lane][m] = lds[(((tb | m) ^ par) << 4) + u]
        for wv, lane, u, l4, l5 in lanes():
            v = V[wv][lane]
            tb = (wv << 6) | (l5 << 5) | (l4 << 4)
            b = lvl(2) + wv * 8
            w = [tw_load((l5 * 4 + l4 * 2) * 8, b), tw_load((l5 * 4 + l4 * 2) * 8, b + 1)]
            for m in range(16):
                if not m & 4:
                    v[m], v[m | 4] = bfly(v[m], v[m | 4], w[m >> 3])
            b = lvl(1) + wv * 16
            w = [tw_load((l5 * 8 + l4 * 4) * 8, b + i) for i in range(4)]
            for m in range(16):
                if not m & 2:
                    v[m], v[m | 2] = bfly(v[m], v[m | 2], w[m >> 2])
            b = lvl(0) + wv * 32
            w = [tw_load((l5 * 16 + l4 * 8) * 8, b + i) for i in range(8)]
            for m in range(16):
                if not m & 1:
                    v[m], v[m | 1] = bfly(v[m], v[m | 1], w[m >> 1])
            d = base + u + (tb << log_d)
            for m in range(16):
                dst[d + (m << log_d)] = v[m]
    else:
        for wv, lane, u, l4, l5 in lanes():
            s = base + u + (((wv << 6) | (l5 << 1) | l4) << log_d)
            for m in range(16):
                V[wv][lane][m] = src[s + (m << (2 + log_d))]
        lvl = lambda k: (1 << (log_d + k)) - 1
        for wv in range(8):
            v = V[wv]
            xl8 = lambda lane: ((lo_tile << LOG_T) + (lane & 15)) * 8
            lane_swap(v, 4, 3)
            for lane in range(64):
                w = tw_load(xl8(lane), lvl(0))
                for m in range(8):
                    v[lane][m], v[lane][m | 8] = bfly(v[lane][m], v[lane][m | 8], w)
            lane_swap(v, 5, 2)
            for lane in range(64):
                w = [tw_load(xl8(lane), lvl(1)), tw_load(xl8(lane), lvl(1) + (1 << log_d))]
                for m in range(16):
                    if not m & 4:
                        v[lane][m], v[lane][m | 4] = bfly(v[lane][m], v[lane][m | 4], w[m >> 3])
            for lane in range(64):
                w = [tw_load(xl8(lane), lvl(2) + (i << log_d)) for i in range(4)]
                for m in range(16):
                    if not m & 1:
                        v[lane][m], v[lane][m | 1] = bfly(v[lane][m], v[lane][m | 1], w[((m >> 2) & 1) * 2 + (m >> 3)])
            for lane in range(64):
                w = [tw_load(xl8(lane), lvl(3) + (i << log_d)) for i in range(8)]
                for m in range(16):
                    if not m & 2:
                        v[lane][m], v[lane][m | 2] = bfly(v[lane][m], v[lane][m | 2], w[(m & 1) * 4 + ((m >> 2) & 1) * 2 + (m >> 3)])
            lane_swap(v, 5, 3)
            for lane in range(64):
                l5 = lane >> 5
                lo8 = xl8(lane) + ((l5 << log_d) << 3)
                for m in range(8):
                    w = tw_load(lo8, lvl(4) + ((((m >> 1) & 1) * 8 + (m & 1) * 4 + ((m >> 2) & 1) * 2) << log_d))
                    v[lane][m], v[lane][m | 8] = bfly(v[lane][m], v[lane][m | 8], w)
            lane_swap(v, 4, 2)
            for lane in range(64):
                l4, l5 = (lane >> 4) & 1, lane >> 5
                lo8 = xl8(lane) + (((l4 * 2 + l5) << log_d) << 3)
                w = [tw_load(lo8, lvl(5) + ((((i >> 2) & 1) * 16 + ((i >> 1) & 1) * 8 + (i & 1) * 4) << log_d)) for i in range(8)]
                for m in range(16):
                    if not m & 4:
                        v[lane][m], v[lane][m | 4] = bfly(v[lane][m], v[lane][m | 4], w[(m >> 3) * 4 + (m & 3)])
        for wv, lane, u, l4, l5 in lanes():
            tb = ((wv << 6) | (l4 << 1) | l5) ^ l4
            for m in range(16):
                t = tb | ((m & 4) << 3) | ((m & 8) << 1) | ((m & 2) << 2) | ((m & 1) << 2)
                assert ((t << 4) + u) not in lds
                lds[(t << 4) + u] = V[wv][lane][m]
        for wv, lane, u, l4, l5 in lanes():
            v = V[wv][lane]
            tb = (wv << 2) | (l5 << 1) | l4
            tp = tb ^ l5
            for m in range(16):
                v[m] = lds[((m << 9) + (tp << 4)) + u]
            lo8 = ((lo_tile << LOG_T) + u) * 8 + ((tb << log_d) << 3)
            w = [tw_load(lo8, lvl(6)), tw_load(lo8, lvl(6) + (32 << log_d))]
            for m in range(16):
                if not m & 2:
                    v[m], v[m | 2] = bfly(v[m], v[m | 2], w[m & 1])
            w = [tw_load(lo8, lvl(7) + ((i * 32) << log_d)) for i in range(4)]
            for m in range(16):
                if not m & 4:
                    v[m], v[m | 4] = bfly(v[m], v[m | 4], w[m & 3])
            w = [tw_load(lo8, lvl(8) + ((i * 32) << log_d)) for i in range(8)]
            for m in range(8):
                v[m], v[m | 8] = bfly(v[m], v[m | 8], w[m])
            d = base + u + (tb << log_d)
            for m in range(16):
                dst[d + (m << (5 + log_d))] = v[m]


def run(dit, log_d, extra_hi):
    log_n = log_d + R + extra_hi
    rng = random.Random(1000 * log_d + 10 * extra_hi + dit)
    data = [rng.randrange(P) for _ in range(1 << log_n)]
    tw = [rng.randrange(P) for _ in range(1 << log_n)]
    want = reference_pass(data, log_n, log_d, tw, dit)
    got = [None] * (1 << log_n)
    for tile_id in range((1 << log_n) >> (R + LOG_T)):
        model_tile(data, got, log_n, log_d, tw, dit, tile_id)
    assert got == want


def test_swap_model_values_to_coeffs():
    run(False, 4, 0)
    run(False, 5, 1)


def test_swap_model_coeffs_to_values():
    run(True, 4, 0)
    run(True, 5, 1)
