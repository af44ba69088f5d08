"""CPU model of the lane-swap NTT kernels (zk_evm_amd/csrc/ntt_swap.cuh): the kernels' register / lane / wave bookkeeping --
which index bit every register of every lane holds after each v_permlane16/32_swap, which twiddle entry each butterfly reads,
the LDS exchanges and wave-local transposes, the two-coset form of the extension's first pass -- restated expression by
expression in Python and run against the plain definition of the pass (the stages of ntt_pass_kernel / ntt_step, one
butterfly at a time).  The strided passes and the values -> coefficients direction use twiddle TABLES of random field elements
(what is checked is the index algebra, and a wrong index cannot cancel); the coefficients -> values contiguous pass uses the
real roots of unity, because evaluating on the second coset through a second scale table is an identity of the field, not of
the indices.  The arithmetic itself is the GPU tests' business (tests/test_gpu_commit.py pins the kernels to ntt_pass_kernel
bit for bit)."""
import random

P = 0xFFFFFFFF00000001
LOG_T = 4
ROOT_2_32 = 7277203076849721926


def bfly(a, b, w):
    t = b * w % P
    return (a + t) % P, (a - t) % P


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


# ---- the definition of a pass (ntt.cuh ntt_pass_kernel / ntt_step) -------------------------------------------------------------
def reference_pass(data, log_n, log_d, r, tw, dit, first_stage=0):
    out = list(data)
    ks = range(first_stage, r) if dit else range(r - 1, -1, -1)
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


# ---- the model's pieces ------------------------------------------------------------------------------------------------
def lane_swap(v, lanebit, regbit):
    """ntt_swap16<LANEBIT, REGBIT>: (register a, lane bit = 1) <-> (register b, lane bit = 0) for a = m, b = m | 1 << regbit"""
    for lane in range(64):
        if (lane >> lanebit) & 1:
            continue
        partner = lane | (1 << lanebit)
        for m in range(16):
            if not m & (1 << regbit):
                a, b = m, m | (1 << regbit)
                v[partner][a], v[lane][b] = v[lane][b], v[partner][a]


def stage16(v, bit, w_of):
    """ZK_NTT_STAGE16 over every lane: w_of(lane, m)"""
    for lane in range(64):
        for m in range(16):
            if not m & (1 << bit):
                v[lane][m], v[lane][m | (1 << bit)] = bfly(v[lane][m], v[lane][m | (1 << bit)], w_of(lane, m))


def L4(lane):
    return (lane >> 4) & 1


def L5(lane):
    return lane >> 5


def tw_load(tw, lane_byte_off, uniform_index):
    assert lane_byte_off % 8 == 0
    return tw[uniform_index + lane_byte_off // 8]


def swap_dif6(v, tw, s_top, H):
    first = lambda j: ((1 << (s_top + j)) - 1) + (H << j)
    # entry: registers [3..0] = [q5 q4 q3 q2], lane 5 = q1, lane 4 = q0: four register stages, then one swap per lane stage
    stage16(v, 3, lambda lane, m: tw[first(0)])
    stage16(v, 2, lambda lane, m: tw[first(1) + (m >> 3)])
    stage16(v, 1, lambda lane, m: tw[first(2) + (m >> 2)])
    stage16(v, 0, lambda lane, m: tw[first(3) + (m >> 1)])
    lane_swap(v, 5, 3)
    stage16(v, 3, lambda lane, m: tw_load(tw, L5(lane) * 64, first(4) + (m & 7)))
    lane_swap(v, 4, 2)
    stage16(v, 2, lambda lane, m: tw_load(tw, (L5(lane) * 16 + L4(lane) * 8) * 8, first(5) + ((m >> 1) & 1) * 4 + (m & 1) * 2 + (m >> 3)))


def dif6_row(m):
    return ((m & 2) << 2) | ((m & 1) << 2) | ((m & 8) >> 2) | ((m & 4) >> 2)


def swap_dit6(vs, tw, log_d, xl8_of):
    """vs: the NB register files that share the twiddles"""
    lvl = lambda k: (1 << (log_d + k)) - 1

    def each(f):
        for v in vs:
            f(v)
    # entry: registers [3..0] = [q0 q1 q3 q2], lane 5 = q4, lane 4 = q5
    each(lambda v: stage16(v, 3, lambda lane, m: tw_load(tw, xl8_of(lane), lvl(0))))
    each(lambda v: stage16(v, 2, lambda lane, m: tw_load(tw, xl8_of(lane), lvl(1) + ((m >> 3) << log_d))))
    each(lambda v: stage16(v, 0, lambda lane, m: tw_load(tw, xl8_of(lane), lvl(2) + ((((m >> 2) & 1) * 2 + (m >> 3)) << log_d))))
    each(lambda v: stage16(v, 1, lambda lane, m: tw_load(tw, xl8_of(lane), lvl(3) + (((m & 1) * 4 + ((m >> 2) & 1) * 2 + (m >> 3)) << log_d))))
    each(lambda v: lane_swap(v, 5, 3))

    def w4(lane, m):
        i = m & 7
        lo8 = xl8_of(lane) + ((L5(lane) << log_d) << 3)
        return tw_load(tw, lo8, lvl(4) + ((((i >> 1) & 1) * 8 + (i & 1) * 4 + ((i >> 2) & 1) * 2) << log_d))
    each(lambda v: stage16(v, 3, w4))
    each(lambda v: lane_swap(v, 4, 2))

    def w5(lane, m):
        i = (m >> 3) * 4 + (m & 3)
        lo8 = xl8_of(lane) + (((L4(lane) * 2 + L5(lane)) << log_d) << 3)
        return tw_load(tw, lo8, lvl(5) + ((((i >> 2) & 1) * 16 + ((i >> 1) & 1) * 8 + (i & 1) * 4) << log_d))
    each(lambda v: stage16(v, 2, w5))


def dit6_row(m):
    return ((m & 4) << 3) | ((m & 8) << 1) | ((m & 2) << 2) | ((m & 1) << 2)


def dit6_row_in(m):
    return ((m & 2) << 2) | ((m & 1) << 2) | ((m & 4) >> 1) | ((m & 8) >> 3)


def check_banks(addr_of_lane):
    """a half wave's 64-bit LDS accesses touch every bank pair at most once"""
    for half in (0, 32):
        seen = set()
        for lane in range(half, half + 32):
            b = addr_of_lane(lane) % 32            # 64 banks of 4 bytes = 32 of 8
            assert b not in seen, "LDS bank conflict"
            seen.add(b)


# ---- ntt_strided_swap_kernel<DIT, R> --------------------------------------------------------------------------------------
def strided_tile(src, dst, log_n, log_d, R, tw, dit, tile_id):
    A = R - 6
    waves = 1 << A
    log_lo_tiles = log_d - LOG_T
    hi_idx, lo_tile = tile_id >> log_lo_tiles, tile_id & ((1 << log_lo_tiles) - 1)
    base = (hi_idx << (log_d + R)) + (lo_tile << LOG_T)
    V = [[[None] * 16 for _ in range(64)] for _ in range(waves)]
    lds = {}
    if not dit:
        for wv in range(waves):
            for lane in range(64):
                s = base + (lane & 15) + (((L5(lane) << (R - 5)) | (L4(lane) << (R - 6)) | wv) << log_d)
                for m in range(16):
                    V[wv][lane][m] = src[s + (m << (R - 4 + log_d))]
        s_top = log_n - log_d - R
        for wv in range(waves):
            swap_dif6(V[wv], tw, s_top, hi_idx)
        for wv in range(waves):
            for m in range(16):
                def addr(lane):
                    t = (L5(lane) << (R - 1)) | (L4(lane) << (R - 2)) | wv | (dif6_row(m) << A)
                    t ^= ((t >> 4) ^ (t >> (R - 2))) & 1
                    return (t << 4) + (lane & 15)
                check_banks(addr)
                for lane in range(64):
                    assert addr(lane) not in lds
                    lds[addr(lane)] = V[wv][lane][m]
        assert len(lds) == 1 << (R + 4)
        lvl = lambda k: ((1 << (s_top + R - 1 - k)) - 1) + (hi_idx << (R - 1 - k))
        for wv in range(waves):
            v = V[wv]
            for m in range(16):
                def addr(lane):
                    t = (wv << 6) | (L5(lane) << 5) | (L4(lane) << 4) | m
                    t ^= ((t >> 4) ^ (t >> (R - 2))) & 1
                    return (t << 4) + (lane & 15)
                check_banks(addr)
                for lane in range(64):
                    v[lane][m] = lds[addr(lane)]
            thl = lambda lane: L5(lane) * 2 + L4(lane)
            thu = wv << 2
            if A >= 4:
                stage16(v, 3, lambda lane, m: tw_load(tw, thl(lane) * 8, lvl(3) + thu))
            if A >= 3:
                stage16(v, 2, lambda lane, m: tw_load(tw, thl(lane) * 16, lvl(2) + thu * 2 + (m >> 3)))
            if A >= 2:
                stage16(v, 1, lambda lane, m: tw_load(tw, thl(lane) * 32, lvl(1) + thu * 4 + (m >> 2)))
            stage16(v, 0, lambda lane, m: tw_load(tw, thl(lane) * 64, lvl(0) + thu * 8 + (m >> 1)))
            for lane in range(64):
                tb = (wv << 6) | (L5(lane) << 5) | (L4(lane) << 4)
                d = base + (lane & 15) + (tb << log_d)
                for m in range(16):
                    dst[d + (m << log_d)] = v[lane][m]
    else:
        for wv in range(waves):
            for lane in range(64):
                s = base + (lane & 15) + (((wv << 6) | (L4(lane) << 5) | (L5(lane) << 4)) << log_d)
                for m in range(16):
                    V[wv][lane][m] = src[s + (dit6_row_in(m) << log_d)]
        xl8 = lambda lane: ((lo_tile << LOG_T) + (lane & 15)) * 8
        for wv in range(waves):
            swap_dit6([V[wv]], tw, log_d, xl8)
        for wv in range(waves):
            for m in range(16):
                def addr(lane):
                    tb = ((wv << 6) | (L4(lane) << 1) | L5(lane)) ^ L4(lane)
                    return ((tb | dit6_row(m)) << 4) + (lane & 15)
                check_banks(addr)
                for lane in range(64):
                    assert addr(lane) not in lds
                    lds[addr(lane)] = V[wv][lane][m]
        assert len(lds) == 1 << (R + 4)
        lvl = lambda k: (1 << (log_d + k)) - 1
        for wv in range(waves):
            v = V[wv]
            tb = lambda lane: (wv << 2) | (L5(lane) << 1) | L4(lane)
            for m in range(16):
                def addr(lane):
                    return ((m << (R - 4)) | (tb(lane) ^ L5(lane))) * 16 + (lane & 15)
                check_banks(addr)
                for lane in range(64):
                    v[lane][m] = lds[addr(lane)]
            lo8 = lambda lane: xl8(lane) + ((tb(lane) << log_d) << 3)
            if R - 4 >= 6:
                stage16(v, 0, lambda lane, m: tw_load(tw, lo8(lane), lvl(R - 4)))
            if R - 3 >= 6:
                stage16(v, 1, lambda lane, m: tw_load(tw, lo8(lane), lvl(R - 3) + (((m & 1) << (R - 4)) << log_d)))
            if R - 2 >= 6:
                stage16(v, 2, lambda lane, m: tw_load(tw, lo8(lane), lvl(R - 2) + (((m & 3) << (R - 4)) << log_d)))
            stage16(v, 3, lambda lane, m: tw_load(tw, lo8(lane), lvl(R - 1) + (((m & 7) << (R - 4)) << log_d)))
            for lane in range(64):
                d = base + (lane & 15) + (tb(lane) << log_d)
                for m in range(16):
                    dst[d + (m << (R - 4 + log_d))] = v[lane][m]


def run_strided(dit, R, log_d, extra_hi):
    log_n = log_d + R + extra_hi
    rng = random.Random(1000 * log_d + 100 * R + 10 * extra_hi + dit)
    data = [rng.randrange(P) for _ in range(1 << log_n)]
    tw = [rng.randrange(P) for _ in range(1 << log_n)]
    want = reference_pass(data, log_n, log_d, R, tw, dit)
    got = [None] * (1 << log_n)
    for tile_id in range((1 << log_n) >> (R + LOG_T)):
        strided_tile(data, got, log_n, log_d, R, tw, dit, tile_id)
    assert got == want


def test_strided_swap_values_to_coeffs():
    run_strided(False, 9, 4, 0)
    run_strided(False, 9, 5, 1)
    run_strided(False, 10, 4, 1)
    run_strided(False, 7, 5, 1)
    run_strided(False, 8, 4, 2)


def test_strided_swap_coeffs_to_values():
    run_strided(True, 9, 4, 0)
    run_strided(True, 9, 5, 1)
    run_strided(True, 10, 4, 1)
    run_strided(True, 7, 5, 1)
    run_strided(True, 8, 4, 2)


# ---- ntt_contig_wave_kernel_dif ------------------------------------------------------------------------------------------------
def lds17(row, col):
    return row * 17 + col


def contig_dif_wave(src, dst, log_n, tw, tile_id):
    base = tile_id << 10
    v = [[None] * 16 for _ in range(64)]
    rb = lambda lane: (L5(lane) << 5) | (L4(lane) << 4)
    for lane in range(64):
        rl = (L5(lane) << 1) | L4(lane)
        for m in range(16):
            v[lane][m] = src[base + ((((m << 2) | rl)) << 4) + (lane & 15)]
    s_top = log_n - 10
    swap_dif6(v, tw, s_top, tile_id)
    lds = {}
    for m in range(16):
        addr = lambda lane: lds17(rb(lane) | dif6_row(m), lane & 15)
        check_banks(addr)
        for lane in range(64):
            assert addr(lane) not in lds
            lds[addr(lane)] = v[lane][m]
    for j in range(16):
        check_banks(lambda lane: lds17(lane, j))
        for lane in range(64):
            v[lane][j] = lds[lds17(lane, j)]
    lvl = lambda k: ((1 << (s_top + 9 - k)) - 1) + (tile_id << (9 - k))
    stage16(v, 3, lambda lane, m: tw_load(tw, lane * 8, lvl(3)))
    stage16(v, 2, lambda lane, m: tw_load(tw, lane * 16, lvl(2) + (m >> 3)))
    stage16(v, 1, lambda lane, m: tw_load(tw, lane * 32, lvl(1) + (m >> 2)))
    stage16(v, 0, lambda lane, m: tw_load(tw, lane * 64, lvl(0) + (m >> 1)))
    lds = {}
    for j in range(16):
        for lane in range(64):
            lds[lds17(lane, j)] = v[lane][j]
    for m in range(16):
        check_banks(lambda lane: lds17(rb(lane) | m, lane & 15))
        for lane in range(64):
            dst[base + ((rb(lane) | m) << 4) + (lane & 15)] = lds[lds17(rb(lane) | m, lane & 15)]


def test_contig_wave_values_to_coeffs():
    for log_n in (10, 12):
        rng = random.Random(77 + log_n)
        data = [rng.randrange(P) for _ in range(1 << log_n)]
        tw = [rng.randrange(P) for _ in range(1 << log_n)]
        want = reference_pass(data, log_n, 0, 10, tw, False)
        got = [None] * (1 << log_n)
        for tile_id in range(1 << (log_n - 10)):
            contig_dif_wave(data, got, log_n, tw, tile_id)
        assert got == want


# ---- ntt_contig_wave_kernel_dit<NB> --------------------------------------------------------------------------------------------
def level_table(log_size):
    """ntt_host.inc get_twiddle_levels: tw[D - 1 + k] = (root of order 2 D)^k"""
    tw = [0] * (1 << log_size)
    for lv in range(log_size):
        D = 1 << lv
        w = pow(ROOT_2_32, 1 << (32 - (lv + 1)), P)
        x = 1
        for k in range(D):
            tw[D - 1 + k] = x
            x = x * w % P
    return tw


def contig_dit_wave(src, dst, NB, tw, scales, tile_id):
    sbase = tile_id << 10
    rb = lambda lane: (L5(lane) << 5) | (L4(lane) << 4)
    vs = []
    for b in range(NB):
        v = [[None] * 16 for _ in range(64)]
        for lane in range(64):
            for m in range(16):
                i = ((rb(lane) | m) << 4) + (lane & 15)
                c = src[sbase + i]
                v[lane][m] = c * scales[b][sbase + i] % P if scales[b] is not None else c
        lds = {}
        for m in range(16):
            check_banks(lambda lane: lds17(rb(lane) | m, lane & 15))
            for lane in range(64):
                lds[lds17(rb(lane) | m, lane & 15)] = v[lane][m]
        for j in range(16):
            for lane in range(64):
                v[lane][j] = lds[lds17(lane, j)]
        vs.append(v)
    for v in vs:
        stage16(v, 0, lambda lane, m: tw[0])
        stage16(v, 1, lambda lane, m: tw[1 + (m & 1)])
        stage16(v, 2, lambda lane, m: tw[3 + (m & 3)])
        stage16(v, 3, lambda lane, m: tw[7 + (m & 7)])
    rl = lambda lane: (L4(lane) << 5) | (L5(lane) << 4)
    for v in vs:
        lds = {}
        for j in range(16):
            for lane in range(64):
                lds[lds17(lane, j)] = v[lane][j]
        for m in range(16):
            for lane in range(64):
                v[lane][m] = lds[lds17(dit6_row_in(m) | rl(lane), lane & 15)]
    swap_dit6(vs, tw, 4, lambda lane: (lane & 15) * 8)
    for lane in range(64):
        ro = (L4(lane) << 1) | L5(lane)
        for m in range(16):
            i = ((dit6_row(m) | ro) << 4) + (lane & 15)
            for b in range(NB):
                dst[(sbase + i) * NB + b] = vs[b][lane][m]


def test_contig_wave_coeffs_to_values_one_and_two_cosets():
    for NB, log_src in ((1, 10), (1, 11), (2, 10), (2, 12)):
        log_n = log_src + NB - 1                     # size of the destination
        rng = random.Random(5 * log_src + NB)
        src = [rng.randrange(P) for _ in range(1 << log_src)]
        tw = level_table(log_n)
        s0 = [rng.randrange(P) for _ in range(1 << log_src)] if (log_src & 1) == 0 or NB == 2 else None
        # the definition: replicate (the skipped stages), then stages first_stage .. r - 1 of the contiguous pass
        r = 10 + NB - 1
        rep = [0] * (1 << log_n)
        for se in range(1 << log_src):
            w = src[se] * s0[se] % P if s0 is not None else src[se]
            for j in range(NB):
                rep[se * NB + j] = w
        want = reference_pass(rep, log_n, 0, r, tw, True, first_stage=NB - 1)
        scales = [s0]
        if NB == 2:
            # ntt_host.inc get_wave_coset2_table: the tile's own 2^11-point transform sees its 2^10 coefficients in LOCAL
            # bit-reversed order, whatever the tile: the second coset's factor is (root of order 2^11)^bitrev_10(i mod 2^10)
            w = pow(ROOT_2_32, 1 << (32 - 11), P)
            scales.append([(s0[i] if s0 is not None else 1) * pow(w, bitrev(i & 1023, 10), P) % P for i in range(1 << log_src)])
        got = [None] * (1 << log_n)
        for tile_id in range(1 << (log_src - 10)):
            contig_dit_wave(src, got, NB, tw, scales, tile_id)
        assert got == want, (NB, log_src)
