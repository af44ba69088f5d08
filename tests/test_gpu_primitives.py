"""-m gpu parity tests: every HIP primitive, called through the C ABI, bit-exact vs the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 14293326489335486720


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import zk_evm_amd
    c = zk_evm_amd.Context(0)
    c.use_torch_current_stream()
    return c


def test_native_library_is_loaded(ctx):
    # the round-end harness records which .so files are mapped; make the requirement explicit
    maps = open("/proc/self/maps").read()
    assert "libzkstark_hip.so" in maps


def test_poseidon_permute(ctx, oracle):
    from tests.gpu_util import to_dev, to_host, ptr, rand_u64
    rng = np.random.default_rng(11)
    n = 1000
    st = rand_u64(rng, (n, 12))
    st[0] = 0
    st[1] = np.arange(12)
    st[2] = 0xFFFFFFFFFFFFFFFF
    d = to_dev(st)
    ctx.check(ctx.lib.zk_poseidon_permute(ctx.handle, ptr(d), n))
    got = to_host(d)
    for i in range(n):
        exp = oracle.poseidon_permute(st[i])
        assert np.array_equal(got[i], exp), i
    assert int(got[0][0]) == 0x3C18A9786CB0B359  # upstream KAT straight from the GPU


def test_keccak_f1600(ctx, oracle):
    from tests.gpu_util import to_dev, to_host, ptr
    rng = np.random.default_rng(12)
    n = 300
    st = rng.integers(0, 1 << 64, size=(n, 25), dtype=np.uint64)
    st[0] = 0
    d = to_dev(st)
    ctx.check(ctx.lib.zk_keccak_f1600(ctx.handle, ptr(d), n))
    got = to_host(d)
    for i in range(n):
        e = st[i].copy()
        oracle.lib.orc_keccak_f1600(e)
        assert np.array_equal(got[i], e), i


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 7, 10, 11, 12, 13, 16])
def test_ifft_fft_roundtrip_and_parity(ctx, oracle, log_n):
    from tests.gpu_util import to_dev, to_host, ptr, rand_u64
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    n_cols = 5 if log_n <= 12 else 2
    vals = rand_u64(rng, (n_cols, n))
    d = to_dev(vals)
    ctx.check(ctx.lib.zk_ifft(ctx.handle, ptr(d), n, n_cols, log_n))
    got = to_host(d)
    for c in range(n_cols):
        e = vals[c].copy()
        oracle.lib.orc_ifft(e, log_n)
        assert np.array_equal(got[c], e), (log_n, c)
    ctx.check(ctx.lib.zk_fft(ctx.handle, ptr(d), n, n_cols, log_n))
    back = to_host(d)
    assert np.array_equal(back, vals % np.uint64(0xFFFFFFFF00000001))


@pytest.mark.parametrize("log_n", [1, 5, 11, 12, 14])
def test_coset_fft_ifft(ctx, oracle, log_n):
    from tests.gpu_util import to_dev, to_host, ptr, rand_u64
    rng = np.random.default_rng(200 + log_n)
    n = 1 << log_n
    n_cols = 3
    shift = int(rng.integers(2, 1 << 63))
    co = rand_u64(rng, (n_cols, n))
    d = to_dev(co)
    ctx.check(ctx.lib.zk_coset_fft(ctx.handle, ptr(d), n, n_cols, log_n, shift))
    got = to_host(d)
    for c in range(n_cols):
        e = co[c].copy()
        oracle.lib.orc_coset_fft(e, log_n, shift)
        assert np.array_equal(got[c], e)
    ctx.check(ctx.lib.zk_coset_ifft(ctx.handle, ptr(d), n, n_cols, log_n, shift))
    assert np.array_equal(to_host(d), co % np.uint64(0xFFFFFFFF00000001))


@pytest.mark.parametrize("log_n,rate_bits", [(0, 1), (1, 1), (4, 1), (4, 3), (10, 1), (11, 1), (12, 2), (15, 1)])
def test_lde(ctx, oracle, log_n, rate_bits):
    from tests.gpu_util import to_dev, to_host, ptr, rand_u64
    import torch
    rng = np.random.default_rng(300 + log_n)
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    n_cols = 3
    co = rand_u64(rng, (n_cols, n))
    d = to_dev(co)
    out = torch.zeros((n_cols, N), dtype=torch.int64, device="cuda")
    ctx.check(ctx.lib.zk_lde(ctx.handle, ptr(d), n, ptr(out), N, n_cols, log_n, rate_bits))
    got = to_host(out)
    for c in range(n_cols):
        e = np.zeros(N, dtype=np.uint64)
        oracle.lib.orc_lde(np.ascontiguousarray(co[c]), log_n, rate_bits, e)
        assert np.array_equal(got[c], e), (log_n, rate_bits, c)


@pytest.mark.parametrize("hasher", [0, 1])
@pytest.mark.parametrize("n_cols", [1, 3, 4, 5, 8, 9, 16, 17, 18, 34, 35, 116])
def test_hash_rows(ctx, oracle, hasher, n_cols):
    from tests.gpu_util import to_dev, to_host, ptr, rand_u64
    import torch
    rng = np.random.default_rng(400 + n_cols)
    n_rows = 300  # not a multiple of the block size: exercises the ragged tail
    cols = rand_u64(rng, (n_cols, n_rows))
    d = to_dev(cols)
    dig = torch.zeros((n_rows, 4), dtype=torch.int64, device="cuda")
    ctx.check(ctx.lib.zk_hash_rows(ctx.handle, hasher, ptr(d), n_rows, n_cols, n_rows, ptr(dig)))
    got = to_host(dig)
    L = oracle.lib
    for r in range(n_rows):
        row = np.ascontiguousarray(cols[:, r])
        if hasher == 0:
            e = np.zeros(4, dtype=np.uint64)
            L.orc_poseidon_hash_or_noop(row, n_cols, e)
        else:
            e8 = np.zeros(32, dtype=np.uint8)
            L.orc_keccak25_hash_or_noop(row, n_cols, e8)
            e = e8.view(np.uint64)
        assert np.array_equal(got[r], e), (hasher, n_cols, r)


@pytest.mark.parametrize("hasher", [0, 1])
@pytest.mark.parametrize("log_leaves,cap_height", [(0, 0), (3, 0), (4, 4), (9, 2), (12, 4)])
def test_merkle_build(ctx, oracle, hasher, log_leaves, cap_height):
    from tests.gpu_util import to_dev, to_host, ptr
    import torch
    rng = np.random.default_rng(500 + log_leaves)
    N = 1 << log_leaves
    leaves = rng.integers(0, 0xFFFFFFFF00000001, size=(N, 6), dtype=np.uint64)
    nd = ctx.lib.zk_merkle_num_digests(log_leaves, cap_height)
    exp = np.zeros((nd, 4), dtype=np.uint64)
    oracle.lib.orc_merkle_build(leaves, log_leaves, 6, cap_height, hasher, exp)
    assert nd == oracle.lib.orc_merkle_num_digests(log_leaves, cap_height)
    dig = torch.zeros((nd, 4), dtype=torch.int64, device="cuda")
    dig[:N] = to_dev(exp[:N])
    ctx.check(ctx.lib.zk_merkle_build(ctx.handle, hasher, ptr(dig), log_leaves, cap_height))
    assert np.array_equal(to_host(dig), exp)


def _edge_operands():
    P = 0xFFFFFFFF00000001
    M = (1 << 64) - 1
    base = [0, 1, 2, 7, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 33) - 1, 1 << 63,
            (1 << 63) + 1, P - 2, P - 1, P, P + 1, M - 1, M, 0xFFFFFFFF00000000,
            0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFE, 0x8000000080000000, 0xFFFFFFFF,
            0x100000000, 0xFFFFFFFFFFFFFFFE, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFE00000001,
            0xFFFFFFFE00000002, 0x0000000100000001, 0xFFFFFFFF80000000, 0x00000001FFFFFFFF]
    return base


def test_field_ops_edge_cases(ctx):
    """Every carry / borrow path of the hand-scheduled gfx950 field ops, vs Python big ints.
    Random data hits the rare fix-up paths with probability ~2^-32, so operands are crafted and
    the test asserts (on the CPU) that each path's trigger condition occurs in the operand set."""
    from tests.gpu_util import to_dev, to_host, ptr
    import torch
    P = 0xFFFFFFFF00000001
    rng = np.random.default_rng(77)
    base = _edge_operands()
    # products whose 128-bit image has special words: search a few structured multiplicands
    extra = [int(x) for x in rng.integers(0, 1 << 64, size=64, dtype=np.uint64)]
    vals = base + extra
    A = np.array([a for a in vals for _ in vals], dtype=np.uint64)
    B = np.array([b for _ in vals for b in vals], dtype=np.uint64)
    # trigger-condition coverage for the multiply fold: T = a*b as words T0..T3
    cov = {"borrow_T3": 0, "carry_T2": 0, "t2_borrow": 0, "add2": 0, "sub2": 0}
    for a, b in zip(A.tolist(), B.tolist()):
        t = a * b
        T0, T1, T2, T3 = (t & 0xFFFFFFFF, (t >> 32) & 0xFFFFFFFF, (t >> 64) & 0xFFFFFFFF, t >> 96)
        lo64 = t & ((1 << 64) - 1)
        if lo64 < T3:
            cov["borrow_T3"] += 1
        v = (lo64 - T3) % (1 << 64)
        if lo64 < T3:
            v = (v - 0xFFFFFFFF) % (1 << 64)
        if (v & 0xFFFFFFFF) < T2:
            cov["t2_borrow"] += 1
        if v + T2 * 0xFFFFFFFF >= 1 << 64:
            cov["carry_T2"] += 1
        s = a + b
        if s >= 1 << 64 and (s % (1 << 64)) + 0xFFFFFFFF >= 1 << 64:
            cov["add2"] += 1
        d = a - b
        if d < 0 and (d % (1 << 64)) < 0xFFFFFFFF:
            cov["sub2"] += 1
    assert all(v > 0 for v in cov.values()), cov
    n = A.size
    dA, dB = to_dev(A), to_dev(B)
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    pyops = {0: lambda a, b: (a + b) % P, 1: lambda a, b: (a - b) % P, 2: lambda a, b: (a * b) % P,
             3: lambda a, b: (a * a) % P, 8: lambda a, b: (a * b) % P}   # 8: gl_mul_fast; cov["borrow_T3"] enters its cold block
    for op, f in pyops.items():
        ctx.check(ctx.lib.zk_gl_vec_op(ctx.handle, op, ptr(dA), ptr(dB), ptr(out), n))
        got = to_host(out)
        exp = np.array([f(a, b) for a, b in zip(A.tolist(), B.tolist())], dtype=np.uint64)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (op, hex(int(A[bad[0]])), hex(int(B[bad[0]])), hex(int(got[bad[0]])), hex(int(exp[bad[0]])))
    # inverse
    ctx.check(ctx.lib.zk_gl_vec_op(ctx.handle, 4, ptr(dA), ptr(dB), ptr(out), len(vals)))
    got = to_host(out)[: len(vals)]
    for a, g in zip(A[: len(vals)].tolist(), got.tolist()):
        assert (a * g) % P == (1 if a % P else 0)


def test_field_ops_random(ctx):
    from tests.gpu_util import to_dev, to_host, ptr
    import torch
    P = 0xFFFFFFFF00000001
    rng = np.random.default_rng(78)
    n = 200000
    A = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    B = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    dA, dB = to_dev(A), to_dev(B)
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    Ao, Bo = A.astype(object), B.astype(object)
    for op, exp in ((0, (Ao + Bo) % P), (1, (Ao - Bo) % P), (2, (Ao * Bo) % P), (3, (Ao * Ao) % P), (8, (Ao * Bo) % P)):
        ctx.check(ctx.lib.zk_gl_vec_op(ctx.handle, op, ptr(dA), ptr(dB), ptr(out), n))
        assert np.array_equal(to_host(out), exp.astype(np.uint64)), op


def test_canonical_product_and_one_correction_butterfly(ctx):
    """The NTT butterfly (csrc/ntt.cuh `ntt_bfly`): `gl_mul_canon` must return a value < p for ANY u64 operands -- its
    extra correction only fires for folds of the form 0xFFFFFFFF:lo, lo >= 1, which random operands never produce -- and
    the one-correction add / sub must be exact for every lazy first operand.  Edge operands are constructed, not drawn."""
    from tests.gpu_util import to_dev, to_host, ptr
    import torch
    P = 0xFFFFFFFF00000001
    M = (1 << 64) - 1
    edge = [0, 1, 2, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, P - 2, P - 1, P, P + 1, P + 2, M - 1, M, 0xFFFFFFFF00000000,
            0xFFFFFFFEFFFFFFFF, 1 << 63, (1 << 63) + 1, 1 << 48, P - (1 << 48), 0x00000001FFFFFFFF, 0xFFFFFFFF00000002]
    pairs = [(a, b) for a in edge for b in edge]
    # products whose 128-bit value folds to [p, 2^64) without the final carry: T = [0 : Y : X] with X + EPS * Y in range
    for y in (0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 1):
        for x_lo in (1, 2, 0xFFFFFFFF):
            target = (0xFFFFFFFF << 32) | x_lo              # the non-canonical fold we want
            x = (target - y * 0xFFFFFFFF) % (1 << 64)
            t = (y << 64) | x
            # factor t = a * b with a = 2^32 when divisible, else use (t, 1) only if it fits in 64 bits
            if t % (1 << 32) == 0 and (t >> 32) < (1 << 64):
                pairs.append((1 << 32, t >> 32))
    # the same with a non-zero top limb: (z (2^32 + 1) + 1) * (2^64 - 1) folds to 2^64 - 2 - 3z without a carry
    pairs += [(z * ((1 << 32) + 1) + 1, M) for z in (1, 2, 12345, 0x55555554)]
    rng = np.random.default_rng(5)
    extra = rng.integers(0, 1 << 64, size=(4096, 2), dtype=np.uint64)
    near = np.array([[(P + int(d)) & M, 1] for d in range(-40, 41)], dtype=np.uint64)   # a * 1: the fold is a itself
    A = np.concatenate([np.array([p[0] for p in pairs], dtype=np.uint64), extra[:, 0], near[:, 0]])
    B = np.concatenate([np.array([p[1] for p in pairs], dtype=np.uint64), extra[:, 1], near[:, 1]])

    def folds_non_canonical(a, b):                          # the lazy fold of gl.cuh, no final carry, result >= p
        t = a * b
        x, y, z = t & M, (t >> 64) & 0xFFFFFFFF, t >> 96
        v = x - z
        if v < 0:
            v = (v - 0xFFFFFFFF) % (1 << 64)
        r = v + y * 0xFFFFFFFF
        return P <= r < (1 << 64)
    hit = sum(1 for a, b in zip(A.tolist(), B.tolist()) if folds_non_canonical(a, b))
    assert sum(1 for a, b in pairs[-4:] if folds_non_canonical(a, b)) == 4
    assert hit >= 40, hit                                   # the no-carry non-canonical fold is really exercised
    n = A.size
    dA, dB = to_dev(A), to_dev(B)
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    Ao, Bo = A.astype(object), B.astype(object)
    for op, exp in ((5, (Ao * Bo) % P), (6, (Ao + Bo * Bo) % P), (7, (Ao - Bo * Bo) % P)):
        ctx.check(ctx.lib.zk_gl_vec_op(ctx.handle, op, ptr(dA), ptr(dB), ptr(out), n))
        got = to_host(out)
        bad = np.nonzero(got != exp.astype(np.uint64))[0]
        assert bad.size == 0, (op, hex(int(A[bad[0]])), hex(int(B[bad[0]])), hex(int(got[bad[0]])), hex(int(exp[bad[0]])))

