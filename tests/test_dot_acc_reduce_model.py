"""The word-level algorithm of the device's `dot_acc_reduce` (zk_evm_amd/csrc/fri.cuh: seven carry adds, one 64-bit
subtraction with one borrow correction, one multiply-add fold with one carry correction) restated with Python integers
and compared with plain modular arithmetic -- including the corner values the real data almost never produces (a borrow
needs [w1:w0] < [w4:w3], probability ~2^-32 per fold; carry counts up to the documented 2^31 bound).  The instruction
encoding itself is covered by the GPU parity suites; this pins the arithmetic they rely on."""
import random

P = 0xFFFFFFFF00000001
M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def fold(s00, s01, s11, h00, h01, h11):
    a0, a1 = s00 & M32, s00 >> 32
    b0, b1 = s01 & M32, s01 >> 32
    c0, c1 = s11 & M32, s11 >> 32
    t = a1 + b0; w1 = t & M32; cy = t >> 32
    t = b1 + c0 + cy; w2 = t & M32; cy = t >> 32
    t = c1 + cy; w3 = t & M32; cy = t >> 32
    t = h11 + cy; w4 = t & M32
    assert t >> 32 == 0
    t = w2 + h00; w2 = t & M32; cy = t >> 32
    t = w3 + h01 + cy; w3 = t & M32; cy = t >> 32
    t = w4 + cy; w4 = t & M32
    assert t >> 32 == 0
    x, y = (w1 << 32) | a0, (w4 << 32) | w3
    q = (x - y) & M64
    if x < y:                       # borrow: -= EPS (== += p); a second borrow is impossible while w4 < 2^31
        assert q >= 0xFFFFFFFF
        q -= 0xFFFFFFFF
    r = q + w2 * 0xFFFFFFFF
    if r >> 64:                     # carry: 2^64 == 2^32 - 1 again; cannot carry twice
        r = (r & M64) + 0xFFFFFFFF
        assert r >> 64 == 0
    return r


def value(s00, s01, s11, h00, h01, h11):
    return (s00 + (h00 << 64) + ((s01 + (h01 << 64)) << 32) + ((s11 + (h11 << 64)) << 64)) % P


def test_dot_acc_reduce_word_algorithm():
    rng = random.Random(5)
    corners64 = [0, 1, M64, M64 - 1, P, P - 1, 1 << 63, M32, 1 << 32]
    corners_h = [0, 1, 2, 5000, 1 << 20, (1 << 31) - 8]

    def r64():
        return rng.choice(corners64) if rng.random() < 0.2 else rng.getrandbits(64)

    def rh():
        return rng.choice(corners_h) if rng.random() < 0.3 else rng.getrandbits(rng.choice([3, 10, 20, 30]))

    borrows = 0
    for _ in range(200000):
        args = (r64(), r64(), r64(), rh(), rh(), rh())
        assert fold(*args) % P == value(*args), args
    # the borrow path on purpose: small [w1:w0], large [w4:w3]
    for _ in range(20000):
        args = (rng.getrandbits(20), rng.getrandbits(10), rng.getrandbits(64), 0, rng.getrandbits(30), rng.getrandbits(30))
        borrows += 1
        assert fold(*args) % P == value(*args), args
    assert borrows
