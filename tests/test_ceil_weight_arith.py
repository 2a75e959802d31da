"""CPU restatement of the arithmetic of edge_w_ceil_i (csrc/tspgpu.hip: the CEIL_2D weight of the matrix-free sweep on integer
coordinates, no f64): f32 root of an f32 d2, k = floor(root), ONE exact remainder e = d2 - k*k in wrapping 32-bit arithmetic,
and the decision between k - 1 .. k + 2.  Checked against the exact integer ceil-sqrt for |dx|, |dy| < 2^22 -- random pairs,
perfect squares and their neighbours, the largest weights the kind is used for -- and for every root the hardware may
return: v_sqrt_f32 is specified to one ulp, so the correctly rounded root AND its two neighbours must all lead to the
same, exact weight."""
import numpy as np


def ceil_sqrt_exact(d2):
    k = np.floor(np.sqrt(d2.astype(np.float64))).astype(np.int64)
    k = np.where(k * k > d2, k - 1, k)
    k = np.where((k + 1) * (k + 1) <= d2, k + 1, k)          # k = isqrt(d2)
    return np.where(k * k == d2, k, k + 1)


def weight(dx, dy, ulps):
    """edge_w_ceil_i with the f32 root moved by `ulps` units in the last place"""
    fx, fy = dx.astype(np.float32), dy.astype(np.float32)                     # exact: |dx| < 2^24
    # fma(fy, fy, fx * fx): the product rounded to f32, then fy*fy + that rounded ONCE (exact in f64: < 2^49)
    p = (fx * fx).astype(np.float32)
    x = (fy.astype(np.float64) * fy.astype(np.float64) + p.astype(np.float64)).astype(np.float32)
    r = np.sqrt(x.astype(np.float64)).astype(np.float32)                      # correctly rounded f32 root
    for _ in range(abs(ulps)):
        r = np.nextafter(r, np.float32(np.inf if ulps > 0 else -np.inf))
    ki = np.maximum(r, 0).astype(np.int64)                                    # (int) truncation
    d2lo = ((dx * dx + dy * dy) & 0xFFFFFFFF)                                 # v_mul_i32_i24 + v_mad_i32_i24: d2 mod 2^32
    e = ((d2lo - ((ki * ki) & 0xFFFFFFFF)) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)
    t1 = 2 * ki + 1
    r0 = ki + (e > 0)
    rare = (e & 0xFFFFFFFF) > (t1 & 0xFFFFFFFF)                               # (unsigned)e > (unsigned)t1: e > 2k + 1 or e < 0
    fix = np.where(e > 0, ki + 2, np.where(e + t1 <= 2, ki - 1, ki))
    return np.where(rare, fix, r0)


def samples():
    rs = np.random.RandomState(11)
    lim = (1 << 22) - 1
    dx = [rs.randint(-lim, lim + 1, size=400000), rs.randint(-3000, 3001, size=200000), rs.randint(-lim, lim + 1, size=100000)]
    dy = [rs.randint(-lim, lim + 1, size=400000), rs.randint(-3000, 3001, size=200000), np.zeros(100000, dtype=np.int64)]
    # perfect squares and their neighbours on an axis, Pythagorean multiples, the corners of the range
    k = rs.randint(1, lim, size=100000)
    for off in (-1, 0, 1):
        dx.append(np.clip(k + off, -lim, lim)); dy.append(np.zeros_like(k))
    m = rs.randint(1, lim // 5, size=100000)
    dx.append(3 * m); dy.append(4 * m)
    dx.append(-(5 * (m // 3))); dy.append(12 * (m // 3))
    dx.append(np.array([0, 1, 0, lim, lim, -lim, lim - 1, 2896309, 2965820])); dy.append(np.array([0, 0, 1, lim, 0, lim, lim, 2896309, 2965821]))
    dx, dy = np.concatenate(dx).astype(np.int64), np.concatenate(dy).astype(np.int64)
    keep = dx * dx + dy * dy < (1 << 44)              # weights below 2^22: what ceil_int() admits
    return dx[keep], dy[keep]


def test_one_remainder_ceil_sqrt_is_exact_for_every_admissible_root():
    dx, dy = samples()
    want = ceil_sqrt_exact(dx * dx + dy * dy)
    assert len(dx) > 1000000 and want.max() > 4000000
    for ulps in (0, -1, 1):
        got = weight(dx, dy, ulps)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (ulps, dx[bad[:5]], dy[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_the_root_stays_within_one_of_the_floor():
    """the bound the decision rests on: floor(f32 root) is floor(sqrt(d2)) - 1, + 0 or + 1, one ulp either way included"""
    dx, dy = samples()
    d2 = dx * dx + dy * dy
    t = ceil_sqrt_exact(d2)
    t = np.where(t * t == d2, t, t - 1)                # floor(sqrt(d2))
    fx, fy = dx.astype(np.float32), dy.astype(np.float32)
    x = (fy.astype(np.float64) ** 2 + (fx * fx).astype(np.float32).astype(np.float64)).astype(np.float32)
    r = np.sqrt(x.astype(np.float64)).astype(np.float32)
    for rr in (r, np.nextafter(r, np.float32(np.inf)), np.nextafter(r, np.float32(-np.inf))):
        k = np.maximum(rr, 0).astype(np.int64)
        assert np.abs(k - t).max() <= 1
