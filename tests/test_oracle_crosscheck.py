"""Cross-checks of the CPU oracle against an independent, slow, exact restatement.

`ieee.*` below simulates IEEE-754 binary32 round-to-nearest-even with exact rational arithmetic
(fractions.Fraction): every add / mul / sub / div / sqrt is computed exactly and then rounded once.
Walking the reference's operation order (tensor_store/src/hnsw.rs:168-229, vector_engine/src/lib.rs:
2231-2266) with it gives the value any conforming implementation of the reference must produce.  The
C oracle and the numpy twin must reproduce it BIT FOR BIT — which also proves the C build is free of
fused multiply-adds, x87 excess precision and reassociation.
"""
import math
import struct
from fractions import Fraction

import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as on

F = np.float32


class ieee:
    MANT = 24
    EMIN = -126
    EMAX = 127

    @staticmethod
    def to_frac(x):
        x = float(np.float32(x))
        return Fraction(x)

    @classmethod
    def round(cls, x):
        """Fraction -> nearest binary32 (ties to even) as python float; handles subnormals/overflow."""
        if x == 0:
            return 0.0
        sign = -1 if x < 0 else 1
        a = abs(x)
        e = a.numerator.bit_length() - a.denominator.bit_length()
        if Fraction(2) ** e > a:
            e -= 1
        elif Fraction(2) ** (e + 1) <= a:
            e += 1
        e = max(e, cls.EMIN)
        scale = Fraction(2) ** (e - (cls.MANT - 1))
        q = a / scale
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
            n += 1
        v = Fraction(n) * scale
        if v >= Fraction(2) ** (cls.EMAX + 1):
            return sign * math.inf
        return sign * float(v)

    @classmethod
    def add(cls, a, b):
        if a == 0 and b == 0:  # signed zeros: (-0)+(-0) = -0, otherwise +0
            return -0.0 if (math.copysign(1, a) < 0 and math.copysign(1, b) < 0) else 0.0
        return cls.round(Fraction(a) + Fraction(b))

    @classmethod
    def sub(cls, a, b):
        return cls.add(a, -b)

    @classmethod
    def mul(cls, a, b):
        if a == 0 or b == 0:
            return math.copysign(0.0, math.copysign(1, a) * math.copysign(1, b))
        return cls.round(Fraction(a) * Fraction(b))

    @classmethod
    def div(cls, a, b):
        if a == 0:
            return math.copysign(0.0, math.copysign(1, a) * math.copysign(1, b))
        return cls.round(Fraction(a) / Fraction(b))

    @classmethod
    def sqrt(cls, a):
        if a == 0:
            return a
        x = Fraction(a)
        # integer sqrt with 80 extra bits decides the rounding exactly unless x is a perfect square
        k = 160
        n = (x.numerator << (2 * k)) // x.denominator
        r = math.isqrt(n)
        exact = (r * r == n) and ((x.numerator << (2 * k)) % x.denominator == 0)
        cand = Fraction(r, 1 << k)
        if not exact:
            cand += Fraction(1, 1 << (k + 8))  # sticky bit: strictly above r/2^k, never a tie
        return cls.round(cand)


def ref_dot8(a, b):
    """hnsw.rs:168-193 walked with exact-rounded ops."""
    n = len(a)
    chunks, rem = divmod(n, 8)
    acc = [0.0] * 8
    for c in range(chunks):
        for l in range(8):
            acc[l] = ieee.add(acc[l], ieee.mul(a[8 * c + l], b[8 * c + l]))
    r = -0.0
    for l in range(8):
        r = ieee.add(r, acc[l])
    for i in range(chunks * 8, n):
        r = ieee.add(r, ieee.mul(a[i], b[i]))
    return r


def ref_score(q, v, metric):
    q = [float(x) for x in q]
    v = [float(x) for x in v]
    if metric == 2:
        return ref_dot8(q, v)
    if metric == 1:  # lib.rs:2249-2253 then 1/(1+d)
        s = -0.0
        for x, y in zip(q, v):
            d = ieee.sub(x, y)
            s = ieee.add(s, ieee.mul(d, d))
        return ieee.div(1.0, ieee.add(1.0, ieee.sqrt(s)))
    qmag = ieee.sqrt(ref_dot8(q, q))
    vmag = ieee.sqrt(ref_dot8(v, v))
    if qmag == 0.0 or vmag == 0.0:
        return 0.0
    return ieee.div(ref_dot8(q, v), ieee.mul(qmag, vmag))


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def test_rounding_primitive_against_numpy():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(300) * 10.0 ** rng.integers(-20, 20, 300)).astype(F)
    b = (rng.standard_normal(300) * 10.0 ** rng.integers(-20, 20, 300)).astype(F)
    with np.errstate(over="ignore", under="ignore"):
        for x, y in zip(a, b):
            assert bits(ieee.add(float(x), float(y))) == bits(F(x) + F(y))
            assert bits(ieee.mul(float(x), float(y))) == bits(F(x) * F(y))
            assert bits(ieee.div(float(x), float(y))) == bits(F(x) / F(y))
            assert bits(ieee.sqrt(abs(float(x)))) == bits(np.sqrt(np.abs(F(x))))
    # subnormal results and ties-to-even
    tiny = float(np.float32(1e-45))
    assert bits(ieee.mul(tiny, 0.5)) == bits(F(tiny) * F(0.5))
    assert bits(ieee.add(16777216.0, 1.0)) == bits(F(16777216.0) + F(1.0))
    assert bits(ieee.add(16777218.0, 1.0)) == bits(F(16777218.0) + F(1.0))


@pytest.mark.parametrize("d", [1, 3, 8, 9, 16, 23, 40, 64])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_oracle_bit_exact_vs_rational_simulator(d, metric):
    rng = np.random.default_rng(100 * d + metric)
    for trial in range(6):
        scale = [1.0, 1e-3, 1e3, 1e-20, 3e5, 1.0][trial]
        q = (rng.standard_normal(d) * scale).astype(F)
        v = (rng.standard_normal(d) * scale).astype(F)
        if trial == 5:
            v = q.copy()  # identical vectors: dist 0 -> score 1, cosine ~1
        exp = ref_score(q, v, metric)
        got_c = oc.score(q, v, metric)
        got_np = on.scores(v[None, :], q, metric)[0]
        assert bits(got_c) == bits(exp), (d, metric, trial, float(got_c), exp)
        assert bits(got_np) == bits(exp), (d, metric, trial, float(got_np), exp)


def test_lane_order_is_observable():
    """A vector where the 8-lane order, a sequential sum and an FMA chain give three different f32
    results: the oracle must produce the 8-lane one."""
    rng = np.random.default_rng(42)
    found = False
    for _ in range(200):
        a = (rng.standard_normal(64) * 100).astype(F)
        b = (rng.standard_normal(64) * 100).astype(F)
        lanes = ref_dot8([float(x) for x in a], [float(x) for x in b])
        seq = 0.0
        for x, y in zip(a, b):
            seq = ieee.add(seq, ieee.mul(float(x), float(y)))
        if bits(lanes) != bits(seq):
            found = True
            assert bits(oc.dot8(a, b)) == bits(lanes)
            assert bits(on.dot8(a, b)) == bits(lanes)
            break
    assert found


def test_native_build_matches_plain_build():
    """-O3 -march=native (the timed CPU baseline) must not change a single bit."""
    rng = np.random.default_rng(9)
    A = rng.standard_normal((2000, 100)).astype(F)
    q = rng.standard_normal(100).astype(F)
    for m in (0, 1, 2):
        a = oc.scores_all(A, q, m, native=False)
        b = oc.scores_all(A, q, m, native=True, nthreads=4)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
