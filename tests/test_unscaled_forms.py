"""The kernel's division and square root without operand scaling (kernels.hip: divisor_of / div_unscaled / sqrt_unscaled,
DESIGN.md 3) restated over exact rationals: for operands in the ranges their call sites guarantee, and for ANY hardware
reciprocal / square-root approximation within 1 ulp (what v_rcp_f32 / v_sqrt_f32 promise), the sequences return the correctly
rounded quotient / root -- i.e. what `/` and sqrtf() return, which is what the oracle computes.  No GPU involved: this pins
the arithmetic argument, the GPU parity tests pin the implementation."""
from fractions import Fraction
import math

import numpy as np


def rn32(x):
    """round-to-nearest-even of an exact rational to binary32 (normal range), as an exact Fraction"""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    assert Fraction(2) ** e <= a < Fraction(2) ** (e + 1) and -126 <= e <= 127
    q = a / Fraction(2) ** (e - 23)
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (n & 1)):
        n += 1
    return s * n * Fraction(2) ** (e - 23)


def fma(a, b, c):
    return rn32(a * b + c)


def ulp(x):
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    return Fraction(2) ** (e - 23)


def divisor_of(d, r0):
    e = fma(-d, r0, Fraction(1))
    return fma(e, r0, r0)


def div_unscaled(n, d, r):
    q = rn32(n * r)
    e = fma(-d, q, n)
    q = fma(e, r, q)
    e = fma(-d, q, n)
    return fma(e, r, q)


def sqrt_unscaled(x, s0):
    down, up = s0 - ulp(s0 - ulp(s0) / 4), s0 + ulp(s0)          # the neighbours of s0 (below a power of two the step halves)
    r_down, r_up = fma(-down, s0, x), fma(-up, s0, x)
    s = down if r_down <= 0 else s0
    return up if r_up > 0 else s


def f32(rng, lo_exp, hi_exp, n):
    m = rng.integers(1 << 23, 1 << 24, n)
    e = rng.integers(lo_exp, hi_exp + 1, n)
    return [Fraction(int(mi)) * Fraction(2) ** int(ei - 23) for mi, ei in zip(m, e)]


def test_quotients_by_a_refined_reciprocal_are_correctly_rounded():
    rng = np.random.default_rng(1)
    cases = []
    # roots of sphere_hit: a within 1e-5 of 1, numerators of any moderate size
    for n, k in zip(f32(rng, -20, 14, 1500), rng.integers(-80, 81, 1500)):
        cases.append((n * int(rng.choice([-1, 1])), Fraction(1) + int(k) * Fraction(2) ** -23 if k >= 0 else Fraction(1) + int(k) * Fraction(2) ** -24))
    # normalize / hit normal: |n| in [2^-90, 2^30], d in [2^-30, 2^30], |n / d| within the normal range
    for n, d in zip(f32(rng, -90, 29, 2500), f32(rng, -30, 29, 2500)):
        cases.append((n * int(rng.choice([-1, 1])), d * int(rng.choice([-1, 1]))))
    # unit sphere: components that are multiples of 2^-24 up to 1, lengths up to sqrt(3)
    for k, d in zip(rng.integers(1, 1 << 24, 1000), f32(rng, -24, 0, 1000)):
        cases.append((Fraction(int(k), 1 << 24), d))
    bad = 0
    for n, d in cases:
        exact = rn32(n / d)
        r_true = rn32(1 / d)
        for r0 in (r_true, r_true + ulp(r_true), r_true - ulp(r_true - ulp(r_true) / 4)):       # any v_rcp_f32 result within 1 ulp
            if div_unscaled(n, d, divisor_of(d, r0)) != exact:
                bad += 1
    assert bad == 0, f"{bad} of {3 * len(cases)} quotients differ from the correctly rounded one"
    # +0 / len stays +0 (the unit sphere's zero components)
    d = Fraction(3, 4)
    assert div_unscaled(Fraction(0), d, divisor_of(d, rn32(1 / d))) == 0


def test_square_roots_from_a_one_ulp_estimate_are_correctly_rounded():
    rng = np.random.default_rng(2)
    xs = f32(rng, -95, 60, 3000) + [Fraction(int(k), 1 << 24) for k in rng.integers(1, 1 << 24, 500)] + [Fraction(1), Fraction(4), Fraction(2) ** -48]
    bad = 0
    for x in xs:
        # the correctly rounded root: the f32 s with s - ulp/2 <= sqrt(x) <= s + ulp/2, found from the integer square root
        scale = 2 ** 200
        approx = Fraction(math.isqrt(int(x * scale * scale)), scale)
        s = rn32(approx)
        for cand in (s - ulp(s - ulp(s) / 4), s, s + ulp(s)):
            lo, hi = cand - ulp(cand - ulp(cand) / 4) / 2, cand + ulp(cand) / 2
            if (lo < 0 or lo * lo <= x) and x <= hi * hi:
                exact = cand
                break
        else:
            raise AssertionError(x)
        for s0 in (exact, exact + ulp(exact), exact - ulp(exact - ulp(exact) / 4)):            # any v_sqrt_f32 result within 1 ulp
            if s0 > 0 and sqrt_unscaled(x, s0) != exact:
                bad += 1
    assert bad == 0, f"{bad} roots differ from the correctly rounded one"
