"""The inequality behind AffineResidual::certain_failures (rocket_path_amd/csrc/ip_core.h), checked on the CPU.

In the reference's post-convergence regime the trial point has become x bit for bit and the residual vector is affine in the step
length: r_i(s) = c0_i + s c1_i.  The device counts, without evaluating them, the trials s, s/2, s/4, ... whose residual test
(onedpath_ip.cpp:941: accept iff |r(s)|^2 <= |r(0)|^2 (1 - 0.01 s)) fails BEYOND DOUBT, namely where
    h(s) = s (B + 0.01 R0) + C s^2 - 64 eps (R0 + s Babs + C s^2) > 0,   B = 2 S c0_i c1_i, Babs = 2 S |c0_i c1_i|, C = S c1_i^2,
and relies on (a) h(s) > 0 implying that the evaluated test fails, with room for the evaluation's rounding, and (b) "h(s 2^-k) > 0"
being true up to some k and false from there on (h(0) < 0, h convex), so that a bisection over k counts them.  Both are
properties of real arithmetic plus a rounding margin; here they are checked with numpy doubles and exact rational arithmetic
on random instances shaped like the regime (components at rounding level of their terms, directions of any size)."""
from fractions import Fraction

import numpy as np

EPS = 2.220446049250313e-16
MARGIN = 64 * EPS
ARMIJO = 0.01


def pieces(rng, m=11):
    scale = 10.0 ** rng.uniform(-14, 0)
    c0 = rng.normal(0, 1, m) * scale * 10.0 ** rng.uniform(-3, 0, m)
    c1 = rng.normal(0, 1, m) * scale * 10.0 ** rng.uniform(-6, 6)
    return c0, c1


def h_coefficients(c0, c1, r0):
    B = float(np.sum(c0 * c1))
    Babs = float(np.sum(np.abs(c0 * c1)))
    C = float(np.sum(c1 * c1))
    return -MARGIN * r0, ARMIJO * r0 + 2 * B - MARGIN * 2 * Babs, C * (1 - 2 * MARGIN)


def evaluated(c0, c1, s):
    r = c1 * s + c0
    return float(np.sum(r * r))


def test_certain_failures_are_failures_and_the_predicate_is_monotone():
    rng = np.random.RandomState(20261005)
    proven = checked = 0
    for _ in range(4000):
        c0, c1 = pieces(rng)
        r0 = evaluated(c0, c1, 0.0)
        h0, h1, h2 = h_coefficients(c0, c1, r0)
        s0 = 0.99 * 2.0 ** -rng.randint(0, 30)
        holds = []
        for k in range(110):
            s = s0 * 2.0 ** -k
            hk = (h2 * s + h1) * s + h0
            holds.append(hk > 0)
            if hk > 0:
                # (a) the evaluated test fails, and not by a rounding: exactly, R(s) - R(0) (1 - 0.01 s) > 13 u (R(s) + R(0))
                R = sum((Fraction(float(a)) + Fraction(s) * Fraction(float(b))) ** 2 for a, b in zip(c0, c1))
                R0 = sum(Fraction(float(a)) ** 2 for a in c0)
                gap = R - R0 * (1 - Fraction(ARMIJO) * Fraction(s))
                assert gap > Fraction(14 * EPS / 2) * (R + R0), (k, float(gap))
                assert evaluated(c0, c1, s) > r0 * (1.0 - ARMIJO * s)
                proven += 1
            checked += 1
        # (b) true up to some k, false from there on
        first_false = holds.index(False) if False in holds else len(holds)
        assert not any(holds[first_false:]), holds
    assert proven > 20000 and checked == 4000 * 110


def test_the_count_leaves_only_the_last_few_halvings_to_evaluate():
    # what the proof buys: of the halvings the reference makes before its test passes (s r' below the rounding of r), all but a
    # few are certain failures
    rng = np.random.RandomState(7)
    left = []
    for _ in range(300):
        c0, c1 = pieces(rng)
        r0 = evaluated(c0, c1, 0.0)
        h0, h1, h2 = h_coefficients(c0, c1, r0)
        s0 = 0.99
        k_accept = next((k for k in range(400) if evaluated(c0, c1, s0 * 2.0 ** -k) <= r0 * (1.0 - ARMIJO * s0 * 2.0 ** -k)), None)
        if k_accept is None or k_accept < 20:
            continue
        k_proven = next((k for k in range(400) if not ((h2 * (s0 * 2.0 ** -k) + h1) * (s0 * 2.0 ** -k) + h0 > 0)), 400)
        assert k_proven <= k_accept
        left.append(k_accept - k_proven)
    assert len(left) > 100 and np.median(left) <= 12, (len(left), np.median(left))
