"""Shared parity checks of the GPU tests.

Iteration counts of a gated solve must equal the oracle's.  The one exception that any two correct
implementations must be allowed is a tie at the gate: the gate compares the surrogate gap, a sum of products
lam_i * c_i with c_i = a - L cancelling to ~1e-7 near convergence, against the tolerance, so the gap itself carries
a relative rounding noise of ~1e-8 by the last steps.  A problem whose ORACLE gap at the decisive check lies within
GATE_TIE (relative) of the tolerance is decided by that noise, not by the algorithm (seed 777, reference-like
distribution, problem 260: gap 1.0000000330e-08 against 1e-8 at step 17).  Such a problem may differ by exactly
one step; every other difference fails the test.  Expected frequency: about one problem per million."""
import numpy as np

GATE_TIE = 1e-6


def certify_iteration_counts(oracle, variant, init_rows, it_gpu, it_ref, tol, max_ties=2):
    """Asserts it_gpu == it_ref except for certified gate ties; returns the indices of the ties."""
    it_gpu, it_ref = np.asarray(it_gpu), np.asarray(it_ref)
    bad = np.nonzero(it_gpu != it_ref)[0]
    assert len(bad) <= max_ties, "%d iteration-count mismatches" % len(bad)
    for i in bad:
        assert abs(int(it_gpu[i]) - int(it_ref[i])) == 1, (i, it_gpu[i], it_ref[i])
        v = np.array(init_rows[i], dtype=np.float64)
        for _ in range(min(int(it_gpu[i]), int(it_ref[i]))):
            oracle.step(variant, v)
        gap = oracle.gap(variant, v)
        assert abs(gap / tol - 1.0) < GATE_TIE, "problem %d: %d vs %d steps with the oracle's gap %.17g not at the gate" % (
            i, it_gpu[i], it_ref[i], gap)
    return bad


def keep_mask(n, ties):
    m = np.ones(n, dtype=bool)
    m[np.asarray(ties, dtype=int)] = False
    return m


# ---- line-search decisions that differ from the oracle's: certified, not budgeted ---------------------------------------
# Both backtracking loops compare computed quantities (onedpath_ip.cpp:919-928: constraint values against 0; :932-945:
# |r(x + s d)|^2 against |r(x)|^2 (1 - 0.01 s)).  Two correct implementations that round differently (the device condenses the
# KKT system, takes reciprocals and evaluates with fused multiply-adds) may decide a trial differently ONLY where the two sides
# lie within rounding of each other.  "Within rounding" is made checkable by the oracle itself (orc_armijo_sides,
# orc_feasibility_margin), which reports next to the quantity how much it changes
#   (a) when ONE coordinate of the point it is evaluated at moves by ONE ulp (its "spread": a lower estimate of the evaluation's
#       own rounding noise, which accumulates a dozen such roundings), and
#   (b) when the trial point is formed with the direction of a SECOND backward-stable solver of the same KKT system (Gaussian
#       elimination with partial pivoting beside the reference's column-pivoted QR): how far the direction itself is determined
#       in double precision -- past convergence the right-hand side is rounding noise and so is the direction (problem 519 of the
#       non-monotone test set, step 24: the two solvers move a constraint value by 2.4e-12 where a one-ulp move makes 5.7e-14).
# The rules:
#   feasibility: at the first trial the two sides decide differently, the oracle's largest constraint value lies within
#                FEAS_TIE x ((a) + (b)) of zero;
#   residual:    at the first trial decided differently, the oracle's |r(trial)|^2 and |r(x)|^2 (1 - 0.01 s) differ by at most
#                RESID_TIE x ((a) of the one + (a) of the other + (b)) -- BOTH sides are evaluations: past convergence they are
#                sums of squares of pure rounding noise (1e-31), which no count of ulps of the value describes -- and never by
#                less than RESID_TIE_ULPS ulps of the value (a trial point that has become x bit for bit may sit where no
#                single coordinate move registers).
# Every differing decision is checked; one that is not a tie fails the test.  Calibration (tests/checks/decision_margins.py,
# profiles/r5_decision_margins.log: 3.4 M problem-steps -- F4 steps 19-28 in two number modes; F3 steps 1-50 / 1-30 on the three
# distributions, sixteen seeds -- 988,000 differing feasibility decisions, 314,000 differing residual decisions): the worst feasibility
# tie sits at 0.5 of its allowance, the worst residual tie at 0.65.
FEAS_TIE = 4.0
RESID_TIE = 8.0
RESID_TIE_ULPS = 8.0


def certify_line_search_decisions(oracle, variant, states_before, nf_gpu, nr_gpu, nf_ref, nr_ref, strict=True):
    """states_before: the states the compared step started from (rows, the oracle's layout).  Returns a dict with the number
    of differing decisions of each loop and the certified distances (in units of the allowance: <= 1 is a tie).  strict=False
    (tests/checks/decision_margins.py, the calibration run) collects the distances without asserting."""
    nf_gpu, nr_gpu, nf_ref, nr_ref = (np.asarray(x, dtype=np.int64) for x in (nf_gpu, nr_gpu, nf_ref, nr_ref))
    out = {"feas_diffs": 0, "resid_diffs": 0, "worst_feas": 0.0, "worst_resid": 0.0, "feas_ratios": [], "resid_ratios": []}
    for i in np.nonzero((nf_gpu != nf_ref) | (nr_gpu != nr_ref))[0]:
        row = np.ascontiguousarray(states_before[i], dtype=np.float64)
        if nf_gpu[i] != nf_ref[i]:
            # the first feasibility trial decided differently: its largest constraint value must sit within rounding of zero
            # (after it the two searches walk different step lengths: the residual counts are then not comparable)
            h = int(min(nf_gpu[i], nf_ref[i]))
            m = oracle.feasibility_margin(variant, row, h)
            assert m is not None, (i, nf_gpu[i], nf_ref[i])
            worst, s, spread, dir_spread = m
            ratio = abs(worst) / (FEAS_TIE * (spread + dir_spread)) if spread + dir_spread > 0 else np.inf
            assert ratio <= 1.0 or not strict, \
                "problem %d: feasibility halvings %d vs %d, but the oracle's largest constraint value at trial %d is %.3g, " \
                "%.1f x the allowance (one-ulp spread %.3g, second-solver spread %.3g)" % (i, nf_gpu[i], nf_ref[i], h, worst, ratio, spread, dir_spread)
            out["feas_diffs"] += 1
            out["feas_ratios"].append(ratio)
            continue
        h = int(min(nr_gpu[i], nr_ref[i]))
        m = oracle.armijo_sides(variant, row, h)
        assert m is not None, (i, nr_gpu[i], nr_ref[i])
        lhs, rhs, s, spread, spread_rhs, dir_spread = m
        allow = max(RESID_TIE * (spread + spread_rhs + dir_spread), RESID_TIE_ULPS * np.spacing(abs(lhs)))
        ratio = abs(lhs - rhs) / allow
        assert ratio <= 1.0 or not strict, \
            "problem %d: residual halvings %d vs %d, but at trial %d the oracle has |r(trial)|^2 = %.17g against %.17g: " \
            "%.1f x the allowance (one-ulp spreads %.3g + %.3g, second-solver spread %.3g)" % (i, nr_gpu[i], nr_ref[i], h, lhs, rhs, ratio, spread, spread_rhs, dir_spread)
        out["resid_diffs"] += 1
        out["resid_ratios"].append(ratio)
    out["worst_feas"] = max(out["feas_ratios"], default=0.0)
    out["worst_resid"] = max(out["resid_ratios"], default=0.0)
    return out
