#!/usr/bin/env python3
"""Calibration run behind tests/parity_util.py's DECISION_TIE (GPU box): every line-search decision of the device that differs from the
oracle's, with its distance from the threshold in units of the allowance -- F4 steps 18..27 (double, fp32 state), F3 steps 1..50, each
step re-started from the ORACLE's state.  Prints the distribution; asserts nothing.   python3 tests/checks/decision_margins.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp
from oracle_api import Oracle, StepInfo
from parity_util import certify_line_search_decisions, FEAS_TIE, RESID_TIE, RESID_TIE_ULPS

oracle = Oracle()
info = StepInfo()
print("FEAS_TIE = %g x (one-ulp spread + second-solver spread); RESID_TIE = %g x (spread of |r(trial)|^2 + spread of the threshold + second-solver "
      "spread), never below %g ulps" % (FEAS_TIE, RESID_TIE, RESID_TIE_ULPS))
SEEDS = tuple(int(x) for x in os.environ.get("SEEDS", "31415,27182,2718,16180").split(","))      # (default: the tests' own seeds among them; needs >= 4)
for variant, dtype, first, steps, n, dist, seed in [(4, rp.DTYPE_F64, 18, 10, 4096, 0, sd) for sd in SEEDS[:2]] + [(4, rp.DTYPE_F32_STATE, 18, 10, 4096, 0, SEEDS[0]),
                                                    (4, rp.DTYPE_F64, 24, 1, 2048, 0, 2718)] + \
        [(3, rp.DTYPE_F64, 0, 50, 2048, 0, sd) for sd in SEEDS] + [(3, rp.DTYPE_F64, 0, 30, 2048, 2, sd) for sd in SEEDS] + \
        [(3, rp.DTYPE_F64, 0, 30, 2048, 1, SEEDS[0])]:
    p0, p1, p2 = rp.problems.generate(seed, 0, n, dist)
    aos = oracle.batch_init_feasible(variant, p0, p1, p2)
    oracle.batch_steps(variant, aos, first, threads=0)
    fr, rr, nd = [], [], 0
    per_step = []
    with rp.Batch(n, variant, dtype) as a:
        for s in range(steps):
            if dtype != rp.DTYPE_F64:
                aos[:] = aos.astype(np.float32).astype(np.float64)
            before = aos.copy()
            a.set_state(aos)
            nf, nr = a.step_counted(1)
            of, orr = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
            for i in range(n):
                oracle.step(variant, aos[i], info)
                of[i], orr[i] = info.feas_halvings, info.resid_halvings
            c = certify_line_search_decisions(oracle, variant, before, nf, nr, of, orr, strict=False)
            fr += c["feas_ratios"]; rr += c["resid_ratios"]
            per_step.append((first + s + 1, c["feas_diffs"], c["resid_diffs"]))
    q = lambda x: "none" if not x else "n %d median %.3g q90 %.3g q99 %.3g max %.3g (above 1: %d)" % (
        len(x), np.median(x), np.quantile(x, .9), np.quantile(x, .99), max(x), int((np.array(x) > 1).sum()))
    print("variant %d dtype %d dist %d seed %d, %d problem-steps from step %d: feasibility differences: %s; residual differences: %s" % (
        variant, dtype, dist, seed, n * steps, first + 1, q(fr), q(rr)), flush=True)
    print("   per step (step, feas, resid):", " ".join("%d:%d/%d" % t for t in per_step if t[1] or t[2]), flush=True)
