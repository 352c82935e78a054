#!/usr/bin/env python3
"""Gated F3 solves from states the reference's keys and set_state can produce: which problems run into the step cap, and are they bitwise
fixed points of the step (starts outside the feasible set: 100 feasibility halvings, no movement)?  (CPU only; profiles/r6_gated_fixed_points.log --
the measurement behind the fixed-point watch of k_solve_chunks<ROUNDS>.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp  # noqa: E402
from oracle_api import Oracle, StepInfo  # noqa: E402

o = Oracle()
N = 1 << 14
rng = np.random.default_rng(5)
fams = {}
for dist, dn in ((0, "monotone"), (2, "non-monotone")):
    p0, p1, p2 = rp.problems.generate(12345, 0, N, dist)
    base = o.batch_init_feasible(3, p0, p1, p2)
    s = base.copy(); s[:, 0] = rng.uniform(-10, 10, N); fams[dn + " vel1 nudged"] = s
    s = base.copy(); s[:, 1] += 1.0; s[:, 2] += 1.0; fams[dn + " durations +1"] = s
    s = base.copy(); s[:, 3:11] = 10.0 ** rng.uniform(-3, 2, (N, 1)); fams[dn + " multipliers per problem"] = s
for name, st0 in fams.items():
    st = st0.copy()
    it = np.asarray(o.batch_solve_gated(3, st, 1e-8, 200)[0])
    hit = it >= 200
    if hit.sum() == 0:
        print("%-40s none reaches the cap (max %d)" % (name, it.max())); continue
    # walk those problems step by step: when does the state freeze bitwise?
    sub = st0[hit].copy()
    first_frozen = np.full(len(sub), -1)
    info = StepInfo()
    fh = np.zeros(len(sub)); rh = np.zeros(len(sub))
    for step in range(1, 61):
        new = sub.copy()
        o.batch_steps(3, new, 1)
        same = (new.view(np.int64) == sub.view(np.int64)).all(axis=1)
        first_frozen = np.where((first_frozen < 0) & same, step, first_frozen)
        sub = new
    # halvings at step 60 for a sample
    for i in range(min(len(sub), 200)):
        row = sub[i].copy(); o.step(3, row, info); fh[i] = info.feas_halvings; rh[i] = info.resid_halvings
    k = min(len(sub), 200)
    print("%-40s reach the cap: %5d of %d (%.2f%%); bitwise frozen within 60 steps: %5d (first at step: median %s); at step 61: feas halvings mean %.1f, resid halvings mean %.1f; satisfied %d/%d" % (
        name, hit.sum(), N, 100.0 * hit.mean(), (first_frozen > 0).sum(), np.median(first_frozen[first_frozen > 0]) if (first_frozen > 0).any() else None,
        fh[:k].mean(), rh[:k].mean(), sum(o.satisfied(3, sub[i]) for i in range(k)), k))
