#!/usr/bin/env python3
"""F4 with its state rounded to fp32 after every step (the oracle's fp64 step): which problems does a step leave bit for bit unchanged,
and does one ever move again?  (CPU only; profiles/r6_f4_fixed_points.log -- the measurement behind run_lane's PARK, ip_kernels.hip.)"""
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
p0, p1, p2 = rp.problems.generate(12345, 0, N, rp.problems.DIST_MONOTONE)
st = o.batch_init_feasible(4, p0, p1, p2)
st = st.astype(np.float32).astype(np.float64)
parked = np.zeros(N, bool)
for step in range(1, 51):
    new = st.copy()
    o.batch_steps(4, new, 1)
    new = new.astype(np.float32).astype(np.float64)
    same = (new.view(np.int64) == st.view(np.int64)).all(axis=1)
    # fixed points are permanent?
    lost = parked & ~same
    parked = same
    if step % 3 == 0 or step < 8:
        print("step %2d: unchanged after rounding %6d (%.2f%%)  per 64-wave: %.2f  waves with >=1: %.2f   un-parked: %d" % (
            step, same.sum(), 100.0 * same.mean(), same.reshape(-1, 64).sum(1).mean(), (same.reshape(-1, 64).sum(1) > 0).mean(), lost.sum()))
    st = new
