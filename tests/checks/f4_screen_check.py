#!/usr/bin/env python3
"""F4's feasibility loop on the device (closed-form ray proof + per-trial proof + evaluation proper, ip_core.h) against the
oracle's loop, decision for decision: 40 steps of 4,096 problems, the device re-started from the ORACLE's state before every
step (F4 trajectories are chaotic), both halving counts of every problem compared.  GPU box:
    python tests/checks/f4_screen_check.py > gpurun_out/f4_screen_check.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp  # noqa: E402
from oracle_api import Oracle, StepInfo  # noqa: E402

o = Oracle()
n, steps = 4096, 40
for dtype, tag in ((rp.DTYPE_F64, "fp64"), (rp.DTYPE_F32_STATE, "fp32 state, fp64 arithmetic")):
    p0, p1, p2 = rp.problems.generate(31415, 0, n, rp.problems.DIST_MONOTONE)
    aos = o.batch_init_feasible(rp.VARIANT_F4, p0, p1, p2)
    info = StepInfo()
    tot_f = tot_r = bad_f = bad_r = long_seq = 0
    worst = 0
    with rp.Batch(n, rp.VARIANT_F4, dtype) as b:
        for s in range(steps):
            if dtype != rp.DTYPE_F64:
                aos[:] = aos.astype(np.float32).astype(np.float64)      # what the batch will hold (state and constants)
            b.set_state(aos)
            nf, nr = b.step_counted(1)
            for i in range(n):
                o.step(rp.VARIANT_F4, aos[i], info)
                tot_f += info.feas_halvings
                tot_r += info.resid_halvings
                bad_f += int(nf[i] != info.feas_halvings)
                bad_r += int(nr[i] != info.resid_halvings)
                long_seq += int(info.feas_halvings >= 10)
                worst = max(worst, info.feas_halvings)
    print("F4 %-28s %d problem-steps: %d feasibility halvings (%d sequences of 10 or more, longest %d), %d residual halvings; "
          "feasibility counts that differ: %d, residual counts that differ: %d" % (tag, n * steps, tot_f, long_seq, worst, tot_r, bad_f, bad_r))
