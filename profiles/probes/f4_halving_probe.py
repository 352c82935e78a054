#!/usr/bin/env python3
"""F4 from the feasible start: per-step feasibility / residual halvings, per lane and per 64-lane wave (the wave pays the maximum)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 8192
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
with rp.Batch(N, rp.VARIANT_F4) as b:
    b.set_problems(p0, p1, p2)
    tf = tr = wf = wr = 0
    for s in range(50):
        nf, nr = b.step_counted(1)
        wfm = nf.reshape(-1, 64).max(axis=1).mean(); wrm = nr.reshape(-1, 64).max(axis=1).mean()
        tf += nf.mean(); tr += nr.mean(); wf += wfm; wr += wrm
        if s % 3 == 0 or s < 8:
            print("step %2d feas mean %5.2f wave-max %5.2f | resid mean %5.2f wave-max %5.2f max %d" % (s, nf.mean(), wfm, nr.mean(), wrm, nr.max()))
    print("sum over 50 steps: feas lane %.0f wave %.0f | resid lane %.0f wave %.0f" % (tf, wf, tr, wr))
