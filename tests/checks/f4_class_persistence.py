#!/usr/bin/env python3
"""Is a problem's F4 feasibility-halving class (none / ~a dozen per step) persistent from step to step, and do the residual-loop
stragglers belong to one class?  (Decides whether regrouping by that class could pay.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
n = 65536
p = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as b:
    b.set_problems(*p)
    prev = None
    for step in range(40):
        nf, nr = b.step_counted(1)
        if prev is not None and step in (8, 12, 20, 30, 39):
            z0, z1 = prev == 0, nf == 0
            print("step %2d: P(0 now | 0 before) %.3f  P(>0 now | >0 before) %.3f  share of 0-halvers %.3f | stragglers (resid > 20): %d, of them in the 0-class %.3f; |nf - prev| mean among >0: %.2f" % (
                step, (z0 & z1).sum() / max(z0.sum(), 1), (~z0 & ~z1).sum() / max((~z0).sum(), 1), z1.mean(), (nr > 20).sum(),
                ((nr > 20) & z1).sum() / max((nr > 20).sum(), 1), np.abs(nf[~z0 & ~z1].astype(int) - prev[~z0 & ~z1].astype(int)).mean()))
        prev = nf
