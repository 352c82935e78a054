#!/usr/bin/env python3
"""Post-convergence regime of a fixed-step F3 run: per step, how many residual trials still need a full evaluation (the trial point
moves) against the halvings made (library built with -DRP_DIAG_MOVING: step_counted's first array counts full evaluations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
n = 65536
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n) as b:
    b.set_problems(p0, p1, p2)
    done = 0
    for upto in (10, 16, 18, 20, 22, 25, 30, 40, 49):
        b.step(upto - done); done = upto
        nm, nr = b.step_counted(1); done += 1
        w = nm.reshape(-1, 64).max(axis=1); wr = nr.reshape(-1, 64).max(axis=1)
        print("step %2d: full evaluations per problem mean %.1f max %d (per wave: mean of max %.1f); residual halvings mean %.1f max %d (per wave %.1f)"
              % (done, nm.mean(), nm.max(), w.mean(), nr.mean(), nr.max(), wr.mean()))
