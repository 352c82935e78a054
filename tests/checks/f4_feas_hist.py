#!/usr/bin/env python3
"""Distribution of F4's feasibility halvings per step (what a lock-step wave pays the maximum of)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
n = 65536
p = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as b:
    b.set_problems(*p)
    slot = b.slot_map()
    for step in range(40):
        nf, nr = b.step_counted(1)
        if step in (10, 20, 30, 39):
            order = np.argsort(slot)                  # problems in batch order: waves are 64 consecutive positions
            w = nf[order][: n // 64 * 64].reshape(-1, 64)
            srt = np.sort(w, axis=1)
            print("step %2d: feas halvings mean %.2f; per wave: max %.1f, 2nd %.1f, 4th %.1f, 8th %.1f, 16th %.1f largest; hist %s" % (
                step, nf.mean(), srt[:, -1].mean(), srt[:, -2].mean(), srt[:, -4].mean(), srt[:, -8].mean(), srt[:, -16].mean(),
                np.bincount(np.minimum(nf, 25))[:26].tolist()))
