#!/usr/bin/env python3
"""End-to-end (set_problems_device + fused solve) rate of 1 Mi fresh problems of each of the three distributions.
RP_BATCH_LIB selects the library, so two builds of schedule.hip can be compared in one gpurun call."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

n = 1 << 20
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
for dist, name in ((rp.problems.DIST_MONOTONE, "monotone"), (rp.problems.DIST_REFERENCE_LIKE, "reference-like"),
                   (rp.problems.DIST_NON_MONOTONE, "non-monotone")):
    p = rp.problems.generate(12345, 0, n, dist)
    d = torch.from_numpy(np.stack(p)).cuda()
    ptrs = [d[j].data_ptr() for j in range(3)]
    lead = rp.Batch(n)
    bs = [lead] + [rp.Batch(n, stream=lead.stream()) for _ in range(7)]
    best = 1e9
    for rep in range(4):
        lead.event_record(0)
        for b in bs:
            b.set_problems_device(*ptrs)
            b.solve(1e-8, 200, 0)
        lead.event_record(1)
        lead.sync()
        best = min(best, lead.event_elapsed_ms(0, 1) / len(bs))
    r = lead.reduce()
    print("%-15s %.1f us per batch, %d steps (mean %.2f), converged %d: %.1f G steps/s end to end"
          % (name, best * 1e3, r["total_steps"], r["total_steps"] / n, r["n_converged"], r["total_steps"] / best / 1e6))
    for b in bs:
        b.close()
