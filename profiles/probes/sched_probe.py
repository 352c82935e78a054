#!/usr/bin/env python3
"""Time rp_batch_set_problems_device (the scheduling pass) alone and followed by the fused solve, 1 Mi problems.
(The scatter variants listed at the end of profiles/r3_sched_probe.log were tuning builds of schedule.hip measured with this script.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

n = 1 << 20
p = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
d = torch.from_numpy(np.stack(p)).cuda()
ptrs = [d[j].data_ptr() for j in range(3)]
lead = rp.Batch(n)
bs = [lead] + [rp.Batch(n, stream=lead.stream()) for _ in range(15)]
for rep in range(3):
    lead.event_record(0)
    for b in bs:
        b.set_problems_device(*ptrs)
    lead.event_record(1)
    lead.sync()
    t_s = lead.event_elapsed_ms(0, 1) / len(bs)
    print("set_problems_device alone: %.1f us per batch" % (t_s * 1e3))
if True:
    for rep in range(3):
        lead.event_record(0)
        for b in bs:
            b.set_problems_device(*ptrs)
            b.solve(1e-8, 200, 0)
        lead.event_record(1)
        lead.sync()
        print("set_problems_device + fused solve: %.1f us per batch" % (lead.event_elapsed_ms(0, 1) / len(bs) * 1e3))
