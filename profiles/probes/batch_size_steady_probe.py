#!/usr/bin/env python3
"""Gated-solve throughput over the batch size at steady clocks (one stream): how much of a 1 Mi-problem launch its two ends cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rocket_path_amd as rp
for logn in (17, 18, 19, 20, 21, 22, 23):
    n = 1 << logn
    p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
    d_pos = torch.from_numpy(np.stack([p0, p1, p2])).cuda()
    ptrs = [d_pos[j].data_ptr() for j in range(3)]
    nb = max(8, min(640, (400 << 20) // n))            # ~70 ms of solves
    lead = rp.Batch(n)
    bs = [lead] + [rp.Batch(n, stream=lead.stream()) for _ in range(nb - 1)]
    res = []
    for rep in range(2):
        for b in bs:
            b.set_problems_device(*ptrs); b.restart()
        lead.sync()
        half = nb // 2
        for b in bs[:half]:
            b.solve(1e-8, 200, 0)                       # conditioning: the first half, untimed
        lead.event_record(0)
        for b in bs[half:]:
            b.solve(1e-8, 200, 0)
        lead.event_record(1)
        lead.sync()
        ms = lead.event_elapsed_ms(0, 1) / (nb - half)
        res.append(ms)
    steps = bs[-1].reduce()["total_steps"]
    print("n = 2^%d (%8d problems, %3d batches): %.4f ms per batch = %.2f G steps/s; per 1 Mi problems %.4f ms" % (
        logn, n, nb, min(res), steps / min(res) / 1e6, min(res) * (1 << 20) / n), flush=True)
    for b in bs:
        b.close()
    del d_pos
