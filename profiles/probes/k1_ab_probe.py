#!/usr/bin/env python3
"""k = 1 launch (one Newton step per launch, the HBM-streaming form), 1 Mi F3 fp64 problems, cold batches: ms and TB/s on the 200 B that move."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

n = 1 << 20
p = rp.problems.generate(12345, 0, n, 0)
d = torch.from_numpy(np.stack(p)).cuda()
ptrs = [d[j].data_ptr() for j in range(3)]
lead = rp.Batch(n)
bs = [lead] + [rp.Batch(n, stream=lead.stream()) for _ in range(15)]
for k in (1, 2):
    for rep in range(3):
        for b in bs:
            b.set_problems_device(*ptrs)
            b.restart()
        lead.sync()
        lead.event_record(0)
        for b in bs:
            b.step(k)
        lead.event_record(1)
        lead.sync()
        ms = lead.event_elapsed_ms(0, 1) / len(bs)
        print("k = %d: %.4f ms per launch, %.2f TB/s on 200 B per problem per launch, %.2f G steps/s" % (k, ms, 200.0 * n / ms / 1e9, k * n / ms / 1e6))
