#!/usr/bin/env python3
"""F4 (BASELINE configs[4]): 1 Mi problems x 50 fused steps, both fp32-state modes, timed; and bit-identity of the fused launch
against 50 single-step launches on a slice (the serial line search)."""
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
for dtype, tag in ((rp.DTYPE_F32_STATE, "fp32 state, fp64 arithmetic"), (rp.DTYPE_F32, "fp32 arithmetic"), (rp.DTYPE_F64, "fp64")):
    with rp.Batch(n, rp.VARIANT_F4, dtype) as b:
        ms = []
        for _ in range(4):
            b.set_problems_device(*ptrs)
            b.restart()
            b.sync()
            b.event_record(0)
            b.step(50)
            b.event_record(1)
            b.sync()
            ms.append(b.event_elapsed_ms(0, 1))
        fused = b.get_state_range(0, 65536)
        print("F4 %-28s 50 fused steps: %s ms -> %.2f G steps/s" % (tag, " ".join("%.3f" % m for m in ms), n * 50 / (min(ms) * 1e-3) / 1e9))
    m = 65536
    with rp.Batch(m, rp.VARIANT_F4, dtype) as c:
        c.set_problems(p[0][:m], p[1][:m], p[2][:m])
        for _ in range(50):
            c.step(1)
        single = c.get_state()
    # the 1 Mi batch and the 65,536 batch schedule their problems differently; compare per problem
    print("    fused vs 50 x step(1), first 65,536 problems: identical %s (max diff %.3e)" % (np.array_equal(fused, single), np.max(np.abs(fused - single))))
