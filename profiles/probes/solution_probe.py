#!/usr/bin/env python3
"""rp_batch_solution_device / bound solution buffer, 1 Mi solved problems: time per pass (HIP events).  A/B across builds with RP_BATCH_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
from hip_util import DeviceBuffer
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
n = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n) as b, DeviceBuffer(32 * n) as out:
    b.set_problems(p0, p1, p2)
    b.solve(1e-8, 200, 0)
    for _ in range(3):
        b.solution_device(out.ptr)
    b.sync()
    b.event_record(0)
    for _ in range(20):
        b.solution_device(out.ptr)
    b.event_record(1)
    b.sync()
    print("solution_device: %.4f ms per pass" % (b.event_elapsed_ms(0, 1) / 20))
    for bound in (False, True):
        b.bind_solution(out.ptr if bound else None)
        ts = []
        for _ in range(12):
            b.set_problems(p0, p1, p2)
            b.sync()
            b.event_record(0)
            b.solve(1e-8, 200, 0)
            b.event_record(1)
            b.sync()
            ts.append(b.event_elapsed_ms(0, 1))
        print("fused START solve, solution buffer %s: min %.4f median %.4f ms" % ("bound" if bound else "not bound", min(ts), float(np.median(ts))))
