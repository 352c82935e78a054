#!/usr/bin/env python3
"""The scheduling pass alone (rp_batch_set_problems_device) and a fresh batch end to end on one stream, HIP-event timed, 1 Mi problems."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
from hip_util import DeviceBuffer
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
N = 1 << 20
q = rp.problems.generate(12345, 0, N, 0)
pos = DeviceBuffer(3 * 8 * N); pos.write(np.stack(q))
ptrs = (pos.ptr, pos.ptr + 8 * N, pos.ptr + 16 * N)
lead = rp.Batch(N); bs = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(39)]
for b in bs: b.set_problems_device(*ptrs); b.solve(1e-8, 200, 0)
lead.sync()
for rnd in range(3):
    lead.event_record(0)
    for b in bs: b.set_problems_device(*ptrs)
    lead.event_record(1); lead.sync()
    sched = lead.event_elapsed_ms(0, 1) / len(bs)
    lead.event_record(0)
    for b in bs: b.set_problems_device(*ptrs); b.solve(1e-8, 200, 0)
    lead.event_record(1); lead.sync()
    e2e = lead.event_elapsed_ms(0, 1) / len(bs)
    print("scheduling pass %.2f us;  positions -> solved batch on one stream %.2f us" % (sched * 1e3, e2e * 1e3), flush=True)
