#!/usr/bin/env python3
"""How long does the power controller's transient last?  From an idle chip: NB gated 1 Mi-problem solves back to back on one stream,
a HIP event every EVERY launches -> ms per launch over time.  (bench.py's default timed region used to sit inside this transient.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rocket_path_amd as rp
NB, EVERY = int(os.environ.get("NB", "640")), int(os.environ.get("EVERY", "8"))
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
d_pos = torch.from_numpy(np.stack([p0, p1, p2])).cuda()
ptrs = [d_pos[j].data_ptr() for j in range(3)]
lead = rp.Batch(N)
bs = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(NB - 1)]
ext = torch.cuda.ExternalStream(lead.stream())
for rep in range(2):
    for b in bs:
        b.set_problems_device(*ptrs); b.restart()
    lead.sync()
    time.sleep(1.0 if rep == 0 else 0.05)          # rep 0: from a chip that has idled for a second; rep 1: 50 ms after the previous burst
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(NB // EVERY + 1)]
    evs[0].record(ext)
    for j, b in enumerate(bs):
        b.solve(1e-8, 200, 0)
        if (j + 1) % EVERY == 0:
            evs[(j + 1) // EVERY].record(ext)
    lead.sync()
    ms = [evs[i].elapsed_time(evs[i + 1]) / EVERY for i in range(len(evs) - 1)]
    t = np.cumsum([0.0] + [m * EVERY for m in ms])
    print("burst %d (%s): ms per launch over groups of %d launches" % (rep, "after 1 s idle" if rep == 0 else "50 ms after the previous burst", EVERY))
    print("  t[ms]: " + " ".join("%6.1f" % x for x in t[:-1]))
    print("  ms   : " + " ".join("%6.4f" % x for x in ms))
    print("  first 20 launches %.4f, launches 21-40 %.4f, 41-100 %.4f, 101-200 %.4f, last 200 %.4f ms" % (
        np.mean(ms[:20 // EVERY + 1]), np.mean(ms[20 // EVERY:40 // EVERY]), np.mean(ms[40 // EVERY:100 // EVERY]), np.mean(ms[100 // EVERY:200 // EVERY]), np.mean(ms[-200 // EVERY:])), flush=True)
