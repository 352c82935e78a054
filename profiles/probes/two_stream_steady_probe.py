#!/usr/bin/env python3
"""At steady clocks: NB gated 1 Mi solves back to back on ONE stream against the same batches dealt alternately onto TWO streams (the next
launch's first chunks fill the previous launch's drain).  Wall clock over the whole burst, after a conditioning burst."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rocket_path_amd as rp
NB = int(os.environ.get("NB", "400"))
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
d_pos = torch.from_numpy(np.stack([p0, p1, p2])).cuda()
ptrs = [d_pos[j].data_ptr() for j in range(3)]
a = rp.Batch(N); b = rp.Batch(N)
one = [a] + [rp.Batch(N, stream=a.stream()) for _ in range(NB - 1)]
two = [a if j % 2 == 0 else b for j in range(2)] + [rp.Batch(N, stream=(a if j % 2 == 0 else b).stream()) for j in range(2, NB)]
def run(bs, label):
    for x in bs:
        x.set_problems_device(*ptrs); x.restart()
    a.sync(); b.sync()
    for x in bs[:160]:                      # conditioning: steady clocks
        x.solve(1e-8, 200, 0)
    for x in bs[:160]:
        x.restart()
    a.sync(); b.sync()
    # (the restart phase above is 3 ms of memory-bound work: a second, shorter conditioning right before the clock starts)
    for x in bs[160:260]:
        x.solve(1e-8, 200, 0)
    a.sync(); b.sync()
    t0 = time.perf_counter()
    for x in bs[:160]:
        x.solve(1e-8, 200, 0)
    a.sync(); b.sync()
    dt = time.perf_counter() - t0
    steps = bs[0].reduce()["total_steps"]
    print("%-28s 160 solves in %.3f ms = %.4f ms per batch = %.2f G steps/s" % (label, dt * 1e3, dt * 1e3 / 160, steps * 160 / dt / 1e9), flush=True)
for rep in range(3):
    run(one, "one stream")
    run(two, "two streams, alternating")
