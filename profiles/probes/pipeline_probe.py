#!/usr/bin/env python3
"""rp_pipeline arrangements (streams, where the scheduling pass runs) at steady clocks: ms per job of "positions in -> solutions out" at
1 Mi problems, wall clock over a burst of jobs after a conditioning burst; the one-stream solve-only figure measured the same way beside it."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP initialises: streams that share one of the default 4 hardware queues serialise (profiles/r6_hw_queues.log)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
from hip_util import DeviceBuffer
N = int(os.environ.get("N", 1 << 20)); JOBS = 120
q = rp.problems.generate(12345, 0, N, 0)
# COLD=1 (default): every job reads its positions from a buffer of its own -- 16 of them, 400 MB, more than the 256 MB Infinity Cache holds -- as a
# caller's fresh inputs would arrive; COLD=0: all jobs read ONE 24 MB buffer (cache-resident after the first job: flatters the scheduling pass)
COLD = os.environ.get("COLD", "1") != "0"
NPOS = 16 if COLD else 1
poss = []
for _ in range(NPOS):
    d = DeviceBuffer(3 * 8 * N); d.write(np.stack(q)); poss.append(d)
outs = [DeviceBuffer(32 * N) for _ in range(8)]
ptrs = (poss[0].ptr, poss[0].ptr + 8 * N, poss[0].ptr + 16 * N)
print("positions:", "a buffer per job (16 x 24 MB, cold)" if COLD else "one 24 MB buffer for all jobs (cache-resident)")

def burst(pipe, jobs):
    for j in range(jobs):
        d = poss[j % NPOS]
        pipe.submit(d.ptr, d.ptr + 8 * N, d.ptr + 16 * N, d_out=outs[j % 8].ptr)
    pipe.wait()

res = {}
for rnd in range(3):
    for name, kw in (("1 stream, inline", dict(depth=2, n_streams=1)), ("2 streams, inline (one-wave sched blocks)", dict(depth=4, n_streams=2)),
                     ("2 streams, inline, 256-thread sched", dict(depth=4, n_streams=2, prep=16)),
                     ("2 streams, prep stream", dict(depth=4, n_streams=2, prep=1)), ("2 streams, prep stream, 256-thread sched", dict(depth=4, n_streams=2, prep=17)),
                     ("3 streams, inline", dict(depth=6, n_streams=3))):
        with rp.Pipeline(N, **kw) as pipe:
            burst(pipe, 160)                     # conditioning + first-use allocations
            t = time.perf_counter(); burst(pipe, JOBS); dt = (time.perf_counter() - t) / JOBS * 1e3
            tot = pipe.batch(JOBS + 160 - 1).reduce()["total_steps"]
        res.setdefault(name, []).append(dt)
        print("round %d  %-40s %.4f ms per job = %.2f G steps/s" % (rnd, name, dt, tot / dt / 1e6), flush=True)
    # solve only, one stream and two: pre-armed batches (restart outside the clock)
    for ns in (1, 2):
        heads = [rp.Batch(N) for _ in range(ns)]
        pool = heads + [rp.Batch(N, stream=heads[j % ns].stream()) for j in range(40 - ns)]
        for b in pool: b.set_problems_device(*ptrs)
        best = None
        for rep in range(3):
            for b in pool: b.restart()
            for b in heads: b.sync()
            t = time.perf_counter()
            for b in pool: b.solve(1e-8, 200, 0)
            for b in heads: b.sync()
            dt = (time.perf_counter() - t) / len(pool) * 1e3
            best = dt if best is None else min(best, dt)
        print("round %d  %-40s %.4f ms per solve = %.2f G steps/s" % (rnd, "solve only, %d stream(s)" % ns, best, tot / best / 1e6), flush=True)
        for b in pool[::-1]: b.close()
print("best:", {k: round(min(v), 4) for k, v in res.items()})
