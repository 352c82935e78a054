#!/usr/bin/env python3
"""Kernel micro-bench used while tuning (GPU box): parity spot-check + timings of the three
launch shapes.  Not part of the driver contract; bench.py is."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "f3_batch.npz"))
n = len(g["init"])
with rp.Batch(n) as b:
    b.set_state(g["init"])
    b.solve(1e-8, 200, 0)
    it, st = b.get_iters()
    s = b.get_state()
    err = np.max(np.abs(s[:, :3] - g["gated"][:, :3]) / np.maximum(np.abs(g["gated"][:, :3]), 1.0))
    print("golden gated: iter mismatches %d, max err %.3e" % (int((it != g["iters"]).sum()), err))
    b.set_state(g["init"])
    b.step(50)
    s = b.get_state()
    err = np.max(np.abs(s[:, :3] - g["after50"][:, :3]) / np.maximum(np.abs(g["after50"][:, :3]), 1.0))
    print("golden fixed50: max err %.3e finite %s" % (err, bool(np.all(np.isfinite(s)))))

N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
reps = 6
batches = [rp.Batch(N) for _ in range(reps)]
stream_b = batches[0]
for b in batches:
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)


def timed(fn, tag, steps_of):
    ms = []
    for b in batches:
        b.event_record(0)
        fn(b)
        b.event_record(1)
        b.sync()
        ms.append(b.event_elapsed_ms(0, 1))
    steps = steps_of(batches[0])
    best, med = min(ms), sorted(ms)[len(ms) // 2]
    print("%-28s med %.4f ms  best %.4f ms  %.2f Gsteps/s (med)  alg %.0f GB/s" % (tag, med, best, steps / med / 1e6, steps * 216 / med / 1e6))


timed(lambda b: b.solve(1e-8, 200, 0), "gated fused 1M", lambda b: b.reduce()["total_steps"])
for b in batches:
    b.init_default()
timed(lambda b: b.solve(1e-8, 200, 0), "gated fused 1M identical", lambda b: b.reduce()["total_steps"])
for b in batches:
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
for b in batches:
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
timed(lambda b: b.step(1), "k=1 ungated 1M (step 1)", lambda b: N)
timed(lambda b: b.step(1), "k=1 ungated 1M (step 2)", lambda b: N)
os.environ["RP_STREAM_PROBE"] = "1"
for b in batches:
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
timed(lambda b: b.step(0), "k=0 probe (14 ld + 11 st)", lambda b: N)
timed(lambda b: b.step(0), "k=0 probe again", lambda b: N)
del os.environ["RP_STREAM_PROBE"]
# (the device-copy ceiling is measured in bench.py, where torch initialises its HIP runtime before this library does)
for b in batches:
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
timed(lambda b: b.step(12), "k=12 ungated 1M", lambda b: 12 * N)
for b in batches:
    b.close()
n2 = 65536
batches = [rp.Batch(n2) for _ in range(4)]
for b in batches:
    b.set_problems(p0[:n2], p1[:n2], p2[:n2])
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
timed(lambda b: b.step(50), "fixed50 65536", lambda b: 50 * n2)
for b in batches:
    b.set_problems(p0[:n2], p1[:n2], p2[:n2])
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
timed(lambda b: b.solve(1e-8, 200, 0), "gated fused 65536", lambda b: b.reduce()["total_steps"])
