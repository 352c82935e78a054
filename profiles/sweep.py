#!/usr/bin/env python3
"""Launch-time sweep over batch sizes and launch shapes (F3 fp64; one MI355X).  Not part of the driver contract."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp

def best(b, fn, reinit, reps=4):
    ms = []
    for _ in range(reps):
        reinit(); b.sync(); b.event_record(0); fn(); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    return min(ms[1:])

print("%9s %-22s %10s %14s" % ("problems", "launch", "ms", "G steps/s"))
for n in (4096, 65536, 262144, 1 << 20, 1 << 22, 1 << 23):
    p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
    with rp.Batch(n) as b:
        b.set_problems(p0, p1, p2)
        re = b.restart
        ms = best(b, lambda: b.solve(1e-8, 200, 0), re); steps = b.reduce()["total_steps"]
        print("%9d %-22s %10.4f %14.2f" % (n, "gated solve (fused)", ms, steps / ms / 1e6))
        for k in (1, 2, 12, 50):
            ms = best(b, lambda: b.step(k), re)
            print("%9d %-22s %10.4f %14.2f" % (n, "step(%d)" % k, ms, n * k / ms / 1e6))
