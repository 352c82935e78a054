#!/usr/bin/env python3
"""Launch times of the Newton kernels at 1 Mi (and 65,536) problems: gated fused solve, k = 12, k = 50, k = 1, configs[1].  A/B across builds with RP_BATCH_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
bs = [rp.Batch(N) for _ in range(10)]
def timed(fn, prep):
    ms = []
    for b in bs:
        prep(b); b.sync(); b.event_record(0); fn(b); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    ms.sort(); return ms[len(ms) // 2], ms[0]
def start(b):
    b.set_problems(p0, p1, p2); b.restart()
for rep in range(2):
    med, best = timed(lambda b: b.solve(1e-8, 200, 0), start)
    print("gated 1M (from written start): med %.4f best %.4f ms  %.2f G steps/s" % (med, best, 16308345 / med / 1e6))
    med, best = timed(lambda b: b.solve(1e-8, 200, 0), lambda b: b.set_problems(p0, p1, p2))
    print("gated 1M (START):              med %.4f best %.4f ms  %.2f G steps/s" % (med, best, 16308345 / med / 1e6))
    med, best = timed(lambda b: b.step(12), start)
    print("k=12 1M:  med %.4f best %.4f ms  %.2f G steps/s" % (med, best, 12 * N / med / 1e6))
    med, best = timed(lambda b: b.step(1), start)
    print("k=1 1M:   med %.4f best %.4f ms  %.2f TB/s on 200 B" % (med, best, 200 * N / med / 1e9))
    med, best = timed(lambda b: b.step(2), start)
    print("k=2 1M:   med %.4f best %.4f ms" % (med, best))
med, best = timed(lambda b: b.step(50), start)
print("k=50 1M:  med %.4f best %.4f ms  %.2f G steps/s" % (med, best, 50 * N / med / 1e6))
for b in bs: b.close()
for nn in (65536, 262144):
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        for steps in (12, 50):
            ms = []
            for _ in range(6):
                b.set_problems(q0, q1, q2); b.restart(); b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            print("n %7d steps %2d: %.4f ms = %.2f G steps/s" % (nn, steps, min(ms[1:]), nn * steps / min(ms[1:]) / 1e6), flush=True)
