#!/usr/bin/env python3
"""BASELINE configs[1] timing for A/B builds (RP_BATCH_LIB): 65,536 F3 problems x 50 / x 12 steps and 1 Mi x 50, HIP events, best of 6."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
tag = os.path.basename(os.environ.get("RP_BATCH_LIB", "(in-tree)"))
out = []
for n, k in ((65536, 50), (65536, 12), (1 << 20, 50)):
    p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
    with rp.Batch(n) as b:
        ms = []
        for _ in range(7):
            b.set_problems(p0, p1, p2); b.restart(); b.sync(); b.event_record(0); b.step(k); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    out.append("%d x %d: %.4f ms" % (n, k, min(ms[1:])))
print("%-34s %s" % (tag, "   ".join(out)), flush=True)
