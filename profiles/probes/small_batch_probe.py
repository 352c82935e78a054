#!/usr/bin/env python3
"""Batches that cannot fill the chip (BASELINE configs[1] is one): launch time of F3 fixed steps over batch size x step count, in the
pre-convergence regime (k steps from the feasible start) and in the post-convergence regime (k more steps from the state 26 steps
leave), always through the register-column chunk kernel (tuning build: RP_REG_COLUMN_UPTO covers every size here).  From the grid:
the fixed cost of a launch, the time per step at 1, 2, 3, 4 waves per SIMD, and what a form that splits a problem over lanes could
gain (same work in more, shorter waves).  RP_LANES_PER_WAVE (tuning build) leaves the upper lanes of every wave empty."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"), "RP_REG_COLUMN_UPTO =", os.environ.get("RP_REG_COLUMN_UPTO", "(default)"),
      "RP_LANES_PER_WAVE =", os.environ.get("RP_LANES_PER_WAVE", "(64)"), flush=True)
sizes = [int(x) for x in os.environ.get("SIZES", "16384,32768,65536,131072,196608,262144").split(",")]
ks = [int(x) for x in os.environ.get("KS", "2,6,12").split(",")]
for nn in sizes:
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        row = []
        for steps in ks:
            ms = []
            for _ in range(5):
                b.set_problems(q0, q1, q2); b.restart(); b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            row.append(min(ms[1:]))
        rowf = []
        for steps in ks:
            ms = []
            for _ in range(4):
                b.set_problems(q0, q1, q2); b.restart(); b.step(26); b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            rowf.append(min(ms[1:]))
        ms = []
        for _ in range(4):
            b.set_problems(q0, q1, q2); b.restart(); b.sync(); b.event_record(0); b.step(50); b.event_record(1); b.sync()
            ms.append(b.event_elapsed_ms(0, 1))
        t50 = min(ms[1:])
        per = lambda r: (r[-1] - r[0]) / (ks[-1] - ks[0]) * 1e3
        print("n %7d  from start k=%s: %s ms  (%.3f us/step)   past convergence: %s ms (%.3f us/step)   k=50: %.4f ms = %.2f G steps/s" % (
            nn, ks, " ".join("%.4f" % x for x in row), per(row), " ".join("%.4f" % x for x in rowf), per(rowf), t50, nn * 50 / t50 / 1e6), flush=True)
