#!/usr/bin/env python3
"""F3 fixed steps at batch sizes around three waves per SIMD: register column against LDS column (tuning build: RP_REG_COLUMN_UPTO)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"), "RP_REG_COLUMN_UPTO =", os.environ.get("RP_REG_COLUMN_UPTO", "(default 196608)"))
for nn in (16384, 65536, 131072, 196608):
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        for steps in (12, 50):
            ms = []
            for _ in range(6):
                b.set_problems(q0, q1, q2); b.restart(); b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            print("n %7d steps %2d: %.4f ms = %.2f G steps/s" % (nn, steps, min(ms[1:]), nn * steps / min(ms[1:]) / 1e6), flush=True)
