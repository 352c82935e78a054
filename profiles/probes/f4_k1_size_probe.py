#!/usr/bin/env python3
"""F4 k = 1 launches (one launch per Newton step, fp32 storage) at 1 Mi and 8 Mi problems, at steady clocks: is the 0.28-0.34 of the HBM
peak these launches reach at 1 Mi problems an artefact of the launch size?  68 B move per problem-step.  Timed: one step
from the feasible start on six fresh batches back to back (the step the earlier rounds' figure was for), and one step from the state 12
steps leave (the line search has set in), each after ~60 launches of the same kind on a scratch batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
for n in (1 << 20, 1 << 23):
    p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
    for dtype, tag in ((rp.DTYPE_F32_STATE, "fp32 state, fp64 arithmetic"), (rp.DTYPE_F32, "fp32")):
        lead = rp.Batch(n, rp.VARIANT_F4, dtype)
        bs = [lead] + [rp.Batch(n, rp.VARIANT_F4, dtype, stream=lead.stream()) for _ in range(6)]
        scratch = bs.pop()
        for depth in (0, 12):
            for b in bs + [scratch]:
                b.set_problems(p0, p1, p2); b.restart()
                if depth: b.step(depth)
            for _ in range(60):
                scratch.step(1)
                if (_ % 8) == 7: scratch.restart()          # (kept near the start: F4's steps get dearer the further it has gone)
            lead.event_record(0)
            for b in bs: b.step(1)
            lead.event_record(1); lead.sync()
            ms = lead.event_elapsed_ms(0, 1) / len(bs)
            print("F4 %-28s %9d problems, k = 1, step %2d: %.4f ms = %.0f GB/s on 68 B = %.3f of 8 TB/s; %.2f G steps/s" % (
                tag, n, depth + 1, ms, 68.0 * n / ms / 1e6, 68.0 * n / ms / 1e6 / 8000, n / ms / 1e6), flush=True)
        for b in bs[::-1] + [scratch]: b.close()
