#!/usr/bin/env python3
"""Time to solution of the gated solve at 1 Mi problems: reference centring (tiled kernel) vs centring by trial (mu_mode 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
with rp.Batch(N) as b:
    b.set_problems(p0, p1, p2)
    for mode in (0, 1, 0, 1):
        b.set_params(mu_mode=mode)
        ms = []
        for _ in range(4):
            b.restart(); b.sync(); b.event_record(0); b.solve(1e-8, 200, 0); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
        r = b.reduce()
        print("mu_mode %d: %.4f ms, %.0f steps (%.2f per problem), %.2f G steps/s, converged %d, max gap %.2e" % (
            mode, min(ms[1:]), r["total_steps"], r["total_steps"] / N, r["total_steps"] / min(ms[1:]) / 1e6, r["n_converged"], r["max_gap"]))
