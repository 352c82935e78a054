#!/usr/bin/env python3
"""Tiny workload for PMC passes: 1M problems, 12 fused ungated steps, then one fused gated solve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
b = rp.Batch(N)
for _ in range(2):
    b.set_problems(p0, p1, p2)
    b.step(12)
    b.sync()
for _ in range(2):
    b.set_problems(p0, p1, p2)
    b.solve(1e-8, 200, 0)
    b.sync()
b.set_problems(p0, p1, p2)
b.solve(1e-8, 200, 1)     # host-polled: one gated launch per Newton step
b.sync()
