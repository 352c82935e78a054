#!/usr/bin/env python3
"""BASELINE configs[1] (65,536 F3 problems, fixed steps) for counter passes: the k = 50 launch twice, then launches of k = 5, 10, 15, 20,
30, 40 steps from the same start (differences between them = the counters of steps 6-10, 11-15, ... of the 50), all through
k_steps_chunks<double, double, 3, true, true> (register column).   rocprofv3 --pmc ... -- python3 this"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
n = int(os.environ.get("N", "65536"))
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n) as b:
    for k in (50, 50, 5, 10, 15, 20, 30, 40):
        b.set_problems(p0, p1, p2); b.restart(); b.sync(); b.step(k); b.sync()
