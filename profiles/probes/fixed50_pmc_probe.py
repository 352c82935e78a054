#!/usr/bin/env python3
"""configs[1] (65,536 x 50 fixed steps) twice, for counter passes: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -- python3 this"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
n = 65536
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n) as b:
    for _ in range(2):
        b.set_problems(p0, p1, p2); b.restart(); b.step(50); b.sync()
    for k in (10, 20, 30, 40):
        b.set_problems(p0, p1, p2); b.restart(); b.step(k); b.sync()
