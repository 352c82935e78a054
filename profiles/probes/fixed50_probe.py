#!/usr/bin/env python3
"""Where does a fixed-50 launch of 65,536 problems spend its time?  Same launch with the halving cap lowered."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
n = 65536
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
for nn in (65536, 131072, 262144 - 512):
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        for cap in (100, 60, 30, 10, 2):
            b.set_params(max_backtracks=cap)
            for steps in (20, 50):
                ms = []
                for _ in range(4):
                    b.set_problems(q0, q1, q2)
                    b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                    ms.append(b.event_elapsed_ms(0, 1))
                print("n %7d cap %3d steps %2d: %.4f ms" % (nn, cap, steps, min(ms[1:])), flush=True)
