#!/usr/bin/env python3
"""GPU box: one F4 step from every golden state in fp32 and fp64; outputs saved for offline analysis
against the oracle (gpurun_out/f4_dump.npz).  Also times 50 fused F4 steps at 1 Mi in both types."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

t = np.load(os.path.join(ROOT, "tests", "golden", "f4_steps.npz"))
n = len(t["state_in"])
out = {}
for dt, name in ((rp.DTYPE_F64, "f64"), (rp.DTYPE_F32, "f32")):
    with rp.Batch(n, rp.VARIANT_F4, dt) as b:
        b.set_state(t["state_in"])
        b.step(1)
        st = b.get_state()
    out[name] = st
    err = np.abs(st[:, :3] - t["state_out"][:, :3]) / np.maximum(np.abs(t["state_out"][:, :3]), 1.0)
    e = err.max(axis=1)
    print(name, "quantiles 50/90/99/99.9/max", [float("%.3g" % np.quantile(e, q)) for q in (.5, .9, .99, .999, 1.0)])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "f4_dump.npz"), **out)

N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for dt, name in ((rp.DTYPE_F64, "f64"), (rp.DTYPE_F32, "f32")):
    with rp.Batch(N, rp.VARIANT_F4, dt) as b:
        for k in (50, 12, 1):
            ms = []
            for _ in range(4):
                b.set_problems(p0, p1, p2)
                b.sync()
                b.event_record(0)
                b.step(k)
                b.event_record(1)
                b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            print("F4 %s k=%d: %.4f ms  %.2f G steps/s" % (name, k, min(ms[1:]), N * k / min(ms[1:]) / 1e6))
