#!/usr/bin/env python3
"""F4 fused fixed steps at 1 Mi problems with growing step counts, for counter passes (SQ_INSTS_VALU ...): instructions per step by phase of the run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
dt = {"f32state": rp.DTYPE_F32_STATE, "f32": rp.DTYPE_F32, "f64": rp.DTYPE_F64}[sys.argv[1] if len(sys.argv) > 1 else "f32state"]
with rp.Batch(N, rp.VARIANT_F4, dt) as b:
    for k in (2, 5, 10, 15, 20, 30, 40, 50):
        b.set_problems(p0, p1, p2); b.restart(); b.step(k); b.sync()
