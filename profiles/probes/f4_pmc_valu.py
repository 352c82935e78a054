#!/usr/bin/env python3
"""F4, 1 Mi problems x 50 fused steps in the two fp32-storage modes, for `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES`: instructions per lane-step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for dtype in (rp.DTYPE_F32_STATE, rp.DTYPE_F32):
    with rp.Batch(N, rp.VARIANT_F4, dtype) as b:
        for _ in range(2):
            b.set_problems(p0, p1, p2); b.restart(); b.step(50); b.sync()
