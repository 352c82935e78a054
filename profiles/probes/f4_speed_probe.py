#!/usr/bin/env python3
"""F4 (BASELINE configs[4]): 1 Mi problems x 50 fused steps and one k = 1 launch, fp32 state with fp64 arithmetic and pure fp32.  A/B across builds with RP_BATCH_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for dtype, tag in ((rp.DTYPE_F32_STATE, "fp32 state, fp64 arithmetic"), (rp.DTYPE_F32, "fp32"), (rp.DTYPE_F64, "fp64")):
    with rp.Batch(N, rp.VARIANT_F4, dtype) as b:
        for k in (50, 12):
            ms = []
            for _ in range(4):
                b.set_problems(p0, p1, p2); b.restart(); b.sync(); b.event_record(0); b.step(k); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            print("F4 %-28s k = %2d: %.4f ms = %.2f G steps/s" % (tag, k, min(ms[1:]), N * k / min(ms[1:]) / 1e6), flush=True)
