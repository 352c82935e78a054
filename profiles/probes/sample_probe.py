#!/usr/bin/env python3
"""Plot data of 1 Mi solved problems into device memory (rp_batch_sample_device): ms per pass.  A/B across builds with RP_BATCH_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
from hip_util import DeviceBuffer
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
n = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
with rp.Batch(n) as b, DeviceBuffer(n * 66 * 8) as dp, DeviceBuffer(n * 4 * 8) as da:
    b.set_problems(p0, p1, p2)
    b.solve(1e-8, 200, 0)
    for _ in range(3):
        b.sample_device(dp.ptr, da.ptr)
    b.sync(); b.event_record(0)
    for _ in range(10):
        b.sample_device(dp.ptr, da.ptr)
    b.event_record(1); b.sync()
    ms = b.event_elapsed_ms(0, 1) / 10
    print("sample_device: %.4f ms per pass = %.2f TB/s on 612 B per problem" % (ms, 612.0 * n / ms / 1e9))
    pos, acc = b.sample()
    got = dp.read(np.float64).reshape(n, 66)
    print("device rows equal the host read-back:", bool(np.array_equal(got, pos)), bool(np.array_equal(da.read(np.float64).reshape(n, 4), acc)))
