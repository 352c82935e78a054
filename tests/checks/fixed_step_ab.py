#!/usr/bin/env python3
"""Fixed-step launches (F3 and F4, every number mode, three launch shapes) saved for a bit-for-bit comparison between two builds of
the library (RP_BATCH_LIB), e.g. -DRP_UNIFORM_LOOPS=0 against the default:
    python tests/checks/fixed_step_ab.py gpurun_out/a.npz;  RP_BATCH_LIB=... python tests/checks/fixed_step_ab.py gpurun_out/b.npz
    python tests/checks/inplace_ab.py cmp gpurun_out/a.npz gpurun_out/b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
out = {}
n = 131072 + 33
for variant in (3, 4):
    for dist in (0, 2):
        p0, p1, p2 = rp.problems.generate(777 + dist, 0, n, dist)
        for dtype in (rp.DTYPE_F64, rp.DTYPE_F32_STATE, rp.DTYPE_F32):
            for bt, mb in ((0.5, 100), (0.5, 3), (0.25, 40)):
                with rp.Batch(n, variant, dtype) as a:
                    a.set_params(backtrack=bt, max_backtracks=mb)
                    a.set_problems(p0, p1, p2)
                    st = a.get_state()
                    st[::9, 1] *= 0.6                  # some starts outside the feasible set
                    a.set_state(st)
                    a.step(50)                          # chunk kernel, through the post-convergence regime
                    tag = "v%d_dist%d_d%d_bt%g_mb%d" % (variant, dist, dtype, bt, mb)
                    out[tag + "_k50"] = a.get_state()
                    a.set_state(st)
                    for _ in range(6):
                        a.step(1)                       # streaming kernel
                    a.step(2)
                    out[tag + "_k1x6_k2"] = a.get_state()
                    nf, nr = a.step_counted(1)          # the counting twin
                    out[tag + "_nf"], out[tag + "_nr"] = nf, nr
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
