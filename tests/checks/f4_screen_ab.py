#!/usr/bin/env python3
"""F4 with and without the proofs that skip certain feasibility trials (ip_core.h: ray_proof, infeasible_beyond_doubt; build flag
-DRP_FEAS_SCREEN=0 evaluates every trial): no decision may change, so 50 fused steps must agree bit for bit in all three number
modes, also with other line-search settings and with non-zero end velocities.  GPU box, repo root:
    python tests/checks/f4_screen_ab.py run gpurun_out/ab_on.npz
    RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_no_screen.so python tests/checks/f4_screen_ab.py run gpurun_out/ab_off.npz
    python tests/checks/inplace_ab.py cmp gpurun_out/ab_on.npz gpurun_out/ab_off.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
out = {}
n = 262144 + 77
for dist in (0, 1, 2):
    p0, p1, p2 = rp.problems.generate(4242 + dist, 0, n, dist)
    for dtype in (rp.DTYPE_F64, rp.DTYPE_F32_STATE, rp.DTYPE_F32):
        for bt, mb in ((0.5, 100), (0.5, 12), (0.25, 40)):
            with rp.Batch(n, rp.VARIANT_F4, dtype) as a:
                a.set_params(backtrack=bt, max_backtracks=mb)
                a.set_problems(p0, p1, p2)
                a.step(50)
                out["dist%d_d%d_bt%g_mb%d" % (dist, dtype, bt, mb)] = a.get_state()
    # non-zero end velocities (the general instantiations), one step at a time and fused
    with rp.Batch(n, rp.VARIANT_F4) as a:
        a.set_problems(p0, p1, p2)
        st = a.get_state()
        st[:, 8] = np.linspace(-3.0, 3.0, n)
        st[:, 11] = np.linspace(2.0, -2.0, n)
        a.set_state(st)
        a.step(30)
        out["dist%d_nzv_fused" % dist] = a.get_state()
        a.set_state(st)
        for _ in range(30):
            a.step(1)
        out["dist%d_nzv_single" % dist] = a.get_state()
# the gated kernel's in-place step makes the same walk: F4 solves (they stall; the stall detector stops them) and gated rounds
for dist in (0, 2):
    p0, p1, p2 = rp.problems.generate(5150 + dist, 0, n, dist)
    for dtype in (rp.DTYPE_F64, rp.DTYPE_F32_STATE, rp.DTYPE_F32):
        with rp.Batch(n, rp.VARIANT_F4, dtype) as a:
            a.set_problems(p0, p1, p2)
            a.solve(1e-6, 40, 0)
            out["gated_dist%d_d%d_solve40" % (dist, dtype)] = a.get_state()
            out["gated_dist%d_d%d_iters" % (dist, dtype)], out["gated_dist%d_d%d_status" % (dist, dtype)] = a.get_iters()
            a.set_params(stall_window=8)
            a.set_problems(p0, p1, p2)
            a.solve(1e-6, 200, 0)
            out["gated_dist%d_d%d_stall" % (dist, dtype)] = a.get_state()
            out["gated_dist%d_d%d_stall_iters" % (dist, dtype)], out["gated_dist%d_d%d_stall_status" % (dist, dtype)] = a.get_iters()
np.savez(sys.argv[2], **out)
print("wrote", sys.argv[2], len(out), "arrays")
