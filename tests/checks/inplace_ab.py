#!/usr/bin/env python3
"""The gated kernel's in-place step (ip_core.h: newton_step_inplace) against the form it replaced (newton_step_to with carried
sums, build flag -DRP_GATED_IN_PLACE=0): same functions, same decisions, so every bit of every iterate must agree -- also
where the line search runs out of halvings (budgets of 0, 1, 3: the last, untested step length is taken), with other
backtrack factors, from starts outside the feasible set (every feasibility trial fails), in rounds of launches, F3 and F4,
all three number modes.  GPU box, repo root:
    python tests/checks/inplace_ab.py run gpurun_out/ab_inplace.npz
    RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_not_in_place.so python tests/checks/inplace_ab.py run gpurun_out/ab_to.npz
    python tests/checks/inplace_ab.py cmp gpurun_out/ab_inplace.npz gpurun_out/ab_to.npz
(the second library: hipcc ... -DRP_GATED_IN_PLACE=0 -DRP_GATED_WAVES=3 -c ip_kernels.hip, linked with schedule.o and rp_batch.o)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [(v, d, bt, mb) for v in (3, 4) for d in (0, 1, 2) for bt, mb in ((0.5, 100), (0.5, 0), (0.5, 1), (0.5, 3), (0.25, 40), (0.75, 7))]

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for key in a.files:
        same = np.array_equal(a[key], b[key], equal_nan=True)
        bad += not same
        if not same:
            print("DIFFERENT:", key, "max abs diff", np.nanmax(np.abs(a[key].astype(np.float64) - b[key].astype(np.float64))))
    print("%d arrays compared (%d values), %d differ" % (len(a.files), sum(a[k].size for k in a.files), bad))
    sys.exit(1 if bad else 0)

import rocket_path_amd as rp  # noqa: E402

print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
out = {}
n, k = 2 * 4096 + 100, 9
p0, p1, p2 = rp.problems.generate(8128, 0, n, 0)
for variant, dtype, bt, mb in CASES:
    with rp.Batch(n, variant, dtype) as a:
        a.set_params(backtrack=bt, max_backtracks=mb)
        a.set_problems(p0, p1, p2)
        st = a.get_state()
        st[::7, 1] *= 0.6                      # every seventh problem starts outside the feasible set
        a.set_state(st)
        a.solve(0.0, k, 0)                     # gap < 0 never holds: k gated steps, then RP_ST_MAXITER
        tag = "v%d_d%d_bt%g_mb%d" % (variant, dtype, bt, mb)
        out[tag + "_fused"] = a.get_state()
        out[tag + "_iters"], out[tag + "_status"] = a.get_iters()
        a.set_state(st)
        a.solve(0.0, 4, 0)                     # ... and in rounds: 4 + 5 steps (the carried sums are rebuilt at the start of a launch)
        a.solve(0.0, k, 0)
        out[tag + "_rounds"] = a.get_state()
    # the benchmark's own solve: fresh batch, feasible start formed in registers, every problem to its gate
    with rp.Batch(n, variant, dtype) as a:
        a.set_problems(p0, p1, p2)
        a.solve(1e-8 if variant == 3 else 1e-6, 60, 0)
        out[tag + "_solve"] = a.get_state()
        out[tag + "_solve_iters"], out[tag + "_solve_status"] = a.get_iters()
np.savez(sys.argv[2], **out)
print("wrote", sys.argv[2], len(out), "arrays")
