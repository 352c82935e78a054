#!/usr/bin/env python3
"""Decision-level comparison with the oracle: per-step feasibility / residual halving counts over 50 fixed steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp
from oracle_api import Oracle, StepInfo
o = Oracle()
N = 2048
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
aos = o.batch_init_feasible(3, p0, p1, p2)
info = StepInfo()
rh = np.zeros((N, 50), int); fh = np.zeros((N, 50), int)
for i in range(N):
    v = aos[i]
    for s in range(50):
        o.step(3, v, info); rh[i, s] = info.resid_halvings; fh[i, s] = info.feas_halvings
grh = np.zeros((N, 50), int); gfh = np.zeros((N, 50), int)
with rp.Batch(N) as b:
    b.set_problems(p0, p1, p2)
    for s in range(50):
        nf, nr = b.step_counted(1)
        gfh[:, s] = nf; grh[:, s] = nr
    st = b.get_state()
print("state err after 50:", np.max(np.abs(st[:, :3] - aos[:, :3]) / np.maximum(np.abs(aos[:, :3]), 1)))
for s in range(50):
    print("step %2d feas: mism %4d (gpu max %3d, orc max %3d) | resid: mism %4d gpu mean %.1f max %3d, orc mean %.1f max %3d" % (
        s, (gfh[:, s] != fh[:, s]).sum(), gfh[:, s].max(), fh[:, s].max(), (grh[:, s] != rh[:, s]).sum(), grh[:, s].mean(), grh[:, s].max(), rh[:, s].mean(), rh[:, s].max()))
np.savez(os.path.join(ROOT, "gpurun_out", "halvings.npz"), grh=grh, gfh=gfh, rh=rh, fh=fh)
