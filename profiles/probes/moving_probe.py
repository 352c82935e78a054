# built with -DRP_DIAG_MOVING (make HIPFLAGS="... -DRP_DIAG_MOVING"): rp_batch_step_counted then returns the number of full residual
# evaluations (trial point != x) in place of the feasibility halvings
import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import rocket_path_amd as rp
N = 2048
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
with rp.Batch(N) as b:
    b.set_problems(p0, p1, p2)
    for s in range(50):
        nm, nr = b.step_counted(1)
        if s >= 14: print("step %2d moving evals: mean %.2f max %3d  p99 %.0f | resid halvings mean %.1f max %d" % (s, nm.mean(), nm.max(), np.quantile(nm, .99), nr.mean(), nr.max()))
