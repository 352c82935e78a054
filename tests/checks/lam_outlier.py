#!/usr/bin/env python3
"""Where the multipliers at the gate differ most between the device and the oracle on one seed of the full-size fuzz sweep
(fuzz_parity.py 1048576 ...): the problem, its multipliers from the device, from the oracle with its own QR and from the oracle
with the reference's Eigen QR, and how far the two CPU evaluations are apart on the same problem.
usage: lam_outlier.py SEED DIST [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp
from oracle_api import Oracle, have_ref
seed, dist = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
o = Oracle()
p = rp.problems.generate(seed, 0, n, dist)
init = o.batch_init_feasible(3, *p)
a = init.copy()
it_a, _ = o.batch_solve_gated(3, a, 1e-8, 200)
with rp.Batch(n) as b:
    b.set_problems(*p)
    b.solve(1e-8, 200, 0)
    it_g, _ = b.get_iters()
    x = b.get_state()
rel = lambda u, v: np.abs(u[:, 3:11] - v[:, 3:11]) / np.max(np.abs(v[:, 3:11]), axis=1, keepdims=True)
same = it_g == it_a
eg = np.where(same[:, None], rel(x, a), 0.0).max(axis=1)
order = np.argsort(eg)[::-1][:5]
print("iteration counts equal on %d of %d problems; worst multiplier differences device vs oracle (relative to the problem's largest):" % (same.sum(), n))
e = None
if have_ref():
    e = init.copy()
    Oracle(eigen=True).batch_solve_gated(3, e, 1e-8, 200)
    ee = rel(a, e).max(axis=1)
    print("the two CPU evaluations (own QR / Eigen QR) against each other: worst %.2e (problem %d)" % (ee.max(), ee.argmax()))
for w in order:
    print("problem %d: device vs oracle %.2e%s; x diff %.2e; dX0 %.6f dX1 %.6f; steps %d" % (
        w, eg[w], "" if e is None else ", oracle own QR vs Eigen QR %.2e" % ee[w],
        np.max(np.abs(x[w, :3] - a[w, :3]) / np.maximum(np.abs(a[w, :3]), 1.0)), p[1][w] - p[0][w], p[2][w] - p[1][w], it_g[w]))
    print("   device  ", " ".join("%.6e" % v for v in x[w, 3:11]))
    print("   oracle  ", " ".join("%.6e" % v for v in a[w, 3:11]))
    if e is not None:
        print("   Eigen QR", " ".join("%.6e" % v for v in e[w, 3:11]))
