#!/usr/bin/env python3
"""Is an F4 lane's line-search cost persistent from step to step?  (Would regrouping lanes between steps pay?)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 8192
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
F = np.zeros((50, N), int); R = np.zeros((50, N), int)
with rp.Batch(N, rp.VARIANT_F4) as b:
    b.set_problems(p0, p1, p2)
    for s in range(50):
        nf, nr = b.step_counted(1)
        F[s] = nf; R[s] = nr
cost = 100 * R + 37 * F          # rough instruction cost of a lane's line search per step
for s in (10, 20, 30, 40, 48):
    hi = R[s] > 20
    nxt = R[s + 1] > 20
    print("step %d: lanes with >20 residual halvings %.3f; of those, again at the next step %.2f; corr(cost_s, cost_s+1) %.2f" % (
        s, hi.mean(), (hi & nxt).sum() / max(1, hi.sum()), np.corrcoef(cost[s], cost[s + 1])[0, 1]))
# what regrouping by last step's cost would buy: wave cost = sum over steps of max over lanes
def wave_cost(order_by_prev):
    tot = 0
    perm = np.arange(N)
    for s in range(50):
        if order_by_prev and s > 0:
            perm = np.argsort(cost[s - 1].reshape(-1, 512), axis=1, kind="stable") + (np.arange(N // 512) * 512)[:, None]
            perm = perm.reshape(-1)
        c = cost[s][perm].reshape(-1, 64)
        tot += c.max(axis=1).sum()
    return tot
a, bb = wave_cost(False), wave_cost(True)
print("line-search instructions per wave over 50 steps: batch order %.0f, regrouped every step by the previous step's cost %.0f (lane mean %.0f)" % (
    a / (N / 64), bb / (N / 64), cost.sum() / N))
