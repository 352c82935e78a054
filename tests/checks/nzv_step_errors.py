#!/usr/bin/env python3
"""Step-by-step error against the oracle on 4,096 problems with non-zero end velocities (the set of
test_non_zero_end_velocities_against_oracle), fused k steps and k single steps, plus the halving counts of the worst problem.
This is the check that caught Cramer's rule losing eps * w^2 at step 2 (profiles/r3_tuning.md): 2.6e-7 on 16 problems, where the
pivoted elimination stays below 6e-11."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp
from oracle_api import Oracle, StepInfo
O = Oracle()
g3 = np.load(os.path.join(ROOT, "tests", "golden", "f3_batch.npz"))
m = 4096
st = g3["init"][:m].copy()
st[:, 12] = np.linspace(-3.0, 3.0, m)
st[:, 15] = np.linspace(2.0, -2.0, m)
def serr(a, b): return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0), axis=1)
exp = st.copy()
info = StepInfo()
for k in range(1, 6):
    O.batch_steps(3, exp, 1)
    with rp.Batch(m) as c:
        c.set_state(st); c.step(k); out = c.get_state()
    with rp.Batch(m) as c:
        c.set_state(st)
        for _ in range(k): c.step(1)
        out1 = c.get_state()
    e, e1 = serr(out[:, :3], exp[:, :3]), serr(out1[:, :3], exp[:, :3])
    i = int(np.argmax(e))
    print("k=%d fused: max err %.3e at %d (n>1e-10: %d) ; k x step(1): %.3e (n %d)" % (k, e.max(), i, (e > 1e-10).sum(), e1.max(), (e1 > 1e-10).sum()))
# halvings of the worst problem, step by step, GPU vs oracle
k = 4
with rp.Batch(m) as c:
    c.set_state(st); c.step(k); out = c.get_state()
ex = st.copy(); O.batch_steps(3, ex, k)
i = int(np.argmax(serr(out[:, :3], ex[:, :3])))
v = st[i].copy()
with rp.Batch(m) as c:
    c.set_state(st)
    for s in range(k):
        nf, nr = c.step_counted(1)
        O.step(3, v, info)
        g = c.get_state()[i]
        print("step %d problem %d: gpu halvings (%d,%d) oracle (%d,%d) err %.3e  state gpu %s oracle %s" % (s + 1, i, nf[i], nr[i], info.feas_halvings, info.resid_halvings,
              np.max(np.abs(g[:3] - v[:3]) / np.maximum(np.abs(v[:3]), 1)), g[:3], v[:3]))
