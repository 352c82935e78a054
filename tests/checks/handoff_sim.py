#!/usr/bin/env python3
"""Idle lane-steps of the gated solve under the straggler hand-off rule (a wave gives its last <= T stepping lanes D more steps, then hands them
to the next round), simulated on the oracle's step counts for state families entered through set_state.  (CPU only; profiles/r6_handoff_sim.log.
The simulation counts lane-steps; on the chip a stalled lane's step costs ~50 normal ones, which is why the measured gains are smaller:
profiles/r6_state_families.log.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp  # noqa: E402
from oracle_api import Oracle, StepInfo  # noqa: E402

o = Oracle()
N = 1 << 17



def shipped_key(p0, p1, p2):
    d0, d1 = np.abs(p1 - p0), np.abs(p2 - p1)
    lo, hi = np.minimum(d0, d1), np.maximum(d0, d1)
    with np.errstate(all="ignore"):
        cls = np.where((lo / hi * 64 >= 0) & (lo / hi * 64 < 64), np.floor(lo / hi * 64), 63).astype(np.int64)
    lvl = np.clip((hi.astype(np.float32).view(np.uint32).astype(np.int64) >> 21) - ((127 + 2) << 2), 0, 31)
    rev = (p1 - p0) * (p2 - p1) < 0
    return np.where(rev, 0, (cls << 5) | lvl)


def sim(steps, order, T, D, rounds, penalty=0.5):
    """adaptive hand-off: a wave gives its last <= T live lanes D more steps, then hands them off; `rounds` rounds, the last one runs to the end"""
    rem = steps[order].astype(np.int64)
    busy = float(rem.sum()); paid = 0.0; handed = []
    for r in range(rounds):
        if len(rem) == 0: break
        pad = (-len(rem)) % 64
        x = np.concatenate([rem, np.zeros(pad, np.int64)]).reshape(-1, 64)
        xs = np.sort(x, axis=1)
        last = r == rounds - 1
        if last or T == 0:
            t_exit = xs[:, -1]
        else:
            t_T = xs[:, 63 - T]            # time when at most T lanes are still live
            t_exit = np.minimum(xs[:, -1], t_T + D)
        paid += 64.0 * t_exit.sum()
        left = x - t_exit[:, None]
        nxt = left[left > 0]
        handed.append(len(nxt))
        paid += penalty * len(nxt) * 2      # store + reload path, in lane-step equivalents
        rem = nxt
    return 1.0 - busy / paid, handed

rng = np.random.default_rng(5)
fams = {}
for dist, dn in ((0, "monotone"), (2, "non-monotone")):
    p0, p1, p2 = rp.problems.generate(12345, 0, N, dist)
    base = o.batch_init_feasible(3, p0, p1, p2)
    fams[dn + " feasible start"] = base
    s = base.copy(); s[:, 1] += 1.0; s[:, 2] += 1.0; fams[dn + " durations +1"] = s
    s = base.copy(); s[:, 1] += rng.choice([0.1, 1.0, 0.0], N); s[:, 2] += rng.choice([0.1, 1.0, 0.0], N); fams[dn + " durations + random{0,.1,1}"] = s
    s = base.copy(); s[:, 3:11] = 10.0 ** rng.uniform(-3, 2, (N, 1)); fams[dn + " multipliers 10^U(-3,2) per problem"] = s
    s = base.copy(); s[:, 3:11] = 10.0 ** rng.uniform(-2, 1, (N, 8)); fams[dn + " multipliers 10^U(-2,1) each"] = s
    s = base.copy(); s[:, 0] = rng.uniform(-1, 1, N) * 10; fams[dn + " vel1 nudged"] = s
parts = []
for dist in (0, 1, 2):
    q = rp.problems.generate(777, 0, N // 4, dist); parts.append(o.batch_init_feasible(3, *q))
q = rp.problems.generate(778, 0, N // 4, 0); x = o.batch_init_feasible(3, *q); x[:, 3:11] = 10.0 ** rng.uniform(-3, 2, (N // 4, 1)); x[:, 1] += 1; parts.append(x)
fams["mixed+nudged shuffled"] = np.concatenate(parts)[rng.permutation(N)]
fams["mixed 3 dists shuffled"] = np.concatenate([o.batch_init_feasible(3, *rp.problems.generate(91 + d, 0, N // 2 if d == 0 else N // 4, d)) for d in (0, 1, 2)])[rng.permutation(N)]
res = {}
for name, st0 in fams.items():
    st = st0.copy()
    steps = np.asarray(o.batch_solve_gated(3, st, 1e-8, 200)[0])
    res[name] = (steps, np.argsort(shipped_key(st0[:, 11], st0[:, 13], st0[:, 14]), kind="stable"))
for (T, D, R) in ((0, 0, 1), (16, 0, 4), (16, 1, 4), (8, 1, 4), (8, 2, 4), (16, 2, 5), (24, 1, 5), (32, 1, 6), (32, 2, 6)):
    print("--- T=%d D=%d rounds=%d" % (T, D, R))
    for name, (steps, order) in res.items():
        f, h = sim(steps, order, T, D, R)
        f0, h0 = sim(steps, np.arange(len(steps)), T, D, R)
        print("  %-46s key order: idle %.3f handed %s | problem order: idle %.3f" % (name, f, h[:-1] if len(h) > 1 else [], f0))
