#!/usr/bin/env python3
"""The fused gated solve in one launch against the same solve in rounds (straggler hand-off) on states whose step counts the scheduled
order does not predict: ms per solve, G steps/s, idle lane-steps of the one-launch form (from the iteration counts, 64 consecutive positions per wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
N = 1 << 20
rng = np.random.default_rng(5)

def feasible(p0, p1, p2):
    with rp.Batch(len(p0)) as b:
        b.set_problems(p0, p1, p2); b.restart(); return b.get_state()

fams = {}
for dist, dn in ((0, "monotone"), (2, "non-monotone")):
    base = feasible(*rp.problems.generate(12345, 0, N, dist))
    fams[dn + ", feasible start (through set_state)"] = base
    s = base.copy(); s[:, 1] += 0.1; s[:, 2] += 0.1; fams[dn + ", durations +0.1"] = s
    s = base.copy(); s[:, 1] += 1.0; s[:, 2] += 1.0; fams[dn + ", durations +1"] = s
    s = base.copy(); s[:, 3:11] = 100.0; fams[dn + ", multipliers 100"] = s
    s = base.copy(); s[:, 3:11] = 0.01; fams[dn + ", multipliers 0.01"] = s
    s = base.copy(); s[:, 3:11] = 10.0 ** rng.uniform(-3, 2, (N, 1)); fams[dn + ", multipliers 10^U(-3,2) per problem"] = s
    s = base.copy(); s[:, 0] = rng.uniform(-10, 10, N); fams[dn + ", vel1 U(-10,10)"] = s
mix = np.concatenate([feasible(*rp.problems.generate(91 + d, 0, N // 2 if d == 0 else N // 4, d)) for d in (0, 1, 2)])[rng.permutation(N)]
fams["three distributions mixed, feasible starts"] = mix
print("%-58s %9s %5s | %-20s | %-20s | %-20s | %s" % ("state family (1,048,576 problems through rp_batch_set_state)", "steps", "max", "plain kernel (-1)", "watched (0, default)", "5 rounds x 24 lanes", "idle lane-steps, one launch"))
for name, st in fams.items():
    res = []
    for rounds in (-1, 0, 5):
        with rp.Batch(N) as b:
            b.set_params(handoff_rounds=rounds)
            ms = []
            for rep in range(3):
                b.set_state(st); b.sync(); b.event_record(0); b.solve(1e-8, 200, 0); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
            it, status = b.get_iters()
            if rounds == -1:
                slot = b.slot_map(); order = np.argsort(slot)
                x = it[order].astype(np.int64); pad = (-len(x)) % 64
                x = np.concatenate([x, np.zeros(pad, np.int64)]).reshape(-1, 64)
                idle = 1.0 - x.sum() / (x.max(axis=1).sum() * 64.0)
                ref_it, ref_st = it, b.get_state()
            else:
                assert np.array_equal(it, ref_it) and np.array_equal(b.get_state(), ref_st, equal_nan=True)
            res.append(min(ms))
    tot = float(ref_it.sum())
    print("%-58s %9.0f %5d | %8.3f ms %6.2f G | %8.3f ms %6.2f G | %8.3f ms %6.2f G | %.3f" % (name, tot, ref_it.max(), res[0], tot / res[0] / 1e6, res[1], tot / res[1] / 1e6, res[2], tot / res[2] / 1e6, idle), flush=True)
