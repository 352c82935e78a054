#!/usr/bin/env python3
"""How many lane-steps of the gated solve idle under a given order of the batch (CPU only: the oracle's step counts of the
benchmark batch).  A wave holds 64 consecutive positions and runs until its slowest lane is done."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp  # noqa: E402
from oracle_api import Oracle  # noqa: E402

N = 1 << 20
o = Oracle()
p0, p1, p2 = rp.problems.generate(12345, 0, N, rp.problems.DIST_MONOTONE)
state = o.batch_init_feasible(3, p0, p1, p2)
steps, _ = o.batch_solve_gated(3, state, 1e-8, 200)
steps = np.asarray(steps)
d0, d1 = np.abs(p1 - p0), np.abs(p2 - p1)
lo, hi = np.minimum(d0, d1), np.maximum(d0, d1)
ratio = lo / hi


def idle(order):
    x = steps[order].reshape(-1, 64)
    return 1.0 - x.sum() / (x.max(axis=1).sum() * 64.0)


def windows(w, key):
    return np.concatenate([s + np.argsort(key[s:s + w], kind="stable") for s in range(0, N, w)])


print("steps per problem: min %d mean %.2f max %d" % (steps.min(), steps.mean(), steps.max()))
print("problem order                                  idle %.4f" % idle(np.arange(N)))
print("512-problem tiles sorted by ratio (round 1)    idle %.4f" % idle(windows(512, ratio)))
print("4,096-problem windows sorted by ratio          idle %.4f" % idle(windows(4096, ratio)))
print("global sort by ratio                           idle %.4f" % idle(np.argsort(ratio, kind="stable")))
cls = np.where((ratio * 64 >= 0) & (ratio * 64 < 64), np.floor(ratio * 64), 63).astype(np.uint64)
bits = hi.astype(np.float32).view(np.uint32).astype(np.uint64) >> np.uint64(5)
round2 = np.argsort((cls << np.uint64(26)) | bits, kind="stable")
print("round 2: 32-bit key (ratio class, length)      idle %.4f" % idle(round2))
lvl64 = np.clip((hi.astype(np.float32).view(np.uint32).astype(np.int64) >> 20) - ((127 + 2) << 3), 0, 63)
print("12-bit key (64 classes x 64 levels)            idle %.4f" % idle(np.argsort((cls.astype(np.int64) << 6) | lvl64, kind="stable")))
lvl = np.clip((hi.astype(np.float32).view(np.uint32).astype(np.int64) >> 21) - ((127 + 2) << 2), 0, 31)
shipped = np.argsort((cls.astype(np.int64) << 5) | lvl, kind="stable")
print("schedule.hip: 11-bit key (64 classes x 32 lvls) idle %.4f" % idle(shipped))
x = steps[shipped].reshape(-1, 64)
print("   step counts inside a chunk differ by %.2f on average; chunk maxima %d .. %d" % ((x.max(axis=1) - x.min(axis=1)).mean(), x.max(axis=1).min(), x.max(axis=1).max()))
print("sorted by the step count itself (bound)        idle %.6f" % idle(np.argsort(steps, kind="stable")))

# the other two distributions (2^18 problems each) under the shipped key, with and without its rule for reversals
N = 1 << 18
for dist, name in ((rp.problems.DIST_MONOTONE, "monotone"), (rp.problems.DIST_REFERENCE_LIKE, "reference-like"),
                   (rp.problems.DIST_NON_MONOTONE, "non-monotone")):
    p0, p1, p2 = rp.problems.generate(12345, 0, N, dist)
    state = o.batch_init_feasible(3, p0, p1, p2)
    steps = np.asarray(o.batch_solve_gated(3, state, 1e-8, 200)[0])
    d0, d1 = np.abs(p1 - p0), np.abs(p2 - p1)
    lo, hi = np.minimum(d0, d1), np.maximum(d0, d1)
    cls = np.minimum(np.floor(lo / hi * 64), 63).astype(np.int64)
    lvl = np.clip((hi.astype(np.float32).view(np.uint32).astype(np.int64) >> 21) - ((127 + 2) << 2), 0, 31)
    rev = (p1 - p0) * (p2 - p1) < 0
    print("%-15s reversals %6d (their steps: %s)   idle: problem order %.4f, classes x levels only %.4f, reversals -> key 0 (shipped) %.4f"
          % (name, rev.sum(), np.unique(steps[rev]).tolist(), idle(np.arange(N)), idle(np.argsort((cls << 5) | lvl, kind="stable")),
             idle(np.argsort(np.where(rev, 0, (cls << 5) | lvl), kind="stable"))))
