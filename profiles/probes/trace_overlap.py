#!/usr/bin/env python3
"""Overlap of the scheduling kernels with the solve kernel in a rocprofv3 --kernel-trace CSV: per kernel name count / mean duration, and for the
last `tail` jobs: fraction of k_sched_* time that lies inside some k_solve_chunks interval, time per job (first solve start to last solve end)."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort(key=lambda r: r[1])
def short(n):
    for k in ("k_solve_chunks", "k_sched_count", "k_sched_scan", "k_sched_scatter", "k_solution", "k_reduce"):
        if k in n: return k
    return n[:40]
solves = [r for r in rows if "k_solve_chunks" in r[0]]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 40
last = solves[-tail:]
t0, t1 = last[0][1], max(r[2] for r in last)
print("kernels in trace:", len(rows), " solves:", len(solves))
import collections
dur = collections.defaultdict(list)
for n, a, b, q, s in rows:
    if a >= t0 and b <= t1: dur[short(n)].append(b - a)
for k, v in sorted(dur.items()): print("  %-18s n %4d  mean %8.1f us  min %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
print("time per job over the last %d solves: %.1f us" % (tail, (t1 - t0) / 1e3 / tail))
# overlap
iv = sorted((r[1], r[2]) for r in rows if "k_solve_chunks" in r[0] and r[2] >= t0 and r[1] <= t1)
def covered(a, b):
    c = 0
    for x, y in iv:
        lo, hi = max(a, x), min(b, y)
        if hi > lo: c += hi - lo
    return min(c, b - a)
sch = [(r[1], r[2]) for r in rows if "k_sched" in r[0] and r[1] >= t0 and r[2] <= t1]
tot = sum(b - a for a, b in sch); cov = sum(covered(a, b) for a, b in sch)
print("k_sched_* time inside a k_solve_chunks interval: %.1f %% (%.1f of %.1f us per job)" % (100.0 * cov / max(tot, 1), cov / 1e3 / tail, tot / 1e3 / tail))
# chip busy: union of all kernel intervals; solves overlapping each other
allv = sorted((r[1], r[2]) for r in rows if r[2] >= t0 and r[1] <= t1)
u = 0; cur_a, cur_b = allv[0]
for a, b in allv[1:]:
    if a > cur_b: u += cur_b - cur_a; cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
u += cur_b - cur_a
print("union of kernel intervals %.1f us of %.1f us (%.1f %% busy)" % (u / 1e3, (t1 - t0) / 1e3, 100.0 * u / (t1 - t0)))
ss = sum(covered(a, b) - (b - a) for a, b in iv)  # not meaningful; placeholder
for r in rows[-14:]: print("   ", short(r[0]), "q", r[3], "start +%.1f us  dur %.1f us" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
