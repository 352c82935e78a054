#!/usr/bin/env python3
"""Per wave-step counters of the fixed-step launches of fixed50_pmc_probe_r5.py (k = 50, 50, 5, 10, 15, 20, 30, 40 in that order): differences
between launches, divided by waves x steps.   python3 fixed50_pmc_summary.py <rocprof dir> [label]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; label = sys.argv[2] if len(sys.argv) > 2 else d
f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_steps_chunks" not in r["Kernel_Name"]:
        continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ks = (50, 50, 5, 10, 15, 20, 30, 40)
disp = list(rows.values())
assert len(disp) == len(ks), len(disp)
by_k = {k: v for k, v in zip(ks, disp)}
waves = by_k[50].get("SQ_WAVES", 1024.0)
def per(a, b):
    return {c: (by_k[b][c] - by_k[a][c]) / (waves * (b - a)) for c in by_k[a] if c != "SQ_WAVES"}
early, late = per(5, 15), per(20, 40)
print("%-28s steps 6-15: %s" % (label, "  ".join("%s %.1f" % (k.replace("SQ_", ""), v) for k, v in sorted(early.items()))))
print("%-28s steps 21-40: %s" % (label, "  ".join("%s %.1f" % (k.replace("SQ_", ""), v) for k, v in sorted(late.items()))))
