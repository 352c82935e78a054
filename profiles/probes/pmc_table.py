#!/usr/bin/env python3
"""Per-dispatch counter values of a rocprofv3 --pmc run:  python3 pmc_table.py <dir> [kernel substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat and pat not in r["Kernel_Name"]:
        continue
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"].replace("void rp::(anonymous namespace)::", "").split("(")[0][:60], r["Grid_Size"])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in rows.items():
    print(k[0], k[1], "grid", k[2], " ".join("%s=%.0f" % kv for kv in sorted(v.items())))
