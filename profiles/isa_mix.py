#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (static counts, per basic block)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ops = collections.Counter()
blocks = []
cur = ["entry", collections.Counter()]
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), collections.Counter()]
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
    if m and not l.strip().startswith((".", ";")):
        ops[m.group(1)] += 1
        cur[1][m.group(1)] += 1
blocks.append(cur)
print("total", sum(ops.values()))
for k, v in ops.most_common(40):
    print("  %-26s %d" % (k, v))
print("blocks (name, instrs, f64 valu, div_scale, rcp):")
for name, c in blocks:
    n = sum(c.values())
    if n >= 25:
        f64 = sum(v for k, v in c.items() if k.endswith("_f64"))
        print("  %-14s %5d %5d %4d %4d" % (name, n, f64, c.get("v_div_scale_f64", 0), c.get("v_rcp_f64_e32", 0) + c.get("v_rcp_f64", 0)))
