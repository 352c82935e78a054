#!/usr/bin/env python3
"""Register / LDS / scratch table of the Newton kernels (compile-time, no GPU needed).
    python profiles/kernel_resources.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "rocket_path_amd", "csrc", "ip_kernels.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                    "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, src] + sys.argv[1:],
                   capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
demangle = subprocess.run(["c++filt"] + [x["name"] for x in rows], capture_output=True, text=True).stdout.splitlines()
print("%-62s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "spill", "scratch", "LDS", "occ"))
for x, d in sorted(zip(rows, demangle), key=lambda t: t[1]):
    d = d.replace("void rp::(anonymous namespace)::", "").split("(")[0]
    if not d.startswith(("k_newton", "k_solve", "k_steps")):
        continue
    print("%-62s %5d %5d %7d %6d %4d" % (d, x.get("VGPRs", -1), x.get("VGPRs Spill", 0), x.get("ScratchSize [bytes/lane]", 0),
                                        x.get("LDS Size [bytes/block]", 0), x.get("Occupancy [waves/SIMD]", -1)))
