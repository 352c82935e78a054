#!/bin/bash
# usage: res.sh [extra flags]   -> resource table of the Newton kernels
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c -o /tmp/isa/ipk.o ip_kernels.hip 2> /tmp/isa/res.txt; grep -E "error" /tmp/isa/res.txt | head; python3 - <<'PY'
import re, subprocess
usage={}; name=None
for line in open('/tmp/isa/res.txt'):
    m=re.search(r"Function Name: (\S+)",line)
    if m: name=m.group(1); usage[name]={}; continue
    m=re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)",line)
    if m and name: usage[name][m.group(1)]=int(m.group(2))
for k,v in usage.items():
    if any(x in k for x in ("k_solve_chunks","k_steps_chunks","k_newton")):
        d=subprocess.run(["/usr/bin/c++filt",k],capture_output=True,text=True).stdout.strip()
        d=d.replace("void rp::(anonymous namespace)::","").split("(")[0]
        print("%-62s VGPR %3d spill %2d scratch %3d LDS %6d occ %d"%(d,v.get("VGPRs",0),v.get("VGPRs Spill",0),v.get("ScratchSize [bytes/lane]",0),v.get("LDS Size [bytes/block]",0),v.get("Occupancy [waves/SIMD]",0)))
PY
