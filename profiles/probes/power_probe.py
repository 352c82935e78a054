"""Round 3's power probe.  SUPERSEDED by `bench.py --sustain-seconds S`, which reads the device's hwmon files instead of starting rocm-smi (a
`#!/usr/bin/env python3` script) from a process that has initialised the GPU -- never run this one under rocprofv3."""
import os, sys, time, subprocess, threading
sys.path.insert(0, '/root/repo')
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
lead = rp.Batch(N)
bs = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(7)]
for b in bs:
    b.set_problems(p0, p1, p2); b.restart()
lead.sync()
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().splitlines()[-1][:200] if out.strip() else "?"))
        except Exception as e:
            samples.append((time.time(), "err %s" % e))
        time.sleep(0.05)
t = threading.Thread(target=poll); t.start()
time.sleep(0.5)
t0 = time.time()
for rep in range(300):          # ~0.6 s of sustained solves
    for b in bs:
        b.restart(); b.solve(1e-8, 200, 0)
lead.sync()
t1 = time.time()
time.sleep(0.5)
stop = True; t.join()
print("sustained region %.3f .. %.3f s" % (0, t1 - t0))
for ts, s in samples: print("%.3f %s" % (ts - t0, s))
