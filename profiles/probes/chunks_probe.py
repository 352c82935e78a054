"""HISTORICAL (kept for the logs profiles/r2_chunks_probe_*.log): ran against probe builds of the library that had the
switches RP_NO_SCHEDULE / RP_PROBE_LINEAR; the shipped library schedules every batch itself (csrc/schedule.hip) and ignores them.
Probe: how much a better schedule would buy.  The problems are pre-sorted on the HOST (what a different scheduled order
inside the batch would give) and the library's own scheduling is switched off (RP_NO_SCHEDULE=1); RP_PROBE_LINEAR=1 makes the
gated kernel walk the chunks from the last to the first.
usage: chunks_probe.py none | ratio | ratio:B+len   (global sorts)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocket_path_amd as rp
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
d0, d1 = np.abs(p1 - p0), np.abs(p2 - p1)
ratio = np.minimum(d0, d1) / np.maximum(d0, d1)
if mode == "ratio":
    o = np.argsort(ratio, kind="stable")
elif mode.startswith("ratio:"):
    B = int(mode.split(":")[1].split("+")[0])
    o = np.lexsort((np.maximum(d0, d1), np.floor(ratio * B)))
else:
    o = np.arange(N)
p0, p1, p2 = p0[o].copy(), p1[o].copy(), p2[o].copy()
bs = [rp.Batch(N) for _ in range(12)]
for rep in range(3):
    for b in bs: b.set_problems(p0, p1, p2)
    ms = []
    for b in bs:
        b.sync(); b.event_record(0); b.solve(1e-8, 200, 0); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    it, st = bs[0].get_iters()
    ms.sort(); print("%s NO_SCHEDULE=%s LINEAR=%s gated 1M: med %.4f best %.4f ms  %.2f G steps/s  (steps %d)" % (mode, os.environ.get("RP_NO_SCHEDULE"), os.environ.get("RP_PROBE_LINEAR"), ms[len(ms)//2], ms[0], it.sum() / ms[len(ms)//2] / 1e6, it.sum()))
