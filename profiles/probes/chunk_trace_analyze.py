"""Timeline statistics from chunk_trace output (one line per chunk: chunk xcc se cu simd start_ticks end_ticks steps; 100 MHz ticks).
usage: chunk_trace_analyze.py label=trace.txt [label=trace.txt ...]"""
import sys
import numpy as np
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    c, xcc, se, cu, simd, t0, t1, steps = np.loadtxt(path, dtype=np.int64).T
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0      # microseconds
    phys = ((xcc * 8 + se) * 16 + cu) * 4 + simd
    ids = np.unique(phys)
    dur = t1 - t0
    print("==", name, "kernel span %.1f us, %d SIMDs, %d chunks" % (t1.max(), len(ids), len(c)))
    print(" chunk duration: mean %.1f us, %.2f us per step (mean steps %.2f)" % (dur.mean(), (dur / steps).mean(), steps.mean()))
    nch = np.array([(phys == p).sum() for p in ids])
    tot = np.array([steps[phys == p].sum() for p in ids])
    fin = np.array([t1[phys == p].max() for p in ids])
    busy = np.array([dur[phys == p].sum() for p in ids])
    print(" chunks per SIMD: min %d max %d" % (nch.min(), nch.max()), "histogram from min:", np.bincount(nch)[nch.min():])
    print(" steps per SIMD: min %d mean %.1f max %d" % (tot.min(), tot.mean(), tot.max()))
    print(" finish time per SIMD: min %.1f mean %.1f max %.1f us" % (fin.min(), fin.mean(), fin.max()))
    print(" wave-slot occupancy: sum(chunk durations) / (3 slots x span x SIMDs) = %.3f" % (busy.sum() / (3 * t1.max() * len(ids))))
    print(" gap between two chunks on a slot: (3 x mean finish - mean busy per SIMD) / (chunks per SIMD - 3) = %.2f us"
          % ((3 * fin.mean() - busy.mean()) / (nch.mean() - 3)))
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc, tt = np.cumsum(ev[:, 1]), ev[:, 0]
    print(" active waves at t us:", ", ".join("%d: %d" % (q, conc[min(np.searchsorted(tt, q), len(conc) - 1)]) for q in (5, 50, 100, 150, 180, 200, 210, 220) if q < t1.max()))
