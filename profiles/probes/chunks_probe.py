"""Probe for the scheduled layout: gated solve of the benchmark batch with the problems pre-sorted on the HOST by the
segment-length ratio (what a permuted HBM layout would give), one 64-problem chunk per single-wave block (RP_CHUNKS=f|r).
usage: chunks_probe.py none|global|window:W"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocket_path_amd as rp
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
d0, d1 = np.abs(p1 - p0), np.abs(p2 - p1)
ratio = np.minimum(d0, d1) / np.maximum(d0, d1)
if mode == "global":
    o = np.argsort(ratio, kind="stable")
elif mode.startswith("window:"):
    W = int(mode.split(":")[1])
    o = np.concatenate([s + np.argsort(ratio[s:s + W], kind="stable") for s in range(0, N, W)])
    if len(sys.argv) > 2 and sys.argv[2] == "interleave":      # chunk c of every window next to each other: rank-major
        nw = N // W
        o = o.reshape(nw, W // 64, 64).transpose(1, 0, 2).reshape(-1)
else:
    o = np.arange(N)
p0, p1, p2 = p0[o].copy(), p1[o].copy(), p2[o].copy()
bs = [rp.Batch(N) for _ in range(12)]
for rep in range(2):
    for b in bs: b.set_problems(p0, p1, p2)
    ms = []
    for b in bs:
        b.sync(); b.event_record(0); b.solve(1e-8, 200, 0); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    it, st = bs[0].get_iters()
    ms.sort(); print("%s RP_CHUNKS=%s gated 1M: med %.4f best %.4f ms  %.2f G steps/s  (steps %d)" % (" ".join(sys.argv[1:]), os.environ.get("RP_CHUNKS"), ms[len(ms)//2], ms[0], it.sum() / ms[len(ms)//2] / 1e6, it.sum()))
