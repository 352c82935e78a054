"""What the drain at the end of each gated launch costs: 24 batches solved back to back on ONE stream (bench.py's form)
against the same batches alternating between TWO streams, where one launch's drain overlaps the next launch's start.
Wall time around each series (host clock, device idle before and after)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocket_path_amd as rp
N = 1 << 20
K = 24
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
own = [rp.Batch(N) for _ in range(2)]                       # two batches with their own streams
streams = [b.stream() for b in own]
def series(nstreams):
    bs = own[:nstreams] + [rp.Batch(N, stream=streams[i % nstreams]) for i in range(nstreams, K)]
    out = []
    for rep in range(3):
        for b in bs: b.set_problems(p0, p1, p2)
        for b in own: b.sync()
        t0 = time.perf_counter()
        for b in bs: b.solve(1e-8, 200, 0)
        for b in own: b.sync()
        out.append((time.perf_counter() - t0) / K * 1e3)
    steps = bs[0].reduce()["total_steps"]
    for b in bs[nstreams:]: b.close()
    return min(out), steps
for ns in (1, 2, 1, 2):
    ms, steps = series(ns)
    print("%d stream(s): %.4f ms per batch, %.2f G steps/s" % (ns, ms, steps / ms / 1e6))
