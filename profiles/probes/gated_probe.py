import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
bs = [rp.Batch(N) for _ in range(12)]
for rep in range(2):
    for b in bs: b.set_problems(p0, p1, p2)
    ms = []
    for b in bs:
        b.sync(); b.event_record(0); b.solve(1e-8, 200, 0); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    ms.sort(); print("gated 1M: med %.4f best %.4f ms  %.2f G steps/s" % (ms[len(ms)//2], ms[0], 16308345 / ms[len(ms)//2] / 1e6))
    for b in bs: b.set_problems(p0, p1, p2)
    ms = []
    for b in bs:
        b.sync(); b.event_record(0); b.step(12); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    ms.sort(); print("k=12 1M:  med %.4f best %.4f ms  %.2f G steps/s" % (ms[len(ms)//2], ms[0], 12 * N / ms[len(ms)//2] / 1e6))
