import os, sys
sys.path.insert(0, '/root/repo')
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
bs = [rp.Batch(N) for _ in range(16)]
for k in (1, 0, 1, 0):
    if k == 0: os.environ["RP_STREAM_PROBE"] = "1"
    for b in bs: b.set_problems(p0, p1, p2)
    ms = []
    for b in bs:
        b.sync(); b.event_record(0); b.step(k); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    os.environ.pop("RP_STREAM_PROBE", None)
    ms.sort()
    print("k=%d med %.4f best %.4f ms  %.0f GB/s on 200 B" % (k, ms[len(ms)//2], ms[0], 200 * N / ms[len(ms)//2] / 1e6))
