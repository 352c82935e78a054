import os, sys
sys.path.insert(0, '/root/repo')
import rocket_path_amd as rp
print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
for nn in (65536, 262144, 1 << 20):
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        for steps in (12, 50):
            ms = []
            for _ in range(5):
                b.set_problems(q0, q1, q2)
                b.restart()
                b.sync(); b.event_record(0); b.step(steps); b.event_record(1); b.sync()
                ms.append(b.event_elapsed_ms(0, 1))
            print("n %7d steps %2d: %.4f ms = %.2f G steps/s" % (nn, steps, min(ms[1:]), nn * steps / min(ms[1:]) / 1e6), flush=True)
