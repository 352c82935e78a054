"""Round 2 found 65,536 x 50 fixed steps 40 % slower (0.46 vs 0.33 ms) when rocPRIM's sort (23 dispatches, 20 of them
1,152-thread merge passes) had just run on the SAME queue, and worked around it with a helper queue.  The library is gone
(schedule.hip is three kernels of ours on the batch's own queue).  Does the effect come back (a) with our kernels, (b) with
many tiny dispatches on the queue, which is what distinguished the library's sort?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rocket_path_amd as rp
n = 65536
p0, p1, p2 = rp.problems.generate(12345, 0, n, 0)
torch.cuda.init()
d_pos = torch.from_numpy(np.stack([p0, p1, p2])).to("cuda:0")
ptrs = [d_pos[j].data_ptr() for j in range(3)]


def run(c2, prep, label):
    ms = []
    for _ in range(6):
        prep(); c2.event_record(4); c2.step(50); c2.event_record(5); c2.sync(); ms.append(c2.event_elapsed_ms(4, 5))
    print("%-64s" % label, " ".join("%.4f" % m for m in ms))


with rp.Batch(n) as c2, rp.Batch(64, stream=c2.stream()) as tiny:
    tiny.init_default()
    c2.set_problems_device(*ptrs)
    c2.restart()
    c2.sync()
    run(c2, lambda: c2.restart(), "restart")
    run(c2, lambda: (c2.set_problems_device(*ptrs), c2.restart()), "set_problems_device (3 kernels of ours, same queue) + restart")
    run(c2, lambda: (c2.restart(), [tiny.nudge(0, 0.0) for _ in range(20)]), "restart + 20 one-wave dispatches on the same queue")
    run(c2, lambda: (c2.restart(), [tiny.nudge(0, 0.0) for _ in range(200)]), "restart + 200 one-wave dispatches on the same queue")
    run(c2, lambda: (c2.restart(), c2.sync(), time.sleep(0.02)), "restart + 20 ms idle")
    run(c2, lambda: c2.restart(), "restart")
