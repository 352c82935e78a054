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

# round 3, second part: kbench.py times the same launch at 0.47 ms where bench.py sees 0.32.  kbench hands its positions over
# from HOST arrays (rp_batch_set_problems: three hipMemcpyAsync host-to-device on the batch's stream, the scheduling pass, a
# stream synchronise), bench from device arrays.  Which ingredient is it?
p0h, p1h, p2h = p0.copy(), p1.copy(), p2.copy()
with rp.Batch(n) as c3:
    c3.set_problems_device(*ptrs)
    c3.restart()
    c3.sync()
    run(c3, lambda: c3.restart(), "restart")
    run(c3, lambda: (c3.set_problems(p0h, p1h, p2h), c3.restart()), "set_problems from HOST arrays + restart")
    run(c3, lambda: (c3.set_problems(p0h, p1h, p2h), c3.restart(), c3.sync()), "set_problems from HOST arrays + restart + sync")
    run(c3, lambda: (c3.set_problems(p0h, p1h, p2h), c3.restart(), c3.sync(), time.sleep(0.002)), "... + 2 ms idle")
    run(c3, lambda: (c3.set_problems_device(*ptrs), c3.restart(), c3.sync()), "set_problems_device + restart + sync")
    run(c3, lambda: c3.restart(), "restart")
with rp.Batch(n) as c4:      # a batch with a stream of its own that never saw a host copy, next to one that did
    c4.set_problems_device(*ptrs)
    c4.restart()
    c4.sync()
    with rp.Batch(n) as other:
        run(c4, lambda: (other.set_problems(p0h, p1h, p2h), other.sync(), c4.restart()), "restart (ANOTHER batch took host arrays on its own stream)")

# third part: is it simply the FIRST long launch of a freshly created batch (new stream, new memory)?
keep = []
for trial in range(4):
    b = rp.Batch(n)
    keep.append(b)                      # keep them alive: every batch gets a stream and memory of its own
    b.set_problems_device(*ptrs)
    b.restart()
    b.sync()
    run(b, lambda: b.restart(), "fresh batch %d (others alive): restart, 6 launches in a row" % trial)
shared = rp.Batch(n, stream=keep[0].stream())
shared.set_problems_device(*ptrs)
shared.restart()
shared.sync()
run(shared, lambda: shared.restart(), "fresh batch on an OLD stream (new memory only)")
for b in keep + [shared]:
    b.close()
