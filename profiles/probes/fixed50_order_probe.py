"""Why did 65,536 x 50 fixed steps take 0.46 ms after set_problems and 0.34 ms after restart, with bit-identical states?"""
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
    for _ in range(5):
        prep(); c2.sync(); c2.event_record(4); c2.step(50); c2.event_record(5); c2.sync(); ms.append(c2.event_elapsed_ms(4, 5))
    print("%-44s" % label, ["%.4f" % m for m in ms])
with rp.Batch(n) as c2:
    c2.set_problems_device(*ptrs)
    run(c2, lambda: c2.set_problems_device(*ptrs), "set_problems_device")
    run(c2, lambda: c2.restart(), "restart")
    run(c2, lambda: (c2.set_problems_device(*ptrs), c2.restart()), "set_problems_device + restart")
    run(c2, lambda: (c2.restart(), c2.sync(), time.sleep(0.005)), "restart + 5 ms idle")
    run(c2, lambda: (c2.set_problems_device(*ptrs), c2.sync(), c2.step(1), c2.restart()), "set_problems_device + step(1) + restart")
    run(c2, lambda: (c2.set_problems_device(*ptrs), c2.step(50), c2.restart()), "set_problems_device + step(50) + restart")
    run(c2, lambda: (c2.set_problems_device(*ptrs), c2.sync(), time.sleep(0.005), c2.restart()), "set_problems_device + 5 ms idle + restart")
    with rp.Batch(n) as other:
        other.set_problems_device(*ptrs)
        run(c2, lambda: (other.set_problems_device(*ptrs), other.sync(), c2.restart()), "restart (another batch was scheduled before)")
    slot = c2.slot_map()
    print("slot map identity?", np.array_equal(slot, np.arange(n)))
# the same problems handed over already sorted (the batch's order is then the identity)
o = np.argsort(slot)      # prob_of
q0, q1, q2 = p0[o].copy(), p1[o].copy(), p2[o].copy()
with rp.Batch(n) as c3:
    run(c3, lambda: c3.set_problems(q0, q1, q2), "set_problems(host), pre-sorted input")
    run(c3, lambda: c3.restart(), "restart, pre-sorted input")
