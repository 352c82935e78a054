import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for dt, name in ((rp.DTYPE_F64, "f64"), (rp.DTYPE_F32_STATE, "f32state"), (rp.DTYPE_F32, "f32")):
    with rp.Batch(N, rp.VARIANT_F4, dt) as b:
        ms = []
        for _ in range(4):
            b.set_problems(p0, p1, p2); b.sync(); b.event_record(0); b.step(50); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
        print("F4 %s 50 steps: %.4f ms  %.2f G steps/s" % (name, min(ms[1:]), N * 50 / min(ms[1:]) / 1e6), flush=True)
