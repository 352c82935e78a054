import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocket_path_amd as rp
N = 1 << 20
p = rp.problems.generate(12345, 0, N, 0)
for variant, name in ((rp.VARIANT_F4, "F4 f64 (160 VGPRs)"), (rp.VARIANT_F3, "F3 f64 (202 VGPRs)")):
    bs = [rp.Batch(N, variant, rp.DTYPE_F64) for _ in range(4)]
    ms = []
    for b in bs:
        b.set_problems(*p)
    for b in bs:
        b.event_record(0); b.step(12); b.event_record(1); b.sync(); ms.append(b.event_elapsed_ms(0, 1))
    print(name, "grid", os.environ.get("RP_STREAM_GRID", "512"), "k=12 ms", ["%.4f" % m for m in ms])
