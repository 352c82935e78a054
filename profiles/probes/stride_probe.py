#!/usr/bin/env python3
"""Does the power-of-two distance between the field arrays (8 MiB at n = 2^20) cost bandwidth?  k = 0 (loads + stores only)
and k = 1 launches over batches created with different stride paddings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp
N = 1 << 20
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for pad in [int(x) for x in os.environ.get("PADS", "0,256,512,768,1024,2304,4352,8448,16640,33024").split(",")]:
    os.environ["RP_STRIDE_PAD"] = str(pad)
    bs = [rp.Batch(N) for _ in range(10)]
    out = []
    for k in (0, 1):
        if k == 0:
            os.environ["RP_STREAM_PROBE"] = "1"
        for b in bs:
            b.set_problems(p0, p1, p2)
        ms = []
        for b in bs:
            b.sync(); b.event_record(0); b.step(k); b.event_record(1); b.sync()
            ms.append(b.event_elapsed_ms(0, 1))
        os.environ.pop("RP_STREAM_PROBE", None)
        ms.sort()
        out.append("k=%d med %.4f best %.4f ms (%.0f GB/s on 200 B)" % (k, ms[len(ms) // 2], ms[0], 200 * N / ms[len(ms) // 2] / 1e6))
    print("pad %6d elems: %s | %s" % (pad, out[0], out[1]), flush=True)
    for b in bs:
        b.close()
