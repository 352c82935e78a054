#!/usr/bin/env python3
"""A short rp_pipeline run for `rocprofv3 --kernel-trace`: MODE = inline | prep | priority, 2 streams, 40 jobs after 80 untimed ones."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP initialises: streams that share one of the default 4 hardware queues serialise (profiles/r6_hw_queues.log)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rocket_path_amd as rp
from hip_util import DeviceBuffer
N = 1 << 20
mode = os.environ.get("MODE", "inline")
q = rp.problems.generate(12345, 0, N, 0)
pos = DeviceBuffer(3 * 8 * N); pos.write(np.stack(q))
outs = [DeviceBuffer(32 * N) for _ in range(4)]
kw = dict(depth=4, n_streams=int(os.environ.get("STREAMS", "2")))
if mode != "inline":
    kw["prep"] = 1 if mode == "prep" else 2
with rp.Pipeline(N, **kw) as pipe:
    for j in range(120):
        pipe.submit(pos.ptr, pos.ptr + 8 * N, pos.ptr + 16 * N, d_out=outs[j % 4].ptr)
    pipe.wait()
print("done", mode)
