#!/usr/bin/env python3
"""BASELINE configs[3] (8,388,608 F3 problems) on the ONE GPU there is, at STEADY clocks (round 5's probe timed three
passes from an idle chip and read 74 G where the steady batch-size log read 103.7 G for the same batch).  Three forms, each run back to back
for >= 60 ms before its clock starts and then timed over several passes: the 8 shards of 1,048,576 (what each of 8 ranks would run) one after
the other on one stream; the same 8 shards as 8 jobs of an rp_pipeline on two streams; ONE batch of 8,388,608.  From bare positions, with
the summary reduction of every shard; the summaries of the three forms must agree."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP initialises: streams that share one of the default 4 hardware queues serialise (profiles/r6_hw_queues.log)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rocket_path_amd as rp
from rocket_path_amd import problems
from hip_util import DeviceBuffer
N, W = 1 << 20, 8
pos = [problems.generate(12345, *problems.shard_range(N * W, r, W), problems.DIST_MONOTONE) for r in range(W)]
dev = []
for p in pos:
    d = DeviceBuffer(3 * 8 * N); d.write(np.stack(p)); dev.append(d)
whole = DeviceBuffer(3 * 8 * N * W); whole.write(np.stack([np.concatenate([p[j] for p in pos]) for j in range(3)]))
lead = rp.Batch(N)
shards = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(W - 1)]
big = rp.Batch(N * W)
outs = [DeviceBuffer(32 * N) for _ in range(4)]

def run_shards():
    for b, d in zip(shards, dev):
        b.set_problems_device(d.ptr, d.ptr + 8 * N, d.ptr + 16 * N); b.solve(1e-8, 200, 0)
def sum_shards():
    parts = [b.reduce() for b in shards]
    return {"max_residual_sq": max(p["max_residual_sq"] for p in parts), "max_gap": max(p["max_gap"] for p in parts),
            "n_converged": sum(p["n_converged"] for p in parts), "total_steps": sum(p["total_steps"] for p in parts)}
def run_whole():
    big.set_problems_device(whole.ptr, whole.ptr + 8 * N * W, whole.ptr + 16 * N * W); big.solve(1e-8, 200, 0)
pipe = rp.Pipeline(N, depth=8, n_streams=2)
def run_pipe():
    for j, d in enumerate(dev):
        pipe.submit(d.ptr, d.ptr + 8 * N, d.ptr + 16 * N, d_out=outs[j % 4].ptr)

def steady(fn, sync, reps=6, warm=40):
    for _ in range(warm): fn()
    sync()
    t = time.perf_counter()
    for _ in range(reps): fn()
    sync()
    return (time.perf_counter() - t) / reps * 1e3
run_shards(); tot = sum_shards()
run_whole(); one = big.reduce()
assert tot == one, (tot, one)
for rnd in range(3):
    ms_s = steady(run_shards, lead.sync)
    ms_p = steady(run_pipe, pipe.wait)
    ms_w = steady(run_whole, big.sync, warm=10)
    g = tot["total_steps"] / 1e6
    print("8 x 1,048,576 from bare positions, steady clocks: 8 shards on one stream %.3f ms (%.2f G steps/s); 8 jobs of rp_pipeline on two streams %.3f ms (%.2f G); "
          "ONE batch of 8,388,608 %.3f ms (%.2f G); %d steps, all %d converged, summaries identical" % (ms_s, g / ms_s, ms_p, g / ms_p, ms_w, g / ms_w, int(tot["total_steps"]), int(tot["n_converged"])), flush=True)
