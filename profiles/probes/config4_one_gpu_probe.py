#!/usr/bin/env python3
"""BASELINE configs[3] (8,388,608 F3 problems) has only ever met ONE GPU here.  For the record: the whole workload on that one GPU,
as one batch and as 8 logical shards of 1,048,576 (what each of 8 ranks would run), from bare positions, with the final summary
reduction -- steps, time, steps/s -- and the two must agree on every summary number."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402
from rocket_path_amd import problems, sharding  # noqa: E402

N, W = 1 << 20, 8
pos = [problems.generate(12345, *problems.shard_range(N * W, r, W), problems.DIST_MONOTONE) for r in range(W)]
dev = [torch.from_numpy(np.stack(p)).cuda() for p in pos]
whole = torch.cat(dev, dim=1).contiguous()
lead = rp.Batch(N)
shards = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(W - 1)]
big = rp.Batch(N * W)


def run_shards():
    for b, d in zip(shards, dev):
        b.set_problems_device(*[d[j].data_ptr() for j in range(3)])
        b.solve(1e-8, 200, 0)
    return [b.reduce() for b in shards]


def run_whole():
    big.set_problems_device(*[whole[j].data_ptr() for j in range(3)])
    big.solve(1e-8, 200, 0)
    return big.reduce()


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    parts = run_shards()
    t1 = time.perf_counter()
    one = run_whole()
    t2 = time.perf_counter()
    tot = {"max_residual_sq": max(p["max_residual_sq"] for p in parts), "max_gap": max(p["max_gap"] for p in parts),
           "n_converged": sum(p["n_converged"] for p in parts), "total_steps": sum(p["total_steps"] for p in parts)}
    assert tot == one, (tot, one)
    print("8 x 1,048,576 as 8 logical shards on one stream: %.3f ms (%.2f G steps/s); as ONE batch of 8,388,608: %.3f ms (%.2f G steps/s); "
          "%d steps, all %d converged, summaries identical" % ((t1 - t0) * 1e3, tot["total_steps"] / (t1 - t0) / 1e9, (t2 - t1) * 1e3,
                                                                one["total_steps"] / (t2 - t1) / 1e9, int(one["total_steps"]), int(one["n_converged"])))
