"""One rank of the world_size-N CPU (gloo) rehearsal of the sharded path.  Test helper.

Each rank generates ITS shard of the seeded batch, "solves" it with the CPU oracle (the GPU
is not available here; the oracle stands in for the device kernel in this host-logic test),
forms the local 4-value summary and all-reduces it exactly as bench.py does on RCCL."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from oracle_api import Oracle  # noqa: E402
from rocket_path_amd import problems, sharding  # noqa: E402


def local_summary(orc, n_total, rank, world):
    first, count = sharding.shard_range(n_total, rank, world)
    p0, p1, p2 = problems.generate(12345, first, count, problems.DIST_MONOTONE)
    aos = orc.batch_init_feasible(3, p0, p1, p2)
    iters, total = orc.batch_solve_gated(3, aos, 1e-8, 200, threads=1)
    gaps = np.array([orc.gap(3, row) for row in aos])
    res = np.array([orc.residual_norm(3, row, g / 80.0) for row, g in zip(aos, gaps)])
    conv = float((gaps < 1e-8).sum())
    return torch.tensor([res.max() if count else 0.0, gaps.max() if count else -1.7976931348623157e308, conv, float(total)],
                        dtype=torch.float64), first, count


def polled(orc, n_total, rank, world, k):
    """sharding.solve_with_global_checks with the oracle standing in for Batch.solve_launch / reduce_device."""
    first, count = sharding.shard_range(n_total, rank, world)
    p0, p1, p2 = problems.generate(12345, first, count, problems.DIST_MONOTONE)
    aos = orc.batch_init_feasible(3, p0, p1, p2)
    iters = np.zeros(count, dtype=np.int64)

    def launch():
        for i in range(count):
            for _ in range(k):
                if orc.gap(3, aos[i]) < 1e-8 or iters[i] >= 200:
                    break
                orc.step(3, aos[i])
                iters[i] += 1

    def local():
        gaps = np.array([orc.gap(3, row) for row in aos]) if count else np.zeros(0)
        res = np.array([orc.residual_norm(3, row, g / 80.0) for row, g in zip(aos, gaps)]) if count else np.zeros(0)
        return torch.tensor([res.max() if count else 0.0, gaps.max() if count else -1.7976931348623157e308,
                             float((gaps < 1e-8).sum()), float(iters.sum())], dtype=torch.float64)
    g, checks = sharding.solve_with_global_checks(launch, local, n_total, 200, k)
    return g, checks, int(iters.max()) if count else 0


def main():
    n_total = int(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    if len(sys.argv) > 2 and sys.argv[2] == "polled":
        g, checks, local_max = polled(orc, n_total, rank, world, int(sys.argv[3]))
        print(json.dumps({"rank": rank, "global": g.tolist(), "checks": checks, "local_max_iters": local_max}))
        dist.destroy_process_group()
        return
    t, first, count = local_summary(orc, n_total, rank, world)
    local = t.clone()
    dist.barrier()
    sharding.allreduce_summary(t)
    print(json.dumps({"rank": rank, "first": first, "count": count, "local": local.tolist(), "global": t.tolist()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
