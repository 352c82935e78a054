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


def main():
    n_total = int(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    t, first, count = local_summary(orc, n_total, rank, world)
    local = t.clone()
    dist.barrier()
    sharding.allreduce_summary(t)
    print(json.dumps({"rank": rank, "first": first, "count": count, "local": local.tolist(), "global": t.tolist()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
