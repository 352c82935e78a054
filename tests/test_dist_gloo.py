"""N > 1 path on CPU: world_size 2 and 3 over gloo, contiguous shards, the MAX/SUM summary."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from rocket_path_amd import problems, sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(world, n_total, *extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(n_total)] + [str(x) for x in extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return sorted(outs, key=lambda o: o["rank"])


def test_shard_ranges_tile_the_batch():
    for n in (0, 1, 7, 8, 1000, 8388608 + 5):
        for w in (1, 2, 3, 8):
            spans = [problems.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == n
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        problems.shard_range(10, 2, 2)


def test_generator_is_shard_invariant():
    whole = problems.generate(99, 0, 1000, problems.DIST_MONOTONE)
    for w in (2, 3):
        parts = [problems.generate(99, *problems.shard_range(1000, r, w), problems.DIST_MONOTONE) for r in range(w)]
        for k in range(3):
            assert np.array_equal(np.concatenate([p[k] for p in parts]), whole[k])


def test_allreduce_summary_single_process_is_identity():
    t = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    assert sharding.allreduce_summary(t.clone()).tolist() == t.tolist()
    with pytest.raises(ValueError):
        sharding.allreduce_summary(torch.zeros(3, dtype=torch.float64))


def test_forced_collectives_in_a_group_of_one_rank_are_the_identity():
    # bench.py --force-process-group: the summary goes through the backend's all-reduces even at world size 1 (on the GPU box:
    # RCCL; here: gloo, in a child process so that this process keeps no process group)
    import subprocess
    import sys
    code = ("import os, sys, torch, torch.distributed as dist\n"
            "sys.path.insert(0, %r)\n"
            "from rocket_path_amd import sharding\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%r)\n"
            "dist.init_process_group('gloo', rank=0, world_size=1)\n"
            "t = torch.tensor([1.5, -2.0, 3.0, 4.0], dtype=torch.float64)\n"
            "assert sharding.allreduce_summary(t.clone(), force=True).tolist() == t.tolist()\n"
            "assert sharding.allreduce_summary(t.clone()).tolist() == t.tolist()\n"
            "dist.barrier(); dist.destroy_process_group(); print('forced ok')\n") % (os.path.dirname(HERE), str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "forced ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_summary_equals_whole_batch(world, oracle):
    n_total = 601     # not divisible: ragged shards
    outs = _run_world(world, n_total)
    assert [o["first"] for o in outs] == [problems.shard_range(n_total, r, world)[0] for r in range(world)]
    assert sum(o["count"] for o in outs) == n_total
    # every rank holds the same global summary
    for o in outs[1:]:
        assert o["global"] == outs[0]["global"]
    g = outs[0]["global"]
    assert g[0] == max(o["local"][0] for o in outs)
    assert g[1] == max(o["local"][1] for o in outs)
    assert g[2] == sum(o["local"][2] for o in outs)
    assert g[3] == sum(o["local"][3] for o in outs)
    # and it is the summary of the unsharded batch
    p0, p1, p2 = problems.generate(12345, 0, n_total, problems.DIST_MONOTONE)
    aos = oracle.batch_init_feasible(3, p0, p1, p2)
    iters, total = oracle.batch_solve_gated(3, aos, 1e-8, 200)
    gaps = np.array([oracle.gap(3, row) for row in aos])
    assert g[3] == float(total)
    assert g[2] == float(n_total)
    assert g[1] == gaps.max()


def test_solve_with_global_checks_stops_every_rank_at_the_same_check(oracle):
    # an all-reduce of the summary at every convergence check.  All ranks leave the loop at the same check -- the
    # first one at which the GLOBAL converged count reaches n_total -- with the same global summary.
    n_total, k = 301, 4
    outs = _run_world(2, n_total, "polled", k)
    assert outs[0]["global"] == outs[1]["global"] and outs[0]["checks"] == outs[1]["checks"]
    p0, p1, p2 = problems.generate(12345, 0, n_total, problems.DIST_MONOTONE)
    aos = oracle.batch_init_feasible(3, p0, p1, p2)
    iters, total = oracle.batch_solve_gated(3, aos, 1e-8, 200)
    assert outs[0]["global"][2] == float(n_total) and outs[0]["global"][3] == float(total)
    assert outs[0]["checks"] == -(-int(iters.max()) // k)            # the slowest problem of the WHOLE batch decides
    assert max(o["local_max_iters"] for o in outs) == int(iters.max())
    # single process: plain loop
    calls = []
    g, checks = sharding.solve_with_global_checks(lambda: calls.append(1), lambda: torch.tensor([0.0, 0.0, 5.0, 9.0], dtype=torch.float64), 5, 200, 7)
    assert checks == 1 and len(calls) == 1 and g.tolist() == [0.0, 0.0, 5.0, 9.0]
    g, checks = sharding.solve_with_global_checks(lambda: calls.append(1), lambda: torch.tensor([0.0, 1.0, 4.0, 9.0], dtype=torch.float64), 5, 20, 7)
    assert checks == 3                                                 # never converges: ceil(20 / 7) rounds, then stop
