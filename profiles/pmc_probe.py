#!/usr/bin/env python3
"""Tiny workload for the SQ counter passes (rocprofv3 --pmc ... -- python3 profiles/pmc_probe.py).
Every launch whose per-lane-step instruction counts are read from the counters is an ALL-LANES-ACTIVE launch of a
known number of steps at 1 Mi problems:
    F3 f64          12 fused ungated steps   -> k_steps_chunks<double, double, 3, true>
    F4 f32          50 fused ungated steps   -> k_steps_chunks<float, float, 4, true>     (BASELINE configs[4] itself: F4's line search
    F4 f32 state    50 fused ungated steps   -> k_steps_chunks<float, double, 4, true>     only sets in from step ~6, so 12 steps undercount)
then the gated kernel (k_solve_chunks) on 524,288 identical default problems (15 steps each: its instructions per step without idle lanes),
the benchmark's gated solve (occupancy / busy counters of the real launch), one k = 1 launch, BASELINE configs[1] (65,536 x 50 fixed steps,
and 15 steps of 65,472 problems beside it), and the feasibility move
(k_move_toward_feasibility) on 1 Mi starts with four violated rows each -- bench.py's `neighbours.feasibility_move` workload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

N = 1 << 20
STEPS = 12
STEPS_F4 = 50
p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
for variant, dtype in ((rp.VARIANT_F3, rp.DTYPE_F64), (rp.VARIANT_F4, rp.DTYPE_F32), (rp.VARIANT_F4, rp.DTYPE_F32_STATE)):
    with rp.Batch(N, variant, dtype) as b:
        for _ in range(2):
            b.set_problems(p0, p1, p2)
            b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
            b.step(STEPS if variant == rp.VARIANT_F3 else STEPS_F4)
            b.sync()
# the gated kernel with every lane doing the same work: 524,288 copies of the default problem (15 steps each, no idle
# lane-steps), told apart from the benchmark's launch by its grid size
with rp.Batch(N // 2) as b:
    for _ in range(2):
        b.init_default()
        b.solve(1e-8, 200, 0)
        b.sync()
    assert int(b.reduce()["total_steps"]) == 15 * (N // 2)
with rp.Batch(N) as b:
    for _ in range(2):
        b.set_problems(p0, p1, p2)
        b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
        b.solve(1e-8, 200, 0)
        b.sync()
    b.set_problems(p0, p1, p2)
    b.restart()      # the feasible start written out (set_problems defers it to a fused solve)
    b.step(1)
    b.sync()
# BASELINE configs[1]: 65,536 problems x 50 fixed steps (one wave per SIMD; k_steps_chunks<double, double, 3, true, true>, the register-column
# instantiation), and the first 15 steps of a batch one chunk smaller (65,472 problems: told apart by its grid size) -- the difference is
# the post-convergence regime's share
for nn, kk in ((65536, 50), (65536 - 64, 15)):
    q0, q1, q2 = rp.problems.generate(12345, 0, nn, 0)
    with rp.Batch(nn) as b:
        for _ in range(2):
            b.set_problems(q0, q1, q2)
            b.restart()
            b.step(kk)
            b.sync()
# the feasibility move on 1 Mi starts pushed out of the feasible set (both durations too short: four violated rows each)
with rp.Batch(N) as b:
    b.set_problems(p0, p1, p2)
    st = b.get_state()
    st[:, 1] *= 0.7
    st[:, 2] *= 0.7
    for _ in range(2):
        b.set_state(st)
        b.move_toward_feasibility()
        b.sync()
