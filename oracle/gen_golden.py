#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- run in the build container (needs oracle/_ref).

TEST INFRASTRUCTURE ONLY.  Generator = the C restatement (oracle/ip_oracle.c) with its linear
solve delegated to the reference's own vendored Eigen 3.3.0 ColPivHouseholderQR compiled from
/root/reference/libs/eigen into oracle/_ref/ (make -C oracle ref).  Every array written is then
checked, bit for bit, against the reference's OWN hot-path functions (oracle/_ref/libref_hotpath.so
= the GL-free line ranges of onedpath_ip.cpp / onedpath2_ip.cpp compiled from where they lie): the
fixtures are the reference's outputs, which tests/test_oracle_reference.py re-asserts on every run
in the build container.

    python oracle/gen_golden.py            # rewrites tests/golden/f3_*.npz, f4_steps.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from oracle_api import Oracle, Reference, StepInfo  # noqa: E402
from rocket_path_amd import problems  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 12345


def trajectory(orc, variant, var, steps):
    m = 3 + orc.num_constraints(variant)
    states = np.zeros((steps + 1, m))
    halv = np.zeros((steps, 2), dtype=np.int32)
    gaps = np.zeros(steps + 1)
    states[0] = var[:m]
    gaps[0] = orc.gap(variant, var)
    info = StepInfo()
    for s in range(steps):
        orc.step(variant, var, info)
        states[s + 1] = var[:m]
        halv[s] = (info.feas_halvings, info.resid_halvings)
        gaps[s + 1] = orc.gap(variant, var)
    return states, halv, gaps


def main():
    orc = Oracle(eigen=True)
    os.makedirs(OUT, exist_ok=True)

    # (i) named F3 / F4 trajectories
    out = {}
    v = orc.init_default(3)
    out["default_const"] = v[11:].copy()
    out["default_states"], out["default_halvings"], out["default_gaps"] = trajectory(orc, 3, v, 50)
    v = orc.init_stuck()
    out["stuck_const"] = v[11:].copy()
    out["stuck_states"], out["stuck_halvings"], out["stuck_gaps"] = trajectory(orc, 3, v, 30)
    v = orc.init_default(3)
    v[13] = 350.0                       # infeasible start of SURVEY.md 8c
    out["infeasible_const"] = v[11:].copy()
    out["infeasible_states"], out["infeasible_halvings"], out["infeasible_gaps"] = trajectory(orc, 3, v, 4)
    v = orc.init_default(4)
    out["f4_default_const"] = v[7:].copy()
    out["f4_default_states"], out["f4_default_halvings"], out["f4_default_gaps"] = trajectory(orc, 4, v, 50)
    np.savez_compressed(os.path.join(OUT, "f3_trajectories.npz"), **out)

    # (ii) seeded random F3 problems: 4096 monotone, 1024 non-monotone, 1024 reference-like
    parts = [(problems.DIST_MONOTONE, 4096), (problems.DIST_NON_MONOTONE, 1024), (problems.DIST_REFERENCE_LIKE, 1024)]
    pos, dist = [], []
    first = 0
    for d, n in parts:
        p0, p1, p2 = problems.generate(SEED, first, n, d)
        pos.append(np.stack([p0, p1, p2], axis=1))
        dist.append(np.full(n, d, dtype=np.int32))
        first += n
    pos = np.concatenate(pos)
    dist = np.concatenate(dist)
    n = len(pos)
    init = np.zeros((n, 16))
    after1 = np.zeros((n, 11))
    after5 = np.zeros((n, 11))
    after50 = np.zeros((n, 11))
    gated = np.zeros((n, 11))
    iters = np.zeros(n, dtype=np.int32)
    for i in range(n):
        v = orc.init_feasible(3, *pos[i])
        init[i] = v
        w = v.copy()
        for s in range(1, 51):
            orc.step(3, w)
            if s == 1:
                after1[i] = w[:11]
            elif s == 5:
                after5[i] = w[:11]
        after50[i] = w[:11]
        w = v.copy()
        iters[i] = orc.solve_gated(3, w, 1e-8, 200)
        gated[i] = w[:11]
    np.savez_compressed(os.path.join(OUT, "f3_batch.npz"), seed=np.int64(SEED), pos=pos, dist=dist, init=init,
                        after1=after1, after5=after5, after50=after50, gated=gated, iters=iters)

    # (iii) F4 single steps from 4096 states sampled along trajectories; inputs are rounded to
    # float32 first so that the fp32 device path starts from bit-identical numbers
    p0, p1, p2 = problems.generate(SEED + 1, 0, 4096, problems.DIST_MONOTONE)
    rng = np.random.RandomState(7)
    nsteps = rng.randint(0, 12, size=4096)
    s_in = np.zeros((4096, 12))
    s_out = np.zeros((4096, 7))
    for i in range(4096):
        v = orc.init_feasible(4, p0[i], p1[i], p2[i])
        for _ in range(nsteps[i]):
            orc.step(4, v)
        v = v.astype(np.float32).astype(np.float64)
        s_in[i] = v
        orc.step(4, v)
        s_out[i] = v[:7]
    np.savez_compressed(os.path.join(OUT, "f4_steps.npz"), state_in=s_in, state_out=s_out, presteps=nsteps.astype(np.int32))
    # the fixtures are the reference's own outputs
    ref = Reference()
    a = init.copy()
    ref.batch_steps(3, a, 1)
    assert np.array_equal(a[:, :11], after1)
    ref.batch_steps(3, a, 4)
    assert np.array_equal(a[:, :11], after5)
    ref.batch_steps(3, a, 45)
    assert np.array_equal(a[:, :11], after50)
    a = init.copy()
    it_ref, _ = ref.batch_solve_gated(3, a, 1e-8, 200)
    assert np.array_equal(a[:, :11], gated) and np.array_equal(it_ref, iters)
    b = s_in.copy()
    ref.batch_steps(4, b, 1)
    assert np.array_equal(b[:, :7], s_out)
    for name, start, variant, nstate in (("default", ref.init_default(3), 3, 11), ("stuck", ref.init_stuck(), 3, 11),
                                         ("f4_default", ref.init_default(4), 4, 7)):
        v = start.copy()
        for s in range(1, len(out[name + "_states"])):
            ref.step(variant, v)
            assert np.array_equal(v[:nstate], out[name + "_states"][s]), (name, s)
    print("wrote", sorted(os.listdir(OUT)), "-- every array equals the compiled reference's output")


if __name__ == "__main__":
    main()
