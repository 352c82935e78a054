#!/usr/bin/env python3
"""Search for F4 states at which the closed-form count of proven feasibility halvings (ip_core.h, skip_certain_halvings) is
unsound without its monotonicity condition: points that are themselves beyond doubt infeasible (g(0) > 0) with a convex
quadratic g along the Newton ray that is positive at the first trial, negative in between and positive again near 0.
Run against a library built with -DRP_RAY_ASSUME_MONOTONE (round 3's form); every state where the stepping launch (proofs on)
and the counted launch (double arithmetic: no proofs, every trial evaluated) disagree is such a state.  GPU box, repo root:
    RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_ray_r3.so python tests/checks/f4_ray_search.py gpurun_out/f4_ray_cases.npz
The states found are committed as tests/golden/f4_ray_cases.npz (inputs only: fp32-representable F4 rows)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rocket_path_amd as rp  # noqa: E402

print("library:", os.environ.get("RP_BATCH_LIB", "(in-tree)"))
n = 262144
rng = np.random.RandomState(20261005)
found = []
tried = 0
with rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as src, rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as a, \
        rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as b:
    for rnd in range(48):
        dist = rnd % 3
        p0, p1, p2 = rp.problems.generate(9000 + rnd, 0, n, dist)
        src.set_problems(p0, p1, p2)
        src.step(rnd % 12)                       # somewhere along the trajectory
        st = src.get_state()
        scale = (0.002, 0.01, 0.05, 0.2)[rnd % 4]
        st[:, 1] *= 1.0 - scale * rng.uniform(0, 1, n)      # durations a little (or a lot) too short: |a| beyond the limit
        st[:, 2] *= 1.0 - scale * rng.uniform(0, 1, n)
        st[:, 0] += rng.normal(0, 30 * scale, n)
        st = st.astype(np.float32).astype(np.float64)
        a.set_state(st)
        b.set_state(st)
        nf, nr = a.step_counted(1)
        b.step(1)
        diff = np.any(a.get_state() != b.get_state(), axis=1)
        tried += n
        if diff.any():
            idx = np.nonzero(diff)[0]
            found.append(st[idx])
            print("round %d (dist %d, scale %g, %d steps in): %d states where the two launches disagree; counted halvings there: %s"
                  % (rnd, dist, scale, rnd % 12, len(idx), nf[idx][:8]), flush=True)
cases = np.concatenate(found) if found else np.zeros((0, 12))
print("tried %d states, found %d" % (tried, len(cases)))
np.savez_compressed(sys.argv[1], states=cases[:512])
