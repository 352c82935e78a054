#!/bin/bash
# Builds the A/B libraries the round-4 probes compare the shipped one with (build container, repo root; the .so files travel to the
# GPU box with the snapshot and are not committed):
#   librp_batch_tuning.so            -DRP_TUNING            the environment knobs of the probes (RP_CHUNKS_FROM_K, RP_STREAM_SCALAR, ...)
#   librp_batch_diag_moving.so       -DRP_DIAG_MOVING       step_counted's first array counts full residual evaluations
#   librp_batch_no_frozen_proof.so   -DRP_FROZEN_PROOF=0    every trial of the post-convergence search is evaluated
#   librp_batch_ray_r3.so            -DRP_RAY_ASSUME_MONOTONE   round 3's closed-form halving count (no monotonicity condition)
#   librp_batch_r4base.so            commit 713c5fd's kernels = round 3's Newton kernels + this round's boundary additions
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function -shared"
S="rocket_path_amd/csrc/ip_kernels.hip rocket_path_amd/csrc/schedule.hip rocket_path_amd/csrc/rp_batch.cpp"
H=/opt/rocm/bin/hipcc
$H $F -DRP_TUNING -o profiles/probes/librp_batch_tuning.so $S &
$H $F -DRP_DIAG_MOVING -o profiles/probes/librp_batch_diag_moving.so $S &
wait
$H $F -DRP_FROZEN_PROOF=0 -o profiles/probes/librp_batch_no_frozen_proof.so $S &
$H $F -DRP_RAY_ASSUME_MONOTONE -o profiles/probes/librp_batch_ray_r3.so $S &
wait
if [ ! -f profiles/probes/librp_batch_r4base.so ]; then
    rm -rf /tmp/rp_r4base && git worktree add -f /tmp/rp_r4base 713c5fd > /dev/null
    (cd /tmp/rp_r4base && $H $F -o "$OLDPWD/profiles/probes/librp_batch_r4base.so" $S)
    git worktree remove --force /tmp/rp_r4base
fi
ls -la profiles/probes/*.so
