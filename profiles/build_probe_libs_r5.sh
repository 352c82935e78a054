#!/bin/bash
# Builds the A/B libraries of round 5's probes (build container, repo root; the .so files travel to the GPU box with the snapshot and are
# not committed):
#   librp_batch_tuning.so        -DRP_TUNING                                   the environment knobs (RP_REG_COLUMN_UPTO, RP_LANES_PER_WAVE, ...)
#   librp_batch_small_w1.so      ... -DRP_SMALL_WAVES=1                         the register-column kernel with one wave's register budget
#   librp_batch_small_ilp.so     ... -mllvm -amdgpu-sched-strategy=max-ilp      scheduled for instruction-level parallelism (three waves)
#   librp_batch_small_w1ilp.so   ... both
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function -shared -DRP_TUNING"
S="rocket_path_amd/csrc/ip_kernels.hip rocket_path_amd/csrc/schedule.hip rocket_path_amd/csrc/rp_batch.cpp"
H=/opt/rocm/bin/hipcc
$H $F -o profiles/probes/librp_batch_tuning.so $S &
$H $F -DRP_SMALL_WAVES=1 -o profiles/probes/librp_batch_small_w1.so $S &
wait
$H $F -mllvm -amdgpu-sched-strategy=max-ilp -o profiles/probes/librp_batch_small_ilp.so $S &
$H $F -DRP_SMALL_WAVES=1 -mllvm -amdgpu-sched-strategy=max-ilp -o profiles/probes/librp_batch_small_w1ilp.so $S &
wait
ls -la profiles/probes/*.so
