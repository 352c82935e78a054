#!/bin/bash
# Round-5 probe logs (GPU box, repo root):  bash profiles/collect_r5_probes.sh  -> gpurun_out/r5probes; copied to profiles/r5_*.log afterwards.
export TMPDIR=/tmp
O=gpurun_out/r5probes
mkdir -p $O
# the driver-run suite with the lines its tests print (ties, certified decisions, worst errors)
python3 -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-500 > $O/gpu_suite.log
tail -2 $O/gpu_suite.log
# every decision that differs from the oracle's, with its distance from the threshold (1.0 M problem-steps)
python3 tests/checks/decision_margins.py > $O/decision_margins.log 2>&1
# the power controller's transient from an idle chip
python3 profiles/probes/transient_probe.py 2>&1 | grep -v amdgpu.ids > $O/transient.log
# BASELINE configs[3] on the one device there is (one batch and 8 logical shards), round-5 kernels
python3 profiles/probes/config4_one_gpu_probe.py 2>&1 | grep -v amdgpu.ids > $O/config4_one_gpu.log
# launch times of every Newton kernel; F4's 50 fused steps in the three number modes
python3 profiles/probes/r4_speed_probe.py 2>&1 | grep -v amdgpu.ids > $O/speed_probe.log
python3 profiles/probes/f4_speed_probe.py 2>&1 | grep -v amdgpu.ids > $O/f4_speed.log
# 1.44 M fresh, distinct gated solves (and 12 fixed steps of the same problems) against the oracle
python3 tests/checks/fuzz_parity.py > $O/fuzz_parity.log 2>&1
tail -3 $O/fuzz_parity.log
echo probes collected
