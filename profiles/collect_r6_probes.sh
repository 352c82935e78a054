#!/bin/bash
# Round-6 probe logs (GPU box, repo root):  bash profiles/collect_r6_probes.sh  -> gpurun_out/r6probes; copied to profiles/r6_*.log afterwards.
export TMPDIR=/tmp
O=gpurun_out/r6probes
mkdir -p $O
# the driver-run suite with the lines its tests print (ties, certified decisions, worst errors, idle lane-steps)
python3 -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-600 > $O/gpu_suite.log
tail -2 $O/gpu_suite.log
# what still carried idle-chip numbers, re-measured at steady clocks
python3 profiles/probes/config4_one_gpu_probe_r6.py 2>&1 | grep -v amdgpu.ids > $O/config4_one_gpu.log
python3 profiles/probes/f4_k1_size_probe.py 2>&1 | grep -v amdgpu.ids > $O/f4_k1_size.log
# rp_pipeline arrangements and the 256-thread / one-wave forms of the scheduling pass on one box
python3 profiles/probes/pipeline_probe.py 2>&1 | grep -v amdgpu.ids > $O/pipeline_probe.log
# states the internal order does not predict: plain kernel, watched kernel, rounds
python3 profiles/probes/rounds_probe.py 2>&1 | grep -v amdgpu.ids > $O/state_families.log
# launch times of every Newton kernel; F4's 50 fused steps in the three number modes; configs[1]
python3 profiles/probes/r4_speed_probe.py 2>&1 | grep -v amdgpu.ids > $O/speed_probe.log
python3 profiles/probes/f4_speed_probe.py 2>&1 | grep -v amdgpu.ids > $O/f4_speed.log
python3 profiles/probes/fixed50_time.py 2>&1 | grep -v amdgpu.ids > $O/fixed50_time.log
# 1.44 M fresh, distinct gated solves (and 12 fixed steps of the same problems) against the oracle
python3 tests/checks/fuzz_parity.py > $O/fuzz_parity.log 2>&1
tail -3 $O/fuzz_parity.log
echo probes collected
