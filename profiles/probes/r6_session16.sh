#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6probes
mkdir -p $O
python3 profiles/probes/config4_one_gpu_probe_r6.py 2>&1 | grep -v amdgpu.ids > $O/config4_one_gpu.log; tail -1 $O/config4_one_gpu.log
python3 profiles/probes/pipeline_probe.py 2>&1 | grep -v amdgpu.ids > $O/pipeline_probe.log; tail -3 $O/pipeline_probe.log
