#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s11
mkdir -p $O
COLD=1 timeout -k 10 600 python profiles/probes/pipeline_probe.py > $O/pipeline_probe_cold.log 2>&1; echo "probe rc $?"; grep -v "round [01]" $O/pipeline_probe_cold.log | tail -12
COLD=0 timeout -k 10 600 python profiles/probes/pipeline_probe.py > $O/pipeline_probe_warm.log 2>&1; echo "probe rc $?"; grep -v "round [01]" $O/pipeline_probe_warm.log | tail -12
