#!/bin/bash
set -o pipefail
O=gpurun_out/r6s2
mkdir -p $O
timeout -k 10 600 python profiles/probes/pipeline_probe.py > $O/pipeline_probe.log 2>&1; echo "probe rc $?"; tail -25 $O/pipeline_probe.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log
