#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s14
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
