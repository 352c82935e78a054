#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s8
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rounds or idle_lane" > $O/rounds_tests.log 2>&1; echo "rounds tests rc $?"; tail -5 $O/rounds_tests.log
echo skip probe
